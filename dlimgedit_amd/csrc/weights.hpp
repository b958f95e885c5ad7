// Reader for the DLW v1 tensor file that replaces the reference's .onnx graphs
// (layout documented in dlimgedit_amd/weights.py; file lookup mirrors
// /root/reference/src/session.cpp:79-83).
#pragma once

#include "common.hpp"

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace dlimg {

struct SamGeometry {
    int embed_dim = 0;
    int depth = 0;
    int num_heads = 0;
    int mlp_dim = 0;
    std::vector<int> global_attn_indexes;

    int head_dim() const { return embed_dim / num_heads; }
    bool is_global(int layer) const {
        for (int g : global_attn_indexes)
            if (g == layer) return true;
        return false;
    }
};

struct HostTensor {
    std::vector<int64_t> dims;
    const float* data = nullptr;
    size_t numel() const {
        size_t n = 1;
        for (auto d : dims) n *= (size_t)d;
        return n;
    }
};

class WeightFile {
  public:
    explicit WeightFile(std::string const& path);
    SamGeometry const& geometry() const { return geom_; }
    HostTensor const& get(std::string const& name) const;
    HostTensor const& get(std::string const& name, std::vector<int64_t> const& expect_dims) const;
    bool has(std::string const& name) const { return tensors_.count(name) != 0; }
    std::string const& path() const { return path_; }

  private:
    std::string path_;
    std::vector<char> raw_;
    SamGeometry geom_;
    std::map<std::string, HostTensor> tensors_;
};

}  // namespace dlimg

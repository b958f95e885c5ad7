#include "weights.hpp"

#include <cstring>
#include <fstream>

namespace dlimg {

namespace {

template <typename T> T read_at(std::vector<char> const& raw, size_t off) {
    if (off > raw.size() || sizeof(T) > raw.size() - off) throw Exception("weight file truncated");
    T v;
    std::memcpy(&v, raw.data() + off, sizeof(T));
    return v;
}

}  // namespace

WeightFile::WeightFile(std::string const& path) : path_(path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw Exception("Could not open weight file '" + path + "'");
    std::streamsize size = f.tellg();
    f.seekg(0);
    raw_.resize((size_t)size);
    if (size > 0 && !f.read(raw_.data(), size)) throw Exception("Could not read weight file '" + path + "'");

    if (raw_.size() < 80 || std::memcmp(raw_.data(), "DLIMGSAM", 8) != 0)
        throw Exception("'" + path + "' is not a DLW weight file");
    uint32_t version = read_at<uint32_t>(raw_, 8);
    uint32_t count = read_at<uint32_t>(raw_, 12);
    if (version != 1) throw Exception("'" + path + "': unsupported DLW version " + std::to_string(version));

    int32_t cfg[16];
    std::memcpy(cfg, raw_.data() + 16, sizeof(cfg));
    geom_.embed_dim = cfg[0];
    geom_.depth = cfg[1];
    geom_.num_heads = cfg[2];
    geom_.mlp_dim = cfg[3];
    int n_global = cfg[4];
    // (bounds generous enough for any SAM size; they keep every product of these numbers far inside 64 bits)
    if (geom_.embed_dim <= 0 || geom_.embed_dim > (1 << 16) || geom_.depth <= 0 || geom_.depth > 1024 || geom_.num_heads <= 0 ||
        geom_.num_heads > geom_.embed_dim || geom_.embed_dim % geom_.num_heads != 0 || geom_.mlp_dim <= 0 ||
        geom_.mlp_dim > (1 << 20) || n_global < 0 || n_global > 8)
        throw Exception("'" + path + "': invalid model geometry in header");
    for (int i = 0; i < n_global; ++i) {
        if (cfg[5 + i] < 0 || cfg[5 + i] >= geom_.depth) throw Exception("'" + path + "': invalid model geometry in header");
        geom_.global_attn_indexes.push_back(cfg[5 + i]);
    }

    constexpr size_t entry = 64 + 4 + 4 + 32 + 8 + 8;
    size_t pos = 80;
    for (uint32_t i = 0; i < count; ++i, pos += entry) {
        if (pos + entry > raw_.size()) throw Exception("'" + path + "': tensor table truncated");
        char name[65] = {0};
        std::memcpy(name, raw_.data() + pos, 64);
        uint32_t dtype = read_at<uint32_t>(raw_, pos + 64);
        uint32_t ndim = read_at<uint32_t>(raw_, pos + 68);
        if (dtype != 0 || ndim < 1 || ndim > 4) throw Exception("'" + path + "': unsupported tensor '" + name + "'");
        HostTensor t;
        uint64_t numel = 1;
        for (uint32_t d = 0; d < ndim; ++d) {
            const uint64_t dim = read_at<uint64_t>(raw_, pos + 72 + 8 * d);
            // no dimension and no product of dimensions beyond what the file itself could hold (no wrap-around either)
            if (dim == 0 || dim > raw_.size() || numel > raw_.size() / dim)
                throw Exception("'" + path + "': tensor '" + name + "' out of bounds");
            numel *= dim;
            t.dims.push_back((int64_t)dim);
        }
        uint64_t off = read_at<uint64_t>(raw_, pos + 104);
        uint64_t nbytes = read_at<uint64_t>(raw_, pos + 112);
        if (numel > raw_.size() / 4 || nbytes != numel * 4 || off % 4 || off > raw_.size() || nbytes > raw_.size() - off)
            throw Exception("'" + path + "': tensor '" + name + "' out of bounds");
        t.data = reinterpret_cast<const float*>(raw_.data() + off);
        tensors_.emplace(name, std::move(t));
    }
}

HostTensor const& WeightFile::get(std::string const& name) const {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw Exception("'" + path_ + "': missing tensor '" + name + "'");
    return it->second;
}

HostTensor const& WeightFile::get(std::string const& name, std::vector<int64_t> const& dims) const {
    HostTensor const& t = get(name);
    if (t.dims != dims) throw Exception("'" + path_ + "': tensor '" + name + "' has unexpected shape");
    return t;
}

}  // namespace dlimg

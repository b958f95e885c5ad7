// EnvironmentImpl: backend gate, model directory, lazily loaded SAM model.
// Counterpart of /root/reference/src/environment.{hpp,cpp}; the onnxruntime environment, provider
// probing through dlopen(libcuda) and the thread-count knob are replaced by a HIP device probe.
#pragma once

#include "common.hpp"
#include "sam_model.hpp"

#include <dlimgedit/dlimgedit.h>

#include <atomic>
#include <filesystem>
#include <memory>
#include <vector>
#include <string>

namespace dlimg {

class EnvironmentImpl {
  public:
    dlimg_Backend backend = dlimg_gpu;
    std::filesystem::path model_directory;
    int device = 0;

    // Cheap, cached, never throws (reference: environment.cpp:103-122).
    static bool is_supported(dlimg_Backend backend) noexcept;
    static int device_count() noexcept;

    explicit EnvironmentImpl(dlimg_Options const& options);

    // Weights + execution lanes, created on first use, once per environment (reference:
    // environment.cpp:144-146).  sam_model() hands out the lanes round-robin; lane(i) addresses one.
    SamModel& sam_model();
    SamModel& lane(int index);
    int lane_count();
    // While set, every request goes to lane 0 (per-kernel clocks must not see other lanes' kernels).
    void set_single_lane(bool on) { single_lane_.store(on); }

  private:
    struct SamLanes {
        SamLanes(std::string const& weight_path, int device, int count);
        std::shared_ptr<SamWeights const> weights;
        std::vector<std::unique_ptr<SamModel>> lanes;
    };
    SamLanes& lanes();
    std::string find_sam_weights() const;
    Lazy<SamLanes> sam_;
    std::atomic<unsigned> next_lane_{0};
    std::atomic<bool> single_lane_{false};
};

}  // namespace dlimg

// EnvironmentImpl: backend gate, model directory, lazily loaded SAM model.
// Counterpart of /root/reference/src/environment.{hpp,cpp}; the onnxruntime environment, provider
// probing through dlopen(libcuda) and the thread-count knob are replaced by a HIP device probe.
#pragma once

#include "common.hpp"
#include "sam_model.hpp"

#include <dlimgedit/dlimgedit.h>

#include <filesystem>
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

    // Created on first use, once per environment (reference: environment.cpp:144-146).
    SamModel& sam_model();

  private:
    std::string find_sam_weights() const;
    Lazy<SamModel> sam_;
};

}  // namespace dlimg

// Helpers shared by the extension entry points (ext_api.cpp) and the test / benchmark hooks (test_hooks.cpp, which is
// not part of the product library).
#pragma once

#include "environment.hpp"
#include "segmentation.hpp"

#include <exception>
#include <utility>

namespace dlimg {

dlimg_Result report_error(char const* what) noexcept;   // dlimgedit.cpp

namespace extapi {

template <typename F> int guarded(F&& body) noexcept {
    try {
        body();
        return 0;
    } catch (std::exception const& e) {
        report_error(e.what());
        return 1;
    } catch (...) {
        report_error("Unknown error");
        return 1;
    }
}

inline EnvironmentImpl& impl(dlimg_Environment h) {
    DLIMG_ASSERT(h != nullptr);
    return *reinterpret_cast<EnvironmentImpl*>(h);
}
inline SegmentationImpl& impl(dlimg_Segmentation h) {
    DLIMG_ASSERT(h != nullptr);
    return *reinterpret_cast<SegmentationImpl*>(h);
}

template <typename T> struct Upload {
    DeviceBuffer<T> buf;
    Upload(T const* host, size_t n) {
        if (host && n) {
            buf.reserve(n);
            HIP_CHECK(hipMemcpy(buf.get(), host, n * sizeof(T), hipMemcpyHostToDevice));
        }
    }
    T* get() const { return buf.get(); }
};

template <typename T> void download(T* host, T const* dev, size_t n) {
    if (host && n) HIP_CHECK(hipMemcpy(host, dev, n * sizeof(T), hipMemcpyDeviceToHost));
}

inline void require_gpu() {
    if (!EnvironmentImpl::is_supported(dlimg_gpu)) throw Exception("No supported GPU (gfx950) found");
}

}  // namespace extapi
}  // namespace dlimg

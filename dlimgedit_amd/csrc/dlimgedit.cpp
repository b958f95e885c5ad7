// C-ABI boundary: dlimg_init() and the function table.
// Counterpart of /root/reference/src/dlimgedit.cpp (table order :102-117, try_ trampoline :26-40).
#include "environment.hpp"
#include "image_memory.hpp"
#include "segmentation.hpp"

#include <dlimgedit/dlimgedit.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace dlimg {

// image_io.cpp (host-side PNG reader / writer behind slots 8 and 9)
uint8_t* load_image_file(char const* filepath, int* out_extent, int* out_channels);
void save_image_file(dlimg_ImageView const& img, char const* filepath);

namespace {

// Per-thread so concurrent callers do not trample each other's message
// (the reference keeps one unsynchronised global: src/dlimgedit.cpp:12).
thread_local std::string last_error_;

dlimg_Result fail(char const* what) noexcept {
    try {
        last_error_ = what;
    } catch (...) {
    }
    return dlimg_error;
}

template <typename F> dlimg_Result guarded(F&& body) noexcept {
    try {
        body();
        return dlimg_success;
    } catch (std::exception const& e) {
        return fail(e.what());
    } catch (...) {
        return fail("Unknown error");
    }
}

EnvironmentImpl& impl(dlimg_Environment h) { return *reinterpret_cast<EnvironmentImpl*>(h); }
SegmentationImpl& impl(dlimg_Segmentation h) { return *reinterpret_cast<SegmentationImpl*>(h); }

// ---- slots 0..12 -----------------------------------------------------------------------------

int is_backend_supported(dlimg_Backend backend) { return EnvironmentImpl::is_supported(backend) ? 1 : 0; }

dlimg_Result create_environment(dlimg_Environment* out, dlimg_Options const* options) {
    return guarded([&] {
        DLIMG_ASSERT(out != nullptr && options != nullptr);
        *out = reinterpret_cast<dlimg_Environment>(new EnvironmentImpl(*options));
    });
}

void destroy_environment(dlimg_Environment h) {
    if (h) delete &impl(h);
}

dlimg_Result process_image_for_segmentation(dlimg_Segmentation* out, dlimg_ImageView const* image,
                                            dlimg_Environment env) {
    return guarded([&] {
        DLIMG_ASSERT(out != nullptr && image != nullptr && env != nullptr);
        auto* seg = new SegmentationImpl(impl(env));
        *out = reinterpret_cast<dlimg_Segmentation>(seg);   // caller owns the handle even if process throws
        seg->process(*image);
    });
}

dlimg_Result get_segmentation_mask(dlimg_Segmentation seg, int const* point, int const* region, uint8_t** out_masks,
                                   float* out_accuracy) {
    return guarded([&] {
        DLIMG_ASSERT(seg != nullptr && out_masks != nullptr);
        Point p;
        Region r;
        if (point) p = Point{point[0], point[1]};
        if (region) r = Region{Point{region[0], region[1]}, Point{region[2], region[3]}};
        impl(seg).compute_mask(point ? &p : nullptr, !point && region ? &r : nullptr, out_masks, out_accuracy);
    });
}

void get_segmentation_extent(dlimg_Segmentation seg, int* out_extent) {
    Extent e = impl(seg).extent();
    out_extent[0] = e.width;
    out_extent[1] = e.height;
}

void destroy_segmentation(dlimg_Segmentation h) {
    if (h) delete &impl(h);
}

dlimg_Result segment_objects(dlimg_ImageView const*, uint8_t*, dlimg_Environment) {
    return fail("segment_objects (BiRefNet) is not part of the MI355X build of dlimgedit");
}

dlimg_Result load_image(char const* filepath, int* out_extent, int* out_channels, uint8_t** out_pixels) {
    return guarded([&] {
        DLIMG_ASSERT(filepath != nullptr && out_extent != nullptr && out_channels != nullptr && out_pixels != nullptr);
        // the decoder's buffer moves into the library's image memory (csrc/image_memory.hpp): one allocator behind
        // destroy_image, and pixels that process() can send to the GPU from where they lie
        uint8_t* decoded = load_image_file(filepath, out_extent, out_channels);
        const size_t bytes = (size_t)out_extent[0] * out_extent[1] * *out_channels;
        uint8_t* pixels = image_alloc(bytes);
        if (pixels) std::memcpy(pixels, decoded, bytes);
        delete[] decoded;
        if (!pixels) throw Exception(std::string("Failed to load image ") + filepath + ": out of memory");
        *out_pixels = pixels;
    });
}

dlimg_Result save_image(dlimg_ImageView const* image, char const* filepath) {
    return guarded([&] {
        DLIMG_ASSERT(image != nullptr && filepath != nullptr);
        save_image_file(*image, filepath);
    });
}

uint8_t* create_image(int w, int h, int channels) {
    if (w <= 0 || h <= 0 || channels <= 0) return nullptr;
    return image_alloc((size_t)w * h * channels);
}

void destroy_image(uint8_t const* pixels) { image_free(pixels); }

char const* last_error() { return last_error_.c_str(); }

// ---- slots 13.. (batch additions) --------------------------------------------------------------

dlimg_Result process_images_for_segmentation(dlimg_Segmentation* out, dlimg_ImageView const* images, int count,
                                             dlimg_Environment env) {
    return guarded([&] {
        DLIMG_ASSERT(out != nullptr && images != nullptr && env != nullptr && count >= 0);
        for (int i = 0; i < count; ++i) out[i] = nullptr;
        std::vector<SegmentationImpl*> segs(count);
        for (int i = 0; i < count; ++i) {
            segs[i] = new SegmentationImpl(impl(env));
            out[i] = reinterpret_cast<dlimg_Segmentation>(segs[i]);
        }
        SegmentationImpl::process_batch(impl(env), segs.data(), images, count);
    });
}

dlimg_Result get_segmentation_masks(dlimg_Segmentation const* segs, int count, int const* points, int const* regions,
                                    uint8_t** out_masks) {
    return guarded([&] {
        DLIMG_ASSERT(segs != nullptr && out_masks != nullptr && count >= 0);
        std::vector<SegmentationImpl const*> s(count);
        for (int i = 0; i < count; ++i) {
            DLIMG_ASSERT(segs[i] != nullptr);
            s[i] = &impl(segs[i]);
        }
        SegmentationImpl::compute_mask_batch(s.data(), count, points, regions, out_masks);
    });
}

const dlimg_Api api_table = {
    is_backend_supported,
    create_environment,
    destroy_environment,
    process_image_for_segmentation,
    get_segmentation_mask,
    get_segmentation_extent,
    destroy_segmentation,
    segment_objects,
    load_image,
    save_image,
    create_image,
    destroy_image,
    last_error,
    process_images_for_segmentation,
    get_segmentation_masks,
};

}  // namespace

// used by the extension entry points (ext_api.cpp) to report through the same channel
dlimg_Result report_error(char const* what) noexcept { return fail(what); }

}  // namespace dlimg

extern "C" DLIMG_API dlimg_Api const* dlimg_init(void) { return &dlimg::api_table; }

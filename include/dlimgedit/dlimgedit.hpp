// dlimgedit.hpp -- C++ convenience layer over the C-ABI of the MI355X build (single header).
//
// Source-compatible with the consumer-facing API of the reference's wrapper
// (reference: src/include/dlimgedit/dlimgedit.hpp:16-191): namespace dlimg, Extent / Channels /
// ImageView / Image, Backend / Options / Environment, Point / Region / Segmentation (process,
// compute_mask, compute_masks, extent), segment_objects, initialize, Exception.  Code written
// against the reference's header compiles against this one; binaries built against the reference's
// header need no rebuild at all because the C table underneath is layout-identical.
//
// Additions: Segmentation::process_batch and Segmentation::compute_mask_batch (batched entry
// points of this build).  Requires C++17.  Define DLIMGEDIT_LOAD_DYNAMIC before including to bind
// the library at run time: dlsym "dlimg_init" yourself and pass the result to dlimg::initialize().
#pragma once

#include "dlimgedit.h"

#include <array>
#include <cstddef>
#include <exception>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#ifndef DLIMGEDIT_NO_FILESYSTEM
#    include <filesystem>
#endif

namespace dlimg {

// ---------------------------------------------------------------------------------------------
// binding to the C table

namespace detail {
inline dlimg_Api const*& table() {
    static dlimg_Api const* t = nullptr;
    return t;
}
}  // namespace detail

#ifdef DLIMGEDIT_LOAD_DYNAMIC
inline void initialize(dlimg_Api const* api) { detail::table() = api; }
#else
inline void initialize(dlimg_Api const* api = dlimg_init()) { detail::table() = api; }
#endif

inline dlimg_Api const& api() {
#ifndef DLIMGEDIT_LOAD_DYNAMIC
    if (!detail::table()) detail::table() = dlimg_init();
#endif
    return *detail::table();
}

class Exception : public std::exception {
  public:
    explicit Exception(std::string message) : message_(std::move(message)) {}
    char const* what() const noexcept override { return message_.c_str(); }

  private:
    std::string message_;
};

namespace detail {
inline void check(dlimg_Result r) {
    if (r != dlimg_success) throw Exception(api().last_error());
}
}  // namespace detail

// ---------------------------------------------------------------------------------------------
// images

struct Extent {
    int width = 0;
    int height = 0;
};
constexpr bool operator==(Extent a, Extent b) { return a.width == b.width && a.height == b.height; }
constexpr bool operator!=(Extent a, Extent b) { return !(a == b); }

// One byte per channel; bgra and argb are four-byte pixels.
enum class Channels { mask = 1, rgb = 3, rgba = 4, bgra, argb };
constexpr int count(Channels c) { return static_cast<int>(c) > 4 ? 4 : static_cast<int>(c); }

class Image;

// Borrowed pixels, row-major, origin top-left.  Same 24-byte layout as dlimg_ImageView.
struct ImageView {
    Extent extent;
    Channels channels = Channels::rgba;
    int stride = 0;  // bytes per row
    uint8_t const* pixels = nullptr;

    ImageView() noexcept = default;
    ImageView(uint8_t const* data, Extent e, Channels c = Channels::rgba) noexcept
        : extent(e), channels(c), stride(e.width * count(c)), pixels(data) {}
    ImageView(Image const& image) noexcept;
};
static_assert(sizeof(ImageView) == sizeof(dlimg_ImageView), "ImageView must mirror dlimg_ImageView");

// Owning, tightly packed pixels allocated by the library.
class Image {
  public:
    explicit Image(Extent e, Channels c = Channels::rgba)
        : extent_(e), channels_(c), pixels_(api().create_image(e.width, e.height, count(c))) {}
    Image(Image&& o) noexcept : extent_(o.extent_), channels_(o.channels_), pixels_(std::exchange(o.pixels_, nullptr)) {}
    Image& operator=(Image&& o) noexcept {
        std::swap(extent_, o.extent_);
        std::swap(channels_, o.channels_);
        std::swap(pixels_, o.pixels_);
        return *this;
    }
    Image(Image const&) = delete;
    Image& operator=(Image const&) = delete;
    ~Image() {
        if (pixels_) api().destroy_image(pixels_);
    }

    Extent extent() const noexcept { return extent_; }
    Channels channels() const noexcept { return channels_; }
    uint8_t* pixels() noexcept { return pixels_; }
    uint8_t const* pixels() const noexcept { return pixels_; }
    size_t size() const noexcept { return size_t(extent_.width) * extent_.height * count(channels_); }

    static Image load(char const* filepath) {
        uint8_t* px = nullptr;
        Extent e;
        int c = 0;
        detail::check(api().load_image(filepath, &e.width, &c, &px));
        return Image(e, static_cast<Channels>(c), px);
    }
    static void save(ImageView const& image, char const* filepath) {
        detail::check(api().save_image(reinterpret_cast<dlimg_ImageView const*>(&image), filepath));
    }
#ifndef DLIMGEDIT_NO_FILESYSTEM
    static Image load(std::filesystem::path const& p) { return load(p.string().c_str()); }
    static void save(ImageView const& image, std::filesystem::path const& p) { save(image, p.string().c_str()); }
#endif

  private:
    Image(Extent e, Channels c, uint8_t* adopted) : extent_(e), channels_(c), pixels_(adopted) {}
    Extent extent_;
    Channels channels_;
    uint8_t* pixels_ = nullptr;
};

inline ImageView::ImageView(Image const& image) noexcept
    : extent(image.extent()), channels(image.channels()), stride(extent.width * count(channels)), pixels(image.pixels()) {}

// ---------------------------------------------------------------------------------------------
// environment

enum class Backend { cpu, gpu };   // gpu = HIP device (MI355X); cpu is not available in this build

struct Options {
    Backend backend = Backend::cpu;
    char const* model_directory = "models";   // holds segmentation/sam_<variant>.dlw
};
static_assert(sizeof(Options) == sizeof(dlimg_Options), "Options must mirror dlimg_Options");

namespace detail {
struct EnvDeleter { void operator()(dlimg_Environment_* h) const { api().destroy_environment(h); } };
struct SegDeleter { void operator()(dlimg_Segmentation_* h) const { api().destroy_segmentation(h); } };
}  // namespace detail

// Owns the model cache; thread-safe; must outlive every Segmentation created from it.
class Environment {
  public:
    static bool is_supported(Backend b) noexcept { return api().is_backend_supported(dlimg_Backend(int(b))) != 0; }

    explicit Environment(Options const& options = {}) {
        dlimg_Environment h = nullptr;
        detail::check(api().create_environment(&h, reinterpret_cast<dlimg_Options const*>(&options)));
        handle_.reset(h);
    }
    Environment(std::nullptr_t) noexcept {}

    dlimg_Environment handle() const noexcept { return handle_.get(); }
    explicit operator bool() const noexcept { return bool(handle_); }

  private:
    std::unique_ptr<dlimg_Environment_, detail::EnvDeleter> handle_;
};

// ---------------------------------------------------------------------------------------------
// segmentation

struct Point {
    int x = 0;
    int y = 0;
};
constexpr bool operator==(Point a, Point b) { return a.x == b.x && a.y == b.y; }
constexpr bool operator!=(Point a, Point b) { return !(a == b); }

struct Region {
    Point top_left;
    Point bottom_right;

    constexpr Region() = default;
    constexpr Region(Point tl, Point br) : top_left(tl), bottom_right(br) {}
    constexpr Region(Point origin, Extent e) : top_left(origin), bottom_right{origin.x + e.width, origin.y + e.height} {}
    constexpr Extent extent() const { return Extent{bottom_right.x - top_left.x, bottom_right.y - top_left.y}; }
};
constexpr bool operator==(Region a, Region b) { return a.top_left == b.top_left && a.bottom_right == b.bottom_right; }
constexpr bool operator!=(Region a, Region b) { return !(a == b); }

// An encoded image (embedding resident in HBM) that answers mask queries cheaply.
class Segmentation {
  public:
    struct Mask {
        Image image;            // Channels::mask, values 0 / 255
        float accuracy = 0.0f;  // predicted IoU
    };

    static Segmentation process(ImageView const& image, Environment const& env) {
        Segmentation s(nullptr);
        dlimg_Segmentation h = nullptr;
        dlimg_Result r = api().process_image_for_segmentation(&h, reinterpret_cast<dlimg_ImageView const*>(&image), env.handle());
        s.handle_.reset(h);     // owned even when encoding failed
        detail::check(r);
        return s;
    }

    // Encodes several independent images in one batched pass (addition of this build).
    static std::vector<Segmentation> process_batch(std::vector<ImageView> const& images, Environment const& env) {
        std::vector<dlimg_Segmentation> hs(images.size(), nullptr);
        dlimg_Result r = api().process_images_for_segmentation(
            hs.data(), reinterpret_cast<dlimg_ImageView const*>(images.data()), int(images.size()), env.handle());
        std::vector<Segmentation> out;
        out.reserve(hs.size());
        for (auto h : hs) {
            out.emplace_back(nullptr);
            out.back().handle_.reset(h);
        }
        detail::check(r);
        return out;
    }

    void compute_mask(Point p, uint8_t* out_mask) const { query(&p.x, nullptr, out_mask); }
    void compute_mask(Region r, uint8_t* out_mask) const { query(nullptr, &r.top_left.x, out_mask); }
    Image compute_mask(Point p) const {
        Image m(extent(), Channels::mask);
        compute_mask(p, m.pixels());
        return m;
    }
    Image compute_mask(Region r) const {
        Image m(extent(), Channels::mask);
        compute_mask(r, m.pixels());
        return m;
    }

    // Three candidate masks for an ambiguous point, with their predicted accuracy.
    std::array<Mask, 3> compute_masks(Point p) const {
        std::array<Mask, 3> out{Mask{Image(extent(), Channels::mask)}, Mask{Image(extent(), Channels::mask)},
                                Mask{Image(extent(), Channels::mask)}};
        uint8_t* ptrs[3] = {out[0].image.pixels(), out[1].image.pixels(), out[2].image.pixels()};
        float acc[3] = {0, 0, 0};
        detail::check(api().get_segmentation_mask(handle_.get(), &p.x, nullptr, ptrs, acc));
        for (int i = 0; i < 3; ++i) out[i].accuracy = acc[i];
        return out;
    }

    // One point query per segmentation, decoded as one batch (addition of this build).
    static std::vector<Image> compute_mask_batch(std::vector<Segmentation const*> const& segs, std::vector<Point> const& points) {
        std::vector<dlimg_Segmentation> hs;
        std::vector<Image> out;
        std::vector<uint8_t*> ptrs;
        for (auto* s : segs) {
            hs.push_back(s->handle_.get());
            out.emplace_back(s->extent(), Channels::mask);
            ptrs.push_back(out.back().pixels());
        }
        detail::check(api().get_segmentation_masks(hs.data(), int(hs.size()), &points.data()->x, nullptr, ptrs.data()));
        return out;
    }

    Extent extent() const noexcept {
        Extent e;
        api().get_segmentation_extent(handle_.get(), &e.width);
        return e;
    }

    Segmentation(std::nullptr_t) noexcept {}
    dlimg_Segmentation handle() const noexcept { return handle_.get(); }
    explicit operator bool() const noexcept { return bool(handle_); }

  private:
    void query(int const* point, int const* region, uint8_t* out_mask) const {
        uint8_t* ptrs[3] = {out_mask, nullptr, nullptr};   // second slot null selects single-mask mode
        float acc[3] = {0, 0, 0};
        detail::check(api().get_segmentation_mask(handle_.get(), point, region, ptrs, acc));
    }
    std::unique_ptr<dlimg_Segmentation_, detail::SegDeleter> handle_;
};
static_assert(sizeof(Point) == 8 && sizeof(Region) == 16 && sizeof(Extent) == 8, "POD layouts of the C-ABI");

// Dichotomous foreground segmentation (BiRefNet in the reference): reported as unsupported by this build.
inline void segment_objects(ImageView const& image, uint8_t* out_mask, Environment const& env) {
    detail::check(api().segment_objects(reinterpret_cast<dlimg_ImageView const*>(&image), out_mask, env.handle()));
}
inline Image segment_objects(ImageView const& image, Environment const& env) {
    Image m(image.extent, Channels::mask);
    segment_objects(image, m.pixels(), env);
    return m;
}

}  // namespace dlimg

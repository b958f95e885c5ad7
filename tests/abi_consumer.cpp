// Drop-in proof: a consumer written against the REFERENCE's public C++ wrapper
// (<dlimgedit/dlimgedit.hpp> from the upstream source tree, header-only, DLIMGEDIT_LOAD_DYNAMIC)
// that loads THIS repository's libdlimgedit.so at run time.  Built by oracle/build_ref.py with
// -I/root/reference/src/include into oracle/_ref/abi_consumer; nothing of the reference is copied.
//
//   abi_consumer <libdlimgedit.so> probe
//   abi_consumer <libdlimgedit.so> run <model_dir> <rgba.raw> <w> <h> <px> <py> <out_mask.raw>
//   abi_consumer <libdlimgedit.so> loop <model_dir> <rgba.raw> <w> <h> <px> <py> <seconds> [view]
//       the consumer's natural loop -- process(image), compute_mask(point), one thread -- for <seconds>; prints its rate.
//       The pixels are held in a dlimg::Image, as after Image::load ("view": in the program's own buffer instead)
#define DLIMGEDIT_LOAD_DYNAMIC
#include <dlimgedit/dlimgedit.hpp>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

static int fail(const char* what) {
    std::fprintf(stderr, "abi_consumer: %s\n", what);
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 3) return fail("usage: abi_consumer <lib> probe|run ...");
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(dlerror());
    using InitFn = dlimg_Api const* (*)();
    auto init = reinterpret_cast<InitFn>(dlsym(lib, "dlimg_init"));
    if (!init) return fail("dlimg_init not exported");
    dlimg::initialize(init());

    std::string mode = argv[2];
    if (mode == "probe") {
        std::printf("cpu=%d gpu=%d\n", int(dlimg::Environment::is_supported(dlimg::Backend::cpu)),
                    int(dlimg::Environment::is_supported(dlimg::Backend::gpu)));
        {
            dlimg::Image img(dlimg::Extent{8, 6}, dlimg::Channels::bgra);
            std::memset(img.pixels(), 7, img.size());
            std::printf("image size=%zu\n", img.size());
        }
        try {
            dlimg::Options opts;
            opts.backend = dlimg::Backend::gpu;
            opts.model_directory = "/definitely/not/here";
            dlimg::Environment env(opts);
            return fail("expected an exception for a missing model directory");
        } catch (dlimg::Exception const& e) {
            std::printf("error=%s\n", e.what());
        }
        return 0;
    }
    if (mode == "run") {
        if (argc != 10) return fail("run needs <model_dir> <rgba.raw> <w> <h> <px> <py> <out_mask.raw>");
        const int w = std::atoi(argv[5]), h = std::atoi(argv[6]);
        std::vector<uint8_t> pixels(size_t(w) * h * 4);
        std::ifstream in(argv[4], std::ios::binary);
        if (!in.read(reinterpret_cast<char*>(pixels.data()), std::streamsize(pixels.size()))) return fail("short image file");
        try {
            dlimg::Options opts;
            opts.backend = dlimg::Backend::gpu;
            opts.model_directory = argv[3];
            dlimg::Environment env(opts);
            auto view = dlimg::ImageView(pixels.data(), dlimg::Extent{w, h}, dlimg::Channels::rgba);
            auto seg = dlimg::Segmentation::process(view, env);
            std::printf("extent=%dx%d\n", seg.extent().width, seg.extent().height);
            dlimg::Image mask = seg.compute_mask(dlimg::Point{std::atoi(argv[7]), std::atoi(argv[8])});
            auto masks = seg.compute_masks(dlimg::Point{std::atoi(argv[7]), std::atoi(argv[8])});
            std::printf("accuracy=%.6f %.6f %.6f\n", masks[0].accuracy, masks[1].accuracy, masks[2].accuracy);
            std::ofstream out(argv[9], std::ios::binary);
            out.write(reinterpret_cast<const char*>(mask.pixels()), std::streamsize(mask.size()));
        } catch (dlimg::Exception const& e) {
            std::fprintf(stderr, "dlimg::Exception: %s\n", e.what());
            return 3;
        }
        return 0;
    }
    if (mode == "loop") {
        if (argc != 10 && argc != 11) return fail("loop needs <model_dir> <rgba.raw> <w> <h> <px> <py> <seconds> [view]");
        const int w = std::atoi(argv[5]), h = std::atoi(argv[6]);
        const dlimg::Point point{std::atoi(argv[7]), std::atoi(argv[8])};
        const double seconds = std::atof(argv[9]);
        std::vector<uint8_t> pixels(size_t(w) * h * 4);
        std::ifstream in(argv[4], std::ios::binary);
        if (!in.read(reinterpret_cast<char*>(pixels.data()), std::streamsize(pixels.size()))) return fail("short image file");
        try {
            dlimg::Options opts;
            opts.backend = dlimg::Backend::gpu;
            opts.model_directory = argv[3];
            dlimg::Environment env(opts);
            // the pixels as a consumer holds them after Image::load: in an Image (the raw file stands in for a PNG so that
            // the test needs no encoder); argv[10] == "view" keeps them in the program's own buffer instead
            dlimg::Image image(dlimg::Extent{w, h}, dlimg::Channels::rgba);
            std::memcpy(image.pixels(), pixels.data(), pixels.size());
            const bool own_buffer = argc == 11 && std::string(argv[10]) == "view";
            auto view = own_buffer ? dlimg::ImageView(pixels.data(), dlimg::Extent{w, h}, dlimg::Channels::rgba) : dlimg::ImageView(image);
            using clock = std::chrono::steady_clock;
            auto elapsed = [](clock::time_point a) { return std::chrono::duration<double>(clock::now() - a).count(); };
            size_t set_pixels = 0;
            for (int i = 0; i < 5; ++i) {               // model load, workspaces, first launches
                dlimg::Image mask = dlimg::Segmentation::process(view, env).compute_mask(point);
                set_pixels = 0;
                for (size_t j = 0; j < mask.size(); ++j) set_pixels += mask.pixels()[j] != 0;
            }
            long images = 0;
            const auto t0 = clock::now();
            while (elapsed(t0) < seconds) {
                auto seg = dlimg::Segmentation::process(view, env);
                dlimg::Image mask = seg.compute_mask(point);
                ++images;
            }
            std::printf("images_per_s=%.2f images=%ld set_pixels=%zu\n", double(images) / elapsed(t0), images, set_pixels);
        } catch (dlimg::Exception const& e) {
            std::fprintf(stderr, "dlimg::Exception: %s\n", e.what());
            return 3;
        }
        return 0;
    }
    return fail("unknown mode");
}

#include "environment.hpp"

#include <cstdlib>
#include <cstring>

namespace dlimg {

void throw_error(const char* msg) { throw Exception(msg); }

void assertion_failed(const char* file, int line, const char* expr) {
    std::string msg = std::string("Assertion failed at ") + file + ":" + std::to_string(line) + ": " + expr;
    std::fprintf(stderr, "%s\n", msg.c_str());
    throw Exception(msg);
}

void hip_failed(const char* file, int line, const char* expr, hipError_t err) {
    throw Exception(std::string("HIP error '") + hipGetErrorString(err) + "' at " + file + ":" +
                    std::to_string(line) + " in " + expr);
}

int EnvironmentImpl::device_count() noexcept {
    static const int count = [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        return n;
    }();
    return count;
}

bool EnvironmentImpl::is_supported(dlimg_Backend backend) noexcept {
    if (backend != dlimg_gpu) return false;   // no CPU execution path in this build
    static const bool ok = [] {
        if (device_count() <= 0) return false;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return false;
        // kernels are built for gfx950 only
        return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }();
    return ok;
}

EnvironmentImpl::EnvironmentImpl(dlimg_Options const& options) : backend(options.backend) {
    namespace fs = std::filesystem;
    const char* dir = options.model_directory ? options.model_directory : "models";
    std::error_code ec;
    fs::path p = fs::absolute(dir, ec);
    if (ec || !fs::exists(p)) throw Exception(std::string("Model path ") + dir + " does not exist");
    if (!fs::is_directory(p)) throw Exception(std::string("Model path ") + dir + " is not a directory");
    model_directory = p;
    if (backend != dlimg_gpu)
        throw Exception("The CPU backend is not available in the MI355X build of dlimgedit; use Backend::gpu");
    if (!is_supported(dlimg_gpu)) throw Exception("No supported GPU (gfx950) found for Backend::gpu");
    if (const char* dev = std::getenv("DLIMGEDIT_DEVICE")) device = std::atoi(dev);
    if (device < 0 || device >= device_count())
        throw Exception("DLIMGEDIT_DEVICE=" + std::to_string(device) + " is out of range");
}

std::string EnvironmentImpl::find_sam_weights() const {
    namespace fs = std::filesystem;
    const fs::path dir = model_directory / "segmentation";
    auto missing = [&](std::string const& name) {
        return Exception("Could not find model 'segmentation/" + name + "' in directory '" + model_directory.string() +
                         "'.");
    };
    if (const char* forced = std::getenv("DLIMGEDIT_SAM_MODEL")) {
        std::string name = std::string("sam_") + forced + ".dlw";
        if (!fs::exists(dir / name)) throw missing(name);
        return (dir / name).string();
    }
    for (const char* v : {"vit_b", "vit_l", "vit_h"}) {
        fs::path p = dir / (std::string("sam_") + v + ".dlw");
        if (fs::exists(p)) return p.string();
    }
    std::error_code ec;
    std::string best;
    if (fs::is_directory(dir, ec)) {
        for (auto const& e : fs::directory_iterator(dir, ec)) {
            std::string n = e.path().filename().string();
            if (n.rfind("sam_", 0) == 0 && e.path().extension() == ".dlw" && (best.empty() || n < best)) best = n;
        }
    }
    if (best.empty()) throw missing("sam_vit_b.dlw");
    return (dir / best).string();
}

EnvironmentImpl::SamLanes::SamLanes(std::string const& weight_path, int device, int count)
    : weights(std::make_shared<SamWeights>(weight_path, device)) {
    // images in flight per GPU, measured on MI355X: ViT-B 3 -> 4 lanes +5 % (5-8 lanes worse); ViT-H 4 lanes -6 %
    // against 3 (its kernels already cover the chip)
    if (count <= 0) count = weights->geom_.embed_dim <= 768 ? 4 : 3;
    for (int i = 0; i < count; ++i) lanes.push_back(std::make_unique<SamModel>(weights, i));
    k::gemm_set_shared_gpu(count > 1);
}

EnvironmentImpl::SamLanes& EnvironmentImpl::lanes() {
    return sam_.get_or_make([&] {
        int n = 0;      // chosen from the model size (SamLanes); DLIMGEDIT_LANES overrides (1..8)
        if (const char* e = std::getenv("DLIMGEDIT_LANES")) {
            n = std::atoi(e);
            n = n < 1 ? 1 : (n > 8 ? 8 : n);
        }
        return std::make_tuple(find_sam_weights(), device, n);
    });
}

int EnvironmentImpl::lane_count() { return int(lanes().lanes.size()); }

SamModel& EnvironmentImpl::lane(int index) { return *lanes().lanes.at(index); }

SamModel& EnvironmentImpl::sam_model() {
    SamLanes& l = lanes();
    if (single_lane_.load() || l.lanes.size() == 1) return *l.lanes[0];
    return *l.lanes[next_lane_.fetch_add(1) % l.lanes.size()];
}

}  // namespace dlimg

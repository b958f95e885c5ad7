#include "environment.hpp"
#include "image_memory.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <sched.h>

#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <system_error>

namespace dlimg {

namespace {
// The HIP runtime gives a process four hardware queues by default and multiplexes every stream onto them; the execution
// lanes (four per GPU) want a queue each, beside the host's own streams.  The runtime reads GPU_MAX_HW_QUEUES when IT
// initialises (its first API call), so the one place a library can still ask is its own load: if the host has not chosen,
// ask for eight.  What the library must NOT do is believe its own request: if the host touched HIP before loading this
// library (torch, any HIP host) the runtime has already read the default, and the variable now says 8 while the process has
// 4 queues.  So the load-time constructor records WHO set the variable and whether the HIP runtime had demonstrably not
// been initialised at that moment (it opens /dev/kfd when it initialises; no descriptor of this process pointing there =
// not initialised).  sam_model.cpp picks its stream layout from hardware_queues_trusted().
// Side effects on the host process: one setenv at load time -- inherited by child processes,
// and, like every setenv, not safe against a concurrent getenv in another thread of a host that loads the library late.
// DLIMGEDIT_KEEP_HW_QUEUES=1: hands off, nothing is written.
bool kfd_is_open() {
    std::error_code ec;
    for (auto const& e : std::filesystem::directory_iterator("/proc/self/fd", ec)) {
        std::error_code ec2;
        const auto target = std::filesystem::read_symlink(e.path(), ec2);
        if (!ec2 && target == "/dev/kfd") return true;
    }
    return false;
}

enum class QueueRequest { host_chose, library_asked_in_time, library_asked_too_late, hands_off };
const QueueRequest g_queue_request = [] {
    const char* keep = std::getenv("DLIMGEDIT_KEEP_HW_QUEUES");
    if (keep && std::atoi(keep) != 0) return QueueRequest::hands_off;
    if (std::getenv("GPU_MAX_HW_QUEUES")) return QueueRequest::host_chose;
    const bool hip_up = kfd_is_open();
    if (setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0) != 0) return QueueRequest::hands_off;
    return hip_up ? QueueRequest::library_asked_too_late : QueueRequest::library_asked_in_time;
}();
}  // namespace

bool hardware_queues_trusted() {
    switch (g_queue_request) {
    case QueueRequest::library_asked_in_time: return true;      // the runtime will read the 8 this library wrote
    case QueueRequest::library_asked_too_late: return false;    // the variable says 8, the runtime read its default
    case QueueRequest::host_chose:
    case QueueRequest::hands_off: {
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");       // the host's own number (it knows when it set it)
        return q && std::atoi(q) >= 8;
    }
    }
    return false;
}

std::vector<int> parse_cpu_list(std::string const& text) {
    std::vector<int> cpus;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t end = text.find(',', pos);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(pos, end - pos);
        pos = end + 1;
        char* tail = nullptr;
        const long a = std::strtol(item.c_str(), &tail, 10);
        if (tail == item.c_str() || a < 0) continue;
        long b = a;
        if (*tail == '-') {
            char* tail2 = nullptr;
            b = std::strtol(tail + 1, &tail2, 10);
            if (tail2 == tail + 1 || b < a) continue;
        }
        for (long c = a; c <= b && cpus.size() < 4096; ++c) cpus.push_back((int)c);
    }
    return cpus;
}

int bind_thread_near_device(int device) noexcept {
    try {
        if (const char* e = std::getenv("DLIMGEDIT_NUMA_AFFINITY"))
            if (std::atoi(e) == 0) return 0;
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        std::string id(bus);
        for (auto& c : id) c = (char)std::tolower((unsigned char)c);
        auto read_line = [](std::string const& path) {
            std::string line;
            if (FILE* f = std::fopen(path.c_str(), "r")) {
                char buf[4096];
                if (std::fgets(buf, sizeof(buf), f)) line = buf;
                std::fclose(f);
            }
            while (!line.empty() && (line.back() == '\n' || line.back() == ' ')) line.pop_back();
            return line;
        };
        const std::string node = read_line("/sys/bus/pci/devices/" + id + "/numa_node");
        if (node.empty() || std::atoi(node.c_str()) < 0) return 0;
        const std::vector<int> cpus = parse_cpu_list(read_line("/sys/devices/system/node/node" + node + "/cpulist"));
        if (cpus.empty()) return 0;
        // the CPUs the PROCESS may use, captured the first time any thread asks (a thread that was bound near one GPU before
        // must not intersect the next GPU's node with its own narrowed mask)
        static cpu_set_t process_allowed;
        static const bool have_allowed = [] {
            CPU_ZERO(&process_allowed);
            return sched_getaffinity(0, sizeof(process_allowed), &process_allowed) == 0;
        }();
        if (!have_allowed) return 0;
        cpu_set_t allowed = process_allowed, want;
        CPU_ZERO(&want);
        int n = 0;
        for (int c : cpus)
            if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
                CPU_SET(c, &want);
                ++n;
            }
        if (n == 0 || n == CPU_COUNT(&allowed)) return 0;          // nothing to gain (one node) or nothing allowed there
        return sched_setaffinity(0, sizeof(want), &want) == 0 ? n : 0;
    } catch (...) {
        return 0;
    }
}

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

// A consumer of the reference that default-constructs Options asks for Backend::cpu (the reference's default,
// /root/reference/src/include/dlimgedit/dlimgedit.hpp:91) and cannot be recompiled by whoever swaps the library in.  This
// build has no CPU execution path and will not pretend to: the request is refused -- unless the DEPLOYER says, outside the
// consumer, that such requests are to run on the GPU: DLIMGEDIT_CPU_REQUESTS_ON_GPU=1.  Read per call (tests switch it).
static bool cpu_requests_run_on_gpu() noexcept {
    const char* e = std::getenv("DLIMGEDIT_CPU_REQUESTS_ON_GPU");
    return e && std::atoi(e) != 0;
}

bool EnvironmentImpl::is_supported(dlimg_Backend backend) noexcept {
    if (backend != dlimg_gpu && !(backend == dlimg_cpu && cpu_requests_run_on_gpu())) return false;   // no CPU execution path in this build
    static const bool ok = [] {
        if (device_count() <= 0) return false;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return false;
        // kernels are built for gfx950 only
        return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }();
    return ok;
}

// One operation on the device's NULL stream before any stream of this library exists.  Measured (r05, tools/launch_mt.cpp
// and a probe of slot 4 from four threads): in a process whose first GPU work goes to non-blocking streams, four callers of
// compute_mask() on four lanes complete 3900 prompts/s together -- barely more than one caller's 3200 -- while in a process
// that has used the null stream first (any torch host does: its first tensor is filled there) they complete 9700.  What the
// runtime ties to that first use is not visible from here (queue count and runtime version were ruled out: GPU_MAX_HW_QUEUES
// 4 / 8, ROCm 7.0 / 7.2 behave alike); the operation costs ~20 us once per device and Environment.
// Once per device and PROCESS (r06: it used to run for every Environment and every listing of a device, and waited for the
// whole device -- a host with work in flight on other streams paid a device-wide synchronisation per Environment): the
// effect belongs to the process's first use of the null stream, so later environments have nothing to add, and the wait
// is for the null stream alone.
static void prime_null_stream(int device) {
    static std::mutex m;
    static std::vector<int> primed;
    std::lock_guard<std::mutex> lock(m);
    if (std::find(primed.begin(), primed.end(), device) != primed.end()) return;
    HIP_CHECK(hipSetDevice(device));
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, 4096));
    const hipError_t set = hipMemsetAsync(p, 0, 4096, nullptr);
    const hipError_t done = set == hipSuccess ? hipStreamSynchronize(nullptr) : set;
    (void)hipFree(p);
    HIP_CHECK(done);
    primed.push_back(device);
}

EnvironmentImpl::EnvironmentImpl(dlimg_Options const& options) : backend(options.backend) {
    namespace fs = std::filesystem;
    const char* dir = options.model_directory ? options.model_directory : "models";
    std::error_code ec;
    fs::path p = fs::absolute(dir, ec);
    if (ec || !fs::exists(p)) throw Exception(std::string("Model path ") + dir + " does not exist");
    if (!fs::is_directory(p)) throw Exception(std::string("Model path ") + dir + " is not a directory");
    model_directory = p;
    if (backend == dlimg_cpu && cpu_requests_run_on_gpu()) backend = dlimg_gpu;      // the deployer's decision, not the consumer's
    if (backend != dlimg_gpu)
        throw Exception("The CPU backend is not available in the MI355X build of dlimgedit; use Backend::gpu (a deployer who "
                        "cannot change the consumer sets DLIMGEDIT_CPU_REQUESTS_ON_GPU=1: Backend::cpu requests then run on the GPU)");
    if (!is_supported(dlimg_gpu)) throw Exception("No supported GPU (gfx950) found for Backend::gpu");
    // Device list: DLIMGEDIT_DEVICES ("0,1,2" or "all") wins over DLIMGEDIT_DEVICE (one index); default device 0.
    std::vector<int> devices;
    if (const char* list = std::getenv("DLIMGEDIT_DEVICES")) {
        std::string l(list);
        if (l == "all") {
            for (int i = 0; i < device_count(); ++i) devices.push_back(i);
        } else {
            size_t pos = 0;
            while (pos <= l.size()) {
                size_t end = l.find(',', pos);
                if (end == std::string::npos) end = l.size();
                std::string item = l.substr(pos, end - pos);
                char* tail = nullptr;
                long v = std::strtol(item.c_str(), &tail, 10);
                if (item.empty() || *tail) throw Exception("DLIMGEDIT_DEVICES='" + l + "' is not a list of device indices");
                devices.push_back(int(v));
                pos = end + 1;
            }
        }
    } else if (const char* dev = std::getenv("DLIMGEDIT_DEVICE")) {
        devices.push_back(std::atoi(dev));
    } else {
        devices.push_back(0);
    }
    // DLIMGEDIT_SINGLE_LANE=1: every request runs on lane 0 of its replica (the lanes still exist, so tile choices are
    // those of the shared-GPU configuration): a kernel trace taken this way shows each kernel alone on the chip, which
    // is what bench.py's per-kernel clocks measure
    if (const char* e = std::getenv("DLIMGEDIT_SINGLE_LANE")) forced_single_lane_ = std::atoi(e) != 0;
    if (const char* e = std::getenv("DLIMGEDIT_COALESCE")) {
        const int v = std::atoi(e);
        coalesce = v < 1 ? 1 : (v > 8 ? 8 : v);
    }
    if (const char* e = std::getenv("DLIMGEDIT_STEP_LANES")) {
        const int v = std::atoi(e);
        step_lanes = v < 0 ? 0 : (v > 8 ? 8 : v);
    }
    if (const char* e = std::getenv("DLIMGEDIT_STEP_WORKERS")) use_step_workers = std::atoi(e) != 0;
    if (const char* e = std::getenv("DLIMGEDIT_STEP_DEPTH")) {
        const int v = std::atoi(e);
        step_depth = v < 1 ? 1 : (v > 64 ? 64 : v);
    }
    single_lane_.store(forced_single_lane_);
    image_memory_use_pinned();       // images and masks the library allocates from now on can be reached by the GPU in place
    for (int d : devices) {
        if (d < 0 || d >= device_count())
            throw Exception("GPU index " + std::to_string(d) + " (DLIMGEDIT_DEVICE / DLIMGEDIT_DEVICES) is out of range: " +
                            std::to_string(device_count()) + " device(s) visible");
        prime_null_stream(d);
        auto r = std::make_unique<Replica>();
        r->device = d;
        r->pool = std::make_shared<EmbeddingPool>(d);
        replicas_.push_back(std::move(r));
    }
}

LaneWorker& EnvironmentImpl::lane_worker(int replica, int lane) {
    std::lock_guard<std::mutex> lock(workers_mutex_);
    if ((int)workers_.size() < replica_count()) workers_.resize(replica_count());
    auto& row = workers_.at(replica);
    if ((int)row.size() <= lane) row.resize(lane + 1);
    if (!row[lane]) {
        row[lane] = std::make_unique<LaneWorker>();
        // several GPUs in one environment: a lane's enqueue thread works next to its GPU (first task of the new thread)
        if (replica_count() > 1) {
            const int device = device_of(replica);
            row[lane]->post([device] { (void)bind_thread_near_device(device); });
        }
    }
    return *row[lane];
}

void EnvironmentImpl::drain_step_workers() {
    std::vector<LaneWorker*> all;
    {
        std::lock_guard<std::mutex> lock(workers_mutex_);
        if (!workers_.empty())
            for (auto& w : workers_[0])
                if (w) all.push_back(w.get());
    }
    for (LaneWorker* w : all) w->drain();
}

EnvironmentImpl::~EnvironmentImpl() {
    workers_.clear();            // finishes the passes already handed over, then joins: before the lanes go away
    // Requests accepted by dlimg_amd_encode_and_mask but never launched (a request still waiting for its coalescing
    // partner): the caller destroyed the environment without dlimg_amd_synchronize.  Their masks will not be written;
    // say so instead of dropping them silently (a destructor cannot return an error).
    if (!pending.empty())
        std::fprintf(stderr, "dlimgedit: environment destroyed with %zu queued request(s) that were never launched "
                             "(call dlimg_amd_synchronize first)\n", pending.size());
    for (auto& lane : step_passes)
        for (auto& pass : lane)
            if (pass.ticket && pass.ticket->done) (void)hipEventDestroy(pass.ticket->done);
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
    // the lanes of a GPU tell each other whether they have work in flight (sam_model.hpp, LaneBoard)
    auto board = count > 1 ? std::make_shared<LaneBoard>(device, count) : nullptr;
    for (int i = 0; i < count; ++i) lanes.push_back(std::make_unique<SamModel>(weights, i, count, board));
}

EnvironmentImpl::SamLanes& EnvironmentImpl::lanes(int replica) {
    Replica& r = *replicas_.at(replica);
    return r.sam.get_or_make([&] {
        int n = 0;      // chosen from the model size (SamLanes); DLIMGEDIT_LANES overrides (1..8)
        if (const char* e = std::getenv("DLIMGEDIT_LANES")) {
            n = std::atoi(e);
            n = n < 1 ? 1 : (n > 8 ? 8 : n);
        }
        return std::make_tuple(find_sam_weights(), r.device, n);
    });
}

int EnvironmentImpl::lane_count(int replica) { return int(lanes(replica).lanes.size()); }

SamModel& EnvironmentImpl::lane(int replica, int index) { return *lanes(replica).lanes.at(index); }

SamModel& EnvironmentImpl::next_lane(int replica) {
    SamLanes& l = lanes(replica);
    if (single_lane_.load() || l.lanes.size() == 1) return *l.lanes[0];
    return *l.lanes[replicas_[replica]->next_lane.fetch_add(1) % l.lanes.size()];
}

float* EmbeddingPool::take() {
    {
        std::lock_guard<std::mutex> lock(mutex_);
        if (!free_.empty()) {
            float* p = free_.back();
            free_.pop_back();
            return p;
        }
    }
    HIP_CHECK(hipSetDevice(device_));
    float* p = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), (size_t)kTokens * kEmbedDim * sizeof(float)));
    return p;
}

void EmbeddingPool::give(float* buffer) noexcept {
    if (!buffer) return;
    constexpr size_t kKeep = 1024;                // 4 GiB of the 288 GB at most stay parked
    {
        std::lock_guard<std::mutex> lock(mutex_);
        if (free_.size() < kKeep) {
            // Kernels that read or wrote the buffer were queued by calls that have returned, and every such call
            // waits for its work: nothing on the device refers to the buffer any more.
            free_.push_back(buffer);
            return;
        }
    }
    (void)hipSetDevice(device_);
    (void)hipFree(buffer);
}

EmbeddingPool::~EmbeddingPool() {
    if (free_.empty()) return;
    (void)hipSetDevice(device_);
    for (float* p : free_) (void)hipFree(p);
}

void EnvironmentImpl::load_all() {
    for (int r = 0; r < replica_count(); ++r) (void)lanes(r);
}

}  // namespace dlimg

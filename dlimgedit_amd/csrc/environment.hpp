// EnvironmentImpl: backend gate, model directory, lazily loaded SAM model on one or several GPUs.
// Counterpart of /root/reference/src/environment.{hpp,cpp}; the onnxruntime environment, provider
// probing through dlopen(libcuda) and the thread-count knob are replaced by a HIP device probe.
//
// Multi-GPU (nothing like it in the reference, which appends the CUDA provider with default options = device 0,
// /root/reference/src/session.cpp:63-66): an environment owns one REPLICA per entry of its device list
// (DLIMGEDIT_DEVICES=0,1,...,7 | all; default: the single device DLIMGEDIT_DEVICE or 0).  A replica is a full copy
// of the weights on that GPU plus its execution lanes.  Images are independent, so a batch is dealt image i ->
// replica i mod G and nothing is exchanged between GPUs; every Segmentation handle remembers the replica that holds
// its embedding and its mask queries run there.
#pragma once

#include "common.hpp"
#include "sam_model.hpp"
#include "lane_worker.hpp"

#include <dlimgedit/dlimgedit.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <filesystem>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace dlimg {

// Device buffers for image embeddings ([4096][256] fp32 = 4 MiB each) of one GPU, recycled: hipMalloc costs a fraction
// of a millisecond and hipFree waits for the whole device, which would serialise the lanes once per image.  Shared by
// the environment and its Segmentation handles, so a handle that is destroyed after its environment (the reference
// forbids it, dlimgedit.hpp:98-100, but a garbage-collected host cannot promise the order) still has a home for its buffer.
class EmbeddingPool {
  public:
    explicit EmbeddingPool(int device) : device_(device) {}
    ~EmbeddingPool();
    EmbeddingPool(EmbeddingPool const&) = delete;
    EmbeddingPool& operator=(EmbeddingPool const&) = delete;
    float* take();
    void give(float* buffer) noexcept;

  private:
    int device_;
    std::mutex mutex_;
    std::vector<float*> free_;
};

class EnvironmentImpl {
  public:
    dlimg_Backend backend = dlimg_gpu;
    std::filesystem::path model_directory;

    // Cheap, cached, never throws (reference: environment.cpp:103-122).
    static bool is_supported(dlimg_Backend backend) noexcept;
    static int device_count() noexcept;

    explicit EnvironmentImpl(dlimg_Options const& options);
    ~EnvironmentImpl();

    // Replicas: one per entry of the device list (the same GPU may be listed more than once; each entry is an
    // independent replica with its own weights -- used by the tests to exercise the multi-device code on one GPU).
    int replica_count() const { return int(replicas_.size()); }
    int device_of(int replica) const { return replicas_.at(replica)->device; }
    int first_device() const { return replicas_[0]->device; }

    // Weights + execution lanes of a replica, created on first use, once per environment (reference:
    // environment.cpp:144-146).  next_lane() hands out a replica's lanes round-robin; lane() addresses one.
    SamModel& next_lane(int replica);
    SamModel& lane(int replica, int index);
    int lane_count(int replica = 0);
    // lanes that requests are really spread over (1 while set_single_lane is on)
    int effective_lane_count(int replica = 0) { return single_lane_.load() ? 1 : lane_count(replica); }
    // Replica for the next independent image (round-robin over the device list).
    int next_replica() { return replicas_.size() == 1 ? 0 : int(next_replica_.fetch_add(1) % replicas_.size()); }
    // Loads the model on every replica (reports a missing weight file where the reference does).
    void load_all();
    // Device buffers for image embeddings, recycled per replica (see EmbeddingPool).
    std::shared_ptr<EmbeddingPool> embedding_pool(int replica) const { return replicas_.at(replica)->pool; }

    // The enqueue thread of one lane (LaneWorker), created on first use.  Used by the device-step queue and by batch
    // calls that spread several passes over the lanes (segmentation.cpp, process_batch).
    LaneWorker& lane_worker(int replica, int lane);

    // While set, every request goes to lane 0 of its replica (per-kernel clocks must not see other lanes' kernels).
    void set_single_lane(bool on) { single_lane_.store(on || forced_single_lane_); }

    // Asynchronous steps of the device-resident entry point (dlimg_amd_encode_and_mask) that have been accepted but not
    // launched yet: independent single-image requests are coalesced into ONE batched pass of `coalesce` images (dynamic
    // batching, as a serving host would do): with two images per pass the N = 768 GEMMs (patch, proj, fc2) reach 96
    // tiles of the 256 x 256 kernel instead of 96 of the 128 x 256 one.  Results are bit-identical to single-image
    // passes (kernels/gemm.hip, tile choice).  DLIMGEDIT_COALESCE = 1 switches it off; dlimg_amd_synchronize flushes.
    // process_images_for_segmentation calls in flight (any thread): the pass size adapts to it (segmentation.cpp)
    std::atomic<int> batch_calls_in_flight{0};
    // Passes are handed to the lane with the least work in flight, and at most `step_depth` passes wait on a lane's
    // stream; what arrives beyond that stays in `pending` until a lane has room (looked at on the next call) or until
    // dlimg_amd_synchronize deals it out so that every lane ends up with the same number of images: a burst that is not
    // a multiple of lanes x coalesce then finishes on all lanes together instead of leaving some idle at the end.
    // DLIMGEDIT_STEP_DEPTH (1..64, default 2).
    struct PendingStep { dlimg_ImageView view; int x, y; uint8_t* mask; };
    // A planned pass on its way: handed to the lane's worker (state 0), then on the lane's stream with `done` recorded
    // behind it (1), or failed while being enqueued (2: its requests are counted in dropped_steps)
    // overflow: the pass's own f16-range report (SamModel::last_pass_flag, taken under the lane's mutex right after the pass
    // was enqueued); read once `done` has been reached, by whoever retires the ticket -- nobody else looks at that flag
    struct StepTicket { std::atomic<int> state{0}; hipEvent_t done = nullptr; const volatile int* overflow = nullptr; };
    struct StepPass { std::shared_ptr<StepTicket> ticket; int images; };
    std::mutex pending_mutex;
    std::vector<PendingStep> pending;
    std::vector<std::deque<StepPass>> step_passes;     // per lane of replica 0, oldest first (pending_mutex)
    // Defaults (r05, MI355X, ViT-B, bursts of 8-100 requests): passes of FOUR images on THREE of the lanes.  A four-image
    // pass fills the chip by itself where a two-image pass does not (768 workgroups of global attention = three full
    // rounds of the 256 CUs instead of 1.5; 192-tile stream writers), so fewer passes have to share the chip to fill
    // it: 4 lanes x 2 images 866-886 images/s, 3 x 4 882-902 on the same boxes, 4 x 4 851-867 (five passes of a 20-request
    // block over four lanes end 2 / 1 / 1 / 1).  ViT-H: three, ViT-L: four per pass on two lanes (r06, ext_api.cpp step_queue_width).  step_lanes = 0: every lane.  DLIMGEDIT_COALESCE / DLIMGEDIT_STEP_LANES / DLIMGEDIT_STEP_DEPTH.
    int coalesce = 0;                                  // 0 = chosen from the model when its lanes are created
    int step_lanes = -1;                               // -1 = likewise
    int step_depth = 2;
    int step_cursor = 0;                               // lane after the one used last (pending_mutex)
    // The passes are enqueued by the lanes' own host threads (lane_worker below);
    // DLIMGEDIT_STEP_WORKERS=0: the calling thread enqueues them itself, as before r04
    bool use_step_workers = true;
    void drain_step_workers();                         // every pass handed to a worker is on its stream (or has failed)
    std::mutex step_error_mutex;                       // the two below: written by the workers
    std::string step_error;                            // first failure of a queued pass since the last synchronize (sticky)
    int dropped_steps = 0;                             // requests that failure took with it
    int overflowed_steps = 0;                          // requests whose encoder pass left the f16 range (their masks are not valid)

  private:
    struct SamLanes {
        SamLanes(std::string const& weight_path, int device, int count);
        std::shared_ptr<SamWeights const> weights;
        std::vector<std::unique_ptr<SamModel>> lanes;
    };
    struct Replica {
        int device = 0;
        Lazy<SamLanes> sam;
        std::atomic<unsigned> next_lane{0};
        std::shared_ptr<EmbeddingPool> pool;
    };
    SamLanes& lanes(int replica);
    std::string find_sam_weights() const;
    std::vector<std::unique_ptr<Replica>> replicas_;
    std::mutex workers_mutex_;
    std::vector<std::vector<std::unique_ptr<LaneWorker>>> workers_;      // [replica][lane] (workers_mutex_)
    std::atomic<unsigned> next_replica_{0};
    std::atomic<bool> single_lane_{false};
    bool forced_single_lane_ = false;     // DLIMGEDIT_SINGLE_LANE
};

}  // namespace dlimg

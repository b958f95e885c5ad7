#include "segmentation.hpp"

#include "roctx.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <future>
#include <thread>
#include <vector>

namespace dlimg {

int channel_bytes(int channels) { return channels > 4 ? 4 : channels; }

int scale_coord(int coord, float scale) { return int(float(coord) * scale + 0.5f); }

void ResizeLongestSide::set(Extent image) {
    original = image;
    scale = float(max_side_) / float(std::max(image.width, image.height));
    resized = image;
    if (scale != 1) resized = Extent{scale_coord(image.width, scale), scale_coord(image.height, scale)};
}

void pack_prompt(ResizeLongestSide const& rs, Point const* point, Region const* region, float coords[4],
                 float labels[2]) {
    DLIMG_ASSERT((point != nullptr) != (region != nullptr));
    auto set = [&](int index, Point p, int label) {
        Point t = rs.transform(p);
        coords[index * 2 + 0] = float(t.x);
        coords[index * 2 + 1] = float(t.y);
        labels[index] = float(label);
    };
    if (point) {
        set(0, *point, 1);
        set(1, Point{0, 0}, -1);     // padding point of the exported decoder graph
    } else {
        set(0, region->top_left, 2);
        set(1, region->bottom_right, 3);
    }
}

void check_image(dlimg_ImageView const& image) {
    if (!image.pixels) throw Exception("Image has no pixel data");
    if (image.width <= 0 || image.height <= 0) throw Exception("Image extent must be positive");
    const int c = image.channels;
    if (!(c == 1 || c == 3 || c == 4 || c == 5 || c == 6))
        throw Exception("Unsupported channel order [" + std::to_string(c) + "]");
    DLIMG_ASSERT(image.stride >= image.width * channel_bytes(c));
}

namespace {
constexpr const char* kOverflowMessage =
    "the image encoder produced non-finite values: an activation left the f16 range (65504) of this build's MFMA operands "
    "and residual stream -- the embedding is refused (no masks are computed from it)";

// process() of one image leaves the wait to the first call that needs the embedding (segmentation.hpp)
bool deferred_process() {
    const char* e = std::getenv("DLIMGEDIT_SYNC_PROCESS");       // read per call: a deployer's switch, and the tests'
    return !(e && std::atoi(e) != 0);
}
}  // namespace

SegmentationImpl::SegmentationImpl(EnvironmentImpl& env) : env_(env) {
    env.load_all();     // loads the model (and reports a missing weight file) at the same point as the reference
}

SegmentationImpl::~SegmentationImpl() {
    forget_pending();                            // the pass still writes the buffer that goes back to the pool
    if (pool_) pool_->give(embedding_);
}

std::shared_ptr<SamModel::DeferredPass> SegmentationImpl::pending() const {
    std::lock_guard<std::mutex> lock(pending_mutex_);
    return pending_;
}

void SegmentationImpl::settle() const {
    if (auto p = pending()) {
        std::exception_ptr failed;
        try {
            if (p->settle()) invalid_ = true;
        } catch (...) {
            invalid_ = true;                     // the wait itself failed: nothing is known about the embedding
            failed = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lock(pending_mutex_);
            if (pending_ == p) pending_.reset();
        }
        if (failed) std::rethrow_exception(failed);
    }
    if (invalid_) throw Exception(kOverflowMessage);
}

void SegmentationImpl::forget_pending() noexcept {
    try {
        if (auto p = pending()) p->settle();
    } catch (...) {
    }
    std::lock_guard<std::mutex> lock(pending_mutex_);
    pending_.reset();
    invalid_ = false;
}

float* SegmentationImpl::embedding_storage(int replica) {
    if (embedding_ && replica != replica_) {
        pool_->give(embedding_);
        embedding_ = nullptr;
    }
    replica_ = replica;
    if (!embedding_) {
        pool_ = env_.embedding_pool(replica);
        embedding_ = pool_->take();
    }
    return embedding_;
}

namespace {

// Brings one host image into slot `slot` of the model's patch matrix (resizing on the device when
// the longest side is not 1024; reference: ResizeLongestSide::resize, segmentation.cpp:60-70).
void stage_image(SamModel& model, int slot, int batch, dlimg_ImageView const& image, ResizeLongestSide const& rs) {
    if (rs.scale != 1) {
        model.upload_and_resize_image(slot, batch, image.pixels, image.width, image.height, image.stride,
                                      image.channels, rs.resized.width, rs.resized.height);
    } else {
        model.upload_image(slot, batch, image.pixels, image.width, image.height, image.stride, image.channels);
    }
}

// Runs body(replica) for every replica in `used`: inline when there is one; otherwise the calling thread takes the first
// replica itself and hands the others to helper threads of ITS OWN (staging copies and kernel launches of different GPUs
// then proceed side by side).  The helpers are kept per calling thread and reused from call to call (r06; until then every
// call created and joined G threads): a caller's helpers serve nobody else, so concurrent callers never wait for each
// other here, and a helper that serves GPU g is bound to the CPUs of that GPU's NUMA node the first time it does
// (environment.cpp, bind_thread_near_device; DLIMGEDIT_NUMA_AFFINITY=0 switches that off).  The first exception wins.
struct ReplicaHelpers {
    std::vector<std::unique_ptr<LaneWorker>> workers;       // [helper]: joined when the calling thread ends
    std::vector<int> bound_to;                              // device the helper's thread was last bound near (-1: none)
};
template <typename F> void for_each_replica(EnvironmentImpl& env, std::vector<int> const& used, F&& body) {
    if (used.size() == 1) {
        body(used[0]);
        return;
    }
    thread_local ReplicaHelpers helpers;
    const size_t n_help = used.size() - 1;
    while (helpers.workers.size() < n_help) {
        helpers.workers.push_back(std::make_unique<LaneWorker>());
        helpers.bound_to.push_back(-1);
    }
    struct Handed { std::promise<void> result; std::future<void> answer; };
    std::vector<std::shared_ptr<Handed>> handed;
    for (size_t t = 0; t < n_help; ++t) {
        auto h = std::make_shared<Handed>();
        h->answer = h->result.get_future();
        handed.push_back(h);
        const int replica = used[t + 1], device = env.device_of(replica);
        const bool bind = helpers.bound_to[t] != device;
        helpers.bound_to[t] = device;
        helpers.workers[t]->post([h, &body, replica, device, bind] {
            try {
                if (bind) bind_thread_near_device(device);
                body(replica);
                h->result.set_value();
            } catch (...) {
                h->result.set_exception(std::current_exception());
            }
        });
    }
    std::exception_ptr first;
    try {
        body(used[0]);
    } catch (...) {
        first = std::current_exception();
    }
    for (auto& h : handed) {                     // every task is waited for: they refer to the caller's frame
        try {
            h->answer.get();
        } catch (...) {
            if (!first) first = std::current_exception();
        }
    }
    if (first) std::rethrow_exception(first);
}

// overflow: the pass's report (SamModel::last_pass_flag), read once its event has been waited for
struct Waiting { SamModel* model; hipEvent_t done; const volatile int* overflow; };


// Error paths: a request that threw half-way may have queued kernels or copies that still write into buffers the
// caller is about to hand back (pooled embedding buffers, mask staging slots).  Everything queued on that lane runs
// to completion first; errors of the drain itself are not interesting any more.
void drain_lane(SamModel* model) noexcept {
    if (!model) return;
    try {
        std::lock_guard<std::mutex> lock(model->mutex());
        (void)hipSetDevice(model->device());
        model->synchronize();
    } catch (...) {
    }
}

void wait_all(std::vector<Waiting>& waiting) {
    std::exception_ptr first;
    for (auto& w : waiting) {
        try {
            w.model->wait_and_recycle(w.done);
            if (w.overflow && *w.overflow) {
                *const_cast<volatile int*>(w.overflow) = 0;      // reported here, once
                throw Exception(kOverflowMessage);
            }
        } catch (...) {
            if (!first) first = std::current_exception();
        }
    }
    waiting.clear();
    if (first) std::rethrow_exception(first);
}

}  // namespace

void SegmentationImpl::process(dlimg_ImageView const& image) {
    SegmentationImpl* self = this;
    process_batch(env_, &self, &image, 1);
}

// Independent images: image i goes to replica (GPU) r0 + i mod G and there to the next execution lane, as its own
// batch-1 pass -- upload, pre-processing and encoder of one image overlap those of the others on the lanes' streams,
// and the host copies of image i+1 run while image i is on the GPU.  Nothing is exchanged between GPUs.
void SegmentationImpl::process_batch(EnvironmentImpl& env, SegmentationImpl* const* segs, dlimg_ImageView const* images,
                                     int count) {
    if (count <= 0) return;
    for (int i = 0; i < count; ++i) {
        check_image(images[i]);
        segs[i]->forget_pending();               // a handle processed again: its earlier pass writes the same buffer
        segs[i]->image_size_.set(Extent{images[i].width, images[i].height});
    }
    const bool defer = count == 1 && deferred_process();
    const int G = env.replica_count();
    std::vector<int> replica_of(count), used;
    for (int i = 0; i < count; ++i) {
        replica_of[i] = env.next_replica();
        if (std::find(used.begin(), used.end(), replica_of[i]) == used.end()) used.push_back(replica_of[i]);
    }
    (void)G;
    static const bool trace = std::getenv("DLIMGEDIT_TIMING") != nullptr;     // diagnostic: host time of the two phases
    // Images per batched encoder pass.  Results do not depend on it (kernels/gemm.hip: the tiles a pass may use compute
    // the same bits); throughput does: every lane should get a pass, and passes of two or more images run the N = 768
    // GEMMs on 256 x 256 tiles (8 images through one host thread, ViT-B: 601 images/s as 2 x 4, 649 as 8 x 1, 674 as
    // 4 x 2).  Default: the GPU's share spread over its lanes, at most 4 per pass -- unless other threads have batch calls
    // in flight, which keep the other lanes busy anyway: then fewer, larger passes win (4 threads x 8 images: 685 images/s
    // with passes of 2, 736 with passes of 4).  DLIMGEDIT_ENCODE_BATCH overrides (1..16).
    static const int forced_chunk = [] {
        const char* e = std::getenv("DLIMGEDIT_ENCODE_BATCH");
        const int v = e ? std::atoi(e) : 0;
        return v < 0 ? 0 : (v > 16 ? 16 : v);
    }();
    struct InFlight {
        std::atomic<int>& n;
        int before;
        explicit InFlight(std::atomic<int>& c) : n(c), before(c.fetch_add(1)) {}
        ~InFlight() { n.fetch_sub(1); }
    } in_flight(env.batch_calls_in_flight);
    const bool alone = in_flight.before == 0;    // no other batch call is being worked on right now
    for_each_replica(env, used, [&](int replica) {
        HIP_CHECK(hipSetDevice(env.device_of(replica)));
        std::vector<Waiting> waiting;
        SamModel* enqueueing = nullptr;          // the lane whose request is being put together (error path: drained)
        const auto t0 = std::chrono::steady_clock::now();
        try {
            std::vector<int> mine;
            for (int i = 0; i < count; ++i)
                if (replica_of[i] == replica) mine.push_back(i);
            // chunks of up to `chunk` images: one batched pass per chunk on the next lane
            const size_t lanes = (size_t)std::max(1, env.effective_lane_count(replica));
            const size_t spread = std::max<size_t>(1, (mine.size() + lanes - 1) / lanes);     // one pass per lane
            const size_t chunk = forced_chunk ? (size_t)forced_chunk
                                 : alone      ? std::min<size_t>(4, spread)
                                              : std::min<size_t>(4, std::max(spread, (mine.size() + 1) / 2));
            // One pass: the chunk's images staged (host copy + upload) and encoded on `model`; returns the event behind it.
            struct Queued { hipEvent_t done; const volatile int* overflow; std::shared_ptr<SamModel::DeferredPass> deferred; };
            auto run_chunk = [&](SamModel& model, size_t base, int n) {
                std::vector<float*> emb(n);
                for (int j = 0; j < n; ++j) emb[j] = segs[mine[base + j]]->embedding_storage(replica);
                roctx::Range range("dlimg.process");
                std::lock_guard<std::mutex> lock(model.mutex());
                HIP_CHECK(hipSetDevice(model.device()));
                {
                    roctx::Range r("dlimg.pre");
                    for (int j = 0; j < n; ++j) {
                        const int i = mine[base + j];
                        stage_image(model, j, n, images[i], segs[i]->image_size_);
                    }
                }
                {
                    roctx::Range r("dlimg.encode");
                    model.encode(n, emb.data());
                }
                model.wait_caller_copies();       // pixels read in place (pinned image memory): theirs again when this returns
                if (defer) return Queued{nullptr, nullptr, model.defer_last_pass()};
                const volatile int* overflow = model.last_pass_flag();
                return Queued{model.completion(), overflow, nullptr};
            };
            const size_t chunks = (mine.size() + chunk - 1) / chunk;
            if (chunks > 1 && alone && env.use_step_workers) {
                // several passes and no other caller at work: each pass is put together by its lane's own host thread
                // (LaneWorker) -- packing and uploading the images and ~75 launches per pass take 0.5-1 ms of host time,
                // which one thread would spend lane after lane while the later lanes' part of the chip waits (8 images
                // from one thread: 718 -> 734 images/s).  With several callers the lanes are fed in parallel anyway and
                // the hand-over only costs (two threads x 8 images: 846 without, 824 with)
                // (the promise is shared with the task: it must outlive the task's set_value call, which may still be
                // returning when this thread has its answer)
                struct Handed { SamModel* model; std::promise<Queued> result; std::future<Queued> answer; };
                std::vector<std::shared_ptr<Handed>> handed;
                for (size_t base = 0; base < mine.size(); base += chunk) {
                    const int n = (int)std::min<size_t>(chunk, mine.size() - base);
                    SamModel& model = env.next_lane(replica);
                    auto h = std::make_shared<Handed>();
                    h->model = &model;
                    h->answer = h->result.get_future();
                    handed.push_back(h);
                    env.lane_worker(replica, model.lane_index()).post([h, &run_chunk, base, n] {
                        try {
                            h->result.set_value(run_chunk(*h->model, base, n));
                        } catch (...) {
                            h->result.set_exception(std::current_exception());
                        }
                    });
                }
                std::exception_ptr first;
                for (auto& h : handed) {             // every task is waited for: they refer to this frame
                    try {
                        const Queued q = h->answer.get();
                        waiting.push_back(Waiting{h->model, q.done, q.overflow});
                    } catch (...) {
                        drain_lane(h->model);        // whatever the failed pass queued runs to completion first
                        if (!first) first = std::current_exception();
                    }
                }
                if (first) std::rethrow_exception(first);
            } else {
                for (size_t base = 0; base < mine.size(); base += chunk) {
                    const int n = (int)std::min<size_t>(chunk, mine.size() - base);
                    SamModel& model = env.next_lane(replica);
                    enqueueing = &model;
                    const Queued q = run_chunk(model, base, n);
                    enqueueing = nullptr;
                    if (q.deferred) {
                        // the caller's thread goes back now; the first mask query is queued behind the pass and waits
                        std::lock_guard<std::mutex> lock(segs[mine[base]]->pending_mutex_);
                        segs[mine[base]]->pending_ = q.deferred;
                    } else {
                        waiting.push_back(Waiting{&model, q.done, q.overflow});
                    }
                }
            }
        } catch (...) {
            // the chunk that threw has no completion event: whatever it queued (it writes the handles' embedding
            // buffers, which their destructors return to the pool) is waited for on its stream
            drain_lane(enqueueing);
            try { wait_all(waiting); } catch (...) {}
            throw;
        }
        // process() is synchronous in the reference (Ort::Session::Run returns when the result is there).  A batch is
        // waited for here and its errors surface in this call; one image is left to its first query (segmentation.hpp).
        const auto t1 = std::chrono::steady_clock::now();
        wait_all(waiting);
        if (trace) {
            const auto t2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "process_batch replica %d: enqueue %.3f ms, wait %.3f ms\n", replica,
                         std::chrono::duration<double, std::milli>(t1 - t0).count(),
                         std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    });
}

void SegmentationImpl::compute_mask(Point const* point, Region const* region, uint8_t* const out_masks[3],
                                    float out_accuracy[3]) const {
    DLIMG_ASSERT(point || region);
    DLIMG_ASSERT(embedding_ != nullptr);
    if (invalid_) throw Exception(kOverflowMessage);
    float coords[4], labels[2];
    pack_prompt(image_size_, point, region, coords, labels);
    const bool is_single_mask = out_masks[1] == nullptr;
    if (is_single_mask) {
        DLIMG_ASSERT(out_masks[0] != nullptr);
    } else {
        for (int i = 0; i < 3; ++i) DLIMG_ASSERT(out_masks[i] != nullptr);
    }

    HIP_CHECK(hipSetDevice(env_.device_of(replica_)));
    // an encoder pass nobody has waited for yet: the decoder goes onto ITS lane, behind it in stream order, and the wait
    // for the masks is the wait for both (any number of threads may do so at once: the lane's mutex orders their requests)
    const std::shared_ptr<SamModel::DeferredPass> behind = pending();
    SamModel& model_ = behind ? *behind->lane : env_.next_lane(replica_);
    const Extent o = image_size_.original, r = image_size_.resized;
    k::PostJob jobs[3];
    int n_jobs = 0;
    float iou[4] = {0.f, 0.f, 0.f, 0.f};
    SamModel::MaskSlot& slot = model_.acquire_mask_slot();
    try {
        {
            roctx::Range range("dlimg.compute_mask");
            std::lock_guard<std::mutex> lock(model_.mutex());
            float const* emb = embedding_;
            {
                roctx::Range rd("dlimg.decode");
                model_.decode(&emb, coords, labels, 1);
            }
            if (is_single_mask) {
                // single-mask decoder: best of the four by SamOnnxModel.select_masks, chosen on the device
                jobs[n_jobs++] = k::PostJob{model_.logits(), model_.iou(), out_masks[0], o.width, o.height, r.width, r.height};
            } else {
                // multi-mask decoder: outputs 1..3 (reference: segmentation.cpp:167-172)
                for (int i = 0; i < 3; ++i)
                    jobs[n_jobs++] = k::PostJob{model_.logits() + (size_t)(i + 1) * kLowRes * kLowRes, nullptr, out_masks[i],
                                                o.width, o.height, r.width, r.height};
            }
            roctx::Range rp("dlimg.post");
            model_.enqueue_masks(slot, jobs, n_jobs, is_single_mask ? 0 : 4);
        }
        model_.finish_masks(slot, jobs, n_jobs, iou, is_single_mask ? 0 : 4);      // waits outside the lane's mutex
    } catch (...) {
        drain_lane(&model_);                     // a copy into the slot's staging memory may still be queued
        model_.release_mask_slot(slot);
        throw;
    }
    model_.release_mask_slot(slot);
    if (behind) settle();                        // has completed; an embedding with non-finite values: no masks, the error
    if (!is_single_mask)
        for (int i = 0; i < 3; ++i) out_accuracy[i] = iou[i + 1];
}

// Prompts are grouped by the replica that holds their image's embedding; on each GPU they are cut into chunks of at most
// kPromptChunk, each chunk decoded as one batch on the next lane, its masks copied out while the next chunk runs.
void SegmentationImpl::compute_mask_batch(SegmentationImpl const* const* segs, int count, int const* points,
                                          int const* regions, uint8_t* const* out_masks) {
    if (count <= 0) return;
    DLIMG_ASSERT((points != nullptr) != (regions != nullptr));
    constexpr int kPromptChunk = 8;
    EnvironmentImpl& env = segs[0]->env_;
    std::vector<float> coords((size_t)count * 4), labels((size_t)count * 2);
    std::vector<int> used;
    for (int i = 0; i < count; ++i) {
        DLIMG_ASSERT(&segs[i]->env_ == &env);
        DLIMG_ASSERT(segs[i]->embedding_ != nullptr && out_masks[i] != nullptr);
        segs[i]->settle();                       // prompts of a batch go to any lane: the embeddings are complete first
        if (points) {
            Point p{points[i * 2], points[i * 2 + 1]};
            pack_prompt(segs[i]->image_size_, &p, nullptr, &coords[i * 4], &labels[i * 2]);
        } else {
            Region r{Point{regions[i * 4], regions[i * 4 + 1]}, Point{regions[i * 4 + 2], regions[i * 4 + 3]}};
            pack_prompt(segs[i]->image_size_, nullptr, &r, &coords[i * 4], &labels[i * 2]);
        }
        if (std::find(used.begin(), used.end(), segs[i]->replica_) == used.end()) used.push_back(segs[i]->replica_);
    }
    for_each_replica(env, used, [&](int replica) {
        HIP_CHECK(hipSetDevice(env.device_of(replica)));
        std::vector<int> mine;
        for (int i = 0; i < count; ++i)
            if (segs[i]->replica_ == replica) mine.push_back(i);
        struct Chunk { SamModel* model; SamModel::MaskSlot* slot; std::vector<k::PostJob> jobs; };
        std::vector<Chunk> chunks;
        auto finish = [&](Chunk& c) {
            if (!c.slot) return;
            SamModel::MaskSlot* slot = c.slot;
            c.slot = nullptr;
            try {
                c.model->finish_masks(*slot, c.jobs.data(), (int)c.jobs.size(), nullptr, 0);
            } catch (...) {
                c.model->release_mask_slot(*slot);
                throw;
            }
            c.model->release_mask_slot(*slot);
        };
        try {
            for (size_t base = 0; base < mine.size(); base += kPromptChunk) {
                const int n = (int)std::min<size_t>(kPromptChunk, mine.size() - base);
                std::vector<float const*> emb(n);
                std::vector<float> cc((size_t)n * 4), ll((size_t)n * 2);
                for (int j = 0; j < n; ++j) {
                    const int i = mine[base + j];
                    emb[j] = segs[i]->embedding_;
                    std::copy_n(&coords[(size_t)i * 4], 4, &cc[(size_t)j * 4]);
                    std::copy_n(&labels[(size_t)i * 2], 2, &ll[(size_t)j * 2]);
                }
                SamModel& model = env.next_lane(replica);
                // masks of a chunk are copied to the caller while the two chunks behind it are on the GPU
                if (chunks.size() >= 3) finish(chunks[chunks.size() - 3]);
                Chunk c{&model, &model.acquire_mask_slot(), std::vector<k::PostJob>(n)};
                chunks.push_back(std::move(c));
                Chunk& cur = chunks.back();
                roctx::Range range("dlimg.compute_masks");
                std::lock_guard<std::mutex> lock(model.mutex());
                model.decode(emb.data(), cc.data(), ll.data(), n);
                for (int j = 0; j < n; ++j) {
                    const int i = mine[base + j];
                    const Extent o = segs[i]->image_size_.original, r = segs[i]->image_size_.resized;
                    cur.jobs[j] = k::PostJob{model.logits() + (size_t)j * 4 * kLowRes * kLowRes, model.iou() + (size_t)j * 4,
                                             out_masks[i], o.width, o.height, r.width, r.height};
                }
                model.enqueue_masks(*cur.slot, cur.jobs.data(), n, 0);
            }
            for (auto& c : chunks) finish(c);
        } catch (...) {
            // a chunk whose enqueue failed has no valid completion event: drain the lanes before the slots go back
            for (auto& c : chunks)
                if (c.slot) drain_lane(c.model);
            for (auto& c : chunks) {
                try { finish(c); } catch (...) {}
            }
            throw;
        }
    });
}

// Device-output variant: the "gather" of SURVEY.md 8e.  Every GPU of the environment decodes the prompts whose
// embeddings it holds; a mask whose GPU is the root is written in place by the post-processing kernel, the others cross
// xGMI as one peer copy each (hipMemcpyPeerAsync on the producing lane's stream, so the copy of one chunk runs beside the
// decoder of the next).  No host memory is touched.  The reference has nothing comparable (one device, host tensors:
// /root/reference/src/session.cpp:63-66, /root/reference/src/environment.cpp:142).
void SegmentationImpl::compute_mask_batch_device(SegmentationImpl const* const* segs, int count, int const* points,
                                                 int const* regions, int root_device, uint8_t* dev_out,
                                                 size_t* out_offsets) {
    if (count <= 0) return;
    DLIMG_ASSERT((points != nullptr) != (regions != nullptr));
    DLIMG_ASSERT(dev_out != nullptr);
    if (root_device < 0 || root_device >= EnvironmentImpl::device_count())
        throw Exception("root device " + std::to_string(root_device) + " is out of range: " +
                        std::to_string(EnvironmentImpl::device_count()) + " device(s) visible");
    constexpr int kPromptChunk = 8;
    EnvironmentImpl& env = segs[0]->env_;
    std::vector<float> coords((size_t)count * 4), labels((size_t)count * 2);
    std::vector<size_t> offsets(count);
    std::vector<int> used;
    size_t total = 0;
    for (int i = 0; i < count; ++i) {
        DLIMG_ASSERT(&segs[i]->env_ == &env);
        DLIMG_ASSERT(segs[i]->embedding_ != nullptr);
        segs[i]->settle();
        if (points) {
            Point p{points[i * 2], points[i * 2 + 1]};
            pack_prompt(segs[i]->image_size_, &p, nullptr, &coords[i * 4], &labels[i * 2]);
        } else {
            Region r{Point{regions[i * 4], regions[i * 4 + 1]}, Point{regions[i * 4 + 2], regions[i * 4 + 3]}};
            pack_prompt(segs[i]->image_size_, nullptr, &r, &coords[i * 4], &labels[i * 2]);
        }
        offsets[i] = total;
        total += (size_t)segs[i]->image_size_.original.width * segs[i]->image_size_.original.height;
        if (std::find(used.begin(), used.end(), segs[i]->replica_) == used.end()) used.push_back(segs[i]->replica_);
    }
    if (out_offsets) std::copy(offsets.begin(), offsets.end(), out_offsets);
    for_each_replica(env, used, [&](int replica) {
        HIP_CHECK(hipSetDevice(env.device_of(replica)));
        std::vector<int> mine;
        for (int i = 0; i < count; ++i)
            if (segs[i]->replica_ == replica) mine.push_back(i);
        struct Chunk { SamModel* model; SamModel::MaskSlot* slot; };
        std::vector<Chunk> chunks;
        auto settle = [&](bool drain) {          // wait for every chunk, hand the slots back; first error wins
            std::exception_ptr first;
            for (auto& c : chunks) {
                if (!c.slot) continue;
                if (drain) drain_lane(c.model);
                try {
                    c.model->wait_masks(*c.slot);
                } catch (...) {
                    if (!first) first = std::current_exception();
                }
                c.model->release_mask_slot(*c.slot);
                c.slot = nullptr;
            }
            if (first) std::rethrow_exception(first);
        };
        try {
            for (size_t base = 0; base < mine.size(); base += kPromptChunk) {
                const int n = (int)std::min<size_t>(kPromptChunk, mine.size() - base);
                std::vector<float const*> emb(n);
                std::vector<float> cc((size_t)n * 4), ll((size_t)n * 2);
                std::vector<k::PostJob> jobs(n);
                for (int j = 0; j < n; ++j) {
                    const int i = mine[base + j];
                    emb[j] = segs[i]->embedding_;
                    std::copy_n(&coords[(size_t)i * 4], 4, &cc[(size_t)j * 4]);
                    std::copy_n(&labels[(size_t)i * 2], 2, &ll[(size_t)j * 2]);
                }
                SamModel& model = env.next_lane(replica);
                chunks.push_back(Chunk{&model, &model.acquire_mask_slot()});
                roctx::Range range("dlimg.compute_masks_device");
                std::lock_guard<std::mutex> lock(model.mutex());
                model.decode(emb.data(), cc.data(), ll.data(), n);
                for (int j = 0; j < n; ++j) {
                    const int i = mine[base + j];
                    const Extent o = segs[i]->image_size_.original, r = segs[i]->image_size_.resized;
                    jobs[j] = k::PostJob{model.logits() + (size_t)j * 4 * kLowRes * kLowRes, model.iou() + (size_t)j * 4,
                                         dev_out + offsets[i], o.width, o.height, r.width, r.height};
                }
                model.enqueue_masks_device(*chunks.back().slot, jobs.data(), n, root_device);
            }
            settle(false);
        } catch (...) {
            try { settle(true); } catch (...) {}
            throw;
        }
    });
}

}  // namespace dlimg

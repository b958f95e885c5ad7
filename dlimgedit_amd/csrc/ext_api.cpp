// Extension entry points declared in include/dlimgedit/dlimgedit_amd.h: the device-resident asynchronous path, the device
// gather, stage clocks, getters, and the library-side kernels of segment_objects.  (The single-kernel test hooks and the
// benchmark hooks are in test_hooks.cpp, which only the test and tuning libraries contain.)
#include "ext_common.hpp"
#include "image_memory.hpp"
#include "step_queue.hpp"
#include "resize_tables.hpp"

#include <dlimgedit/dlimgedit_amd.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dlimg {
namespace {

using namespace extapi;

// Shared by encode_and_mask / encode_only: images already on the device.
void encode_device_images(SamModel& model, dlimg_ImageView const* imgs, int count) {
    for (int i = 0; i < count; ++i) {
        check_image(imgs[i]);
        if (std::max(imgs[i].width, imgs[i].height) != kImageSize)
            throw Exception("device-resident images must have their longest side at 1024 pixels");
    }
    model.preprocess_device_images(imgs, count);
    model.encode(count);
}

// every lane of every replica
template <typename F> void for_each_lane(EnvironmentImpl& e, F&& body) {
    for (int r = 0; r < e.replica_count(); ++r)
        for (int i = 0; i < e.lane_count(r); ++i) {
            SamModel& m = e.lane(r, i);
            std::lock_guard<std::mutex> lock(m.mutex());
            HIP_CHECK(hipSetDevice(m.device()));
            body(m);
        }
}

}  // namespace
}  // namespace dlimg

using namespace dlimg;

extern "C" {

DLIMG_API int dlimg_amd_device_count(void) { return EnvironmentImpl::device_count(); }

DLIMG_API int dlimg_amd_model_geometry(dlimg_Environment env, int* out) {
    return guarded([&] {
        SamGeometry const& g = impl(env).lane(0, 0).geometry();
        out[0] = g.embed_dim;
        out[1] = g.depth;
        out[2] = g.num_heads;
        out[3] = g.mlp_dim;
    });
}

DLIMG_API int dlimg_amd_get_embedding(dlimg_Segmentation seg, float* out) {
    return guarded([&] {
        SegmentationImpl& s = impl(seg);
        DLIMG_ASSERT(s.embedding() != nullptr && out != nullptr);
        HIP_CHECK(hipSetDevice(s.environment().device_of(s.replica())));
        download(out, s.embedding(), (size_t)kTokens * kEmbedDim);      // process() has synchronised already
    });
}

DLIMG_API int dlimg_amd_get_logits(dlimg_Segmentation seg, int const* point, int const* region, float* out_logits,
                                   float* out_iou) {
    return guarded([&] {
        SegmentationImpl& s = impl(seg);
        DLIMG_ASSERT(s.embedding() != nullptr);
        Point p;
        Region r;
        if (point) p = Point{point[0], point[1]};
        if (region) r = Region{Point{region[0], region[1]}, Point{region[2], region[3]}};
        float coords[4], labels[2];
        pack_prompt(s.geometry(), point ? &p : nullptr, !point && region ? &r : nullptr, coords, labels);
        SamModel& m = s.environment().next_lane(s.replica());
        std::lock_guard<std::mutex> lock(m.mutex());
        HIP_CHECK(hipSetDevice(m.device()));
        float const* emb = s.embedding();
        m.decode(&emb, coords, labels, 1);
        m.synchronize();
        download(out_logits, m.logits(), (size_t)4 * kLowRes * kLowRes);
        download(out_iou, m.iou(), 4);
    });
}

DLIMG_API int dlimg_amd_decoder_state(dlimg_Segmentation seg, int const* point, float* out, int capacity, char* out_layout,
                                      int layout_capacity) {
    return guarded([&] {
        SegmentationImpl& s = impl(seg);
        DLIMG_ASSERT(s.embedding() != nullptr && point != nullptr);
        std::string layout;
        size_t total = 0;
        for (auto const& part : SamModel::decoder_state_layout()) {
            layout += std::string(part.first) + ":" + std::to_string(part.second) + ",";
            total += part.second;
        }
        if (out_layout && layout_capacity > 0) {
            std::snprintf(out_layout, (size_t)layout_capacity, "%s", layout.c_str());
        }
        if (!out) return;
        if ((size_t)capacity < total) throw Exception("decoder_state: the output buffer is too small");
        Point p{point[0], point[1]};
        float coords[4], labels[2];
        pack_prompt(s.geometry(), &p, nullptr, coords, labels);
        SamModel& m = s.environment().next_lane(s.replica());
        std::lock_guard<std::mutex> lock(m.mutex());
        HIP_CHECK(hipSetDevice(m.device()));
        float const* emb = s.embedding();
        m.decode(&emb, coords, labels, 1);
        m.synchronize();
        m.decoder_state(out);
    });
}

DLIMG_API int dlimg_amd_device_alloc(dlimg_Environment env, size_t bytes, void** out_ptr) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(impl(env).first_device()));
        HIP_CHECK(hipMalloc(out_ptr, bytes));
    });
}

DLIMG_API int dlimg_amd_device_free(dlimg_Environment env, void* ptr) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(impl(env).first_device()));
        HIP_CHECK(hipFree(ptr));
    });
}

DLIMG_API int dlimg_amd_copy_to_device(dlimg_Environment env, void* dst, void const* src, size_t bytes) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(impl(env).first_device()));
        HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    });
}

DLIMG_API int dlimg_amd_copy_to_host(dlimg_Environment env, void* dst, void const* src, size_t bytes) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(impl(env).first_device()));
        HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    });
}

DLIMG_API int dlimg_amd_encode_only(dlimg_Environment env, dlimg_ImageView const* dev_images, int count) {
    return guarded([&] {
        DLIMG_ASSERT(dev_images != nullptr && count > 0);
        SamModel& m = impl(env).next_lane(0);
        std::lock_guard<std::mutex> lock(m.mutex());
        HIP_CHECK(hipSetDevice(m.device()));
        encode_device_images(m, dev_images, count);
    });
}

namespace dlimg {
namespace {

// One batched pass of the device-resident hot path over `steps` on lane `lane` of replica 0: pre-process, encode, decode
// one point prompt per image (single-mask mode), masks to the callers' device buffers.  Enqueues only; returns the
// event recorded behind the pass.
hipEvent_t enqueue_device_steps(EnvironmentImpl& env, int lane, EnvironmentImpl::PendingStep const* steps, int count,
                                const volatile int** out_overflow) {
    SamModel& m = env.lane(0, lane);
    std::lock_guard<std::mutex> lock(m.mutex());
    HIP_CHECK(hipSetDevice(m.device()));
    std::vector<dlimg_ImageView> views(count);
    for (int i = 0; i < count; ++i) views[i] = steps[i].view;
    encode_device_images(m, views.data(), count);
    *out_overflow = m.last_pass_flag();          // THIS pass's report: owned by its ticket from here on
    std::vector<float> coords((size_t)count * 4), labels((size_t)count * 2);
    std::vector<float const*> emb(count);
    std::vector<ResizeLongestSide> rs(count);
    for (int i = 0; i < count; ++i) {
        rs[i].set(Extent{views[i].width, views[i].height});
        Point p{steps[i].x, steps[i].y};
        pack_prompt(rs[i], &p, nullptr, &coords[i * 4], &labels[i * 2]);
        emb[i] = m.embeddings() + (size_t)i * kTokens * kEmbedDim;
    }
    m.decode(emb.data(), coords.data(), labels.data(), count);
    std::vector<k::PostJob> jobs(count);
    for (int i = 0; i < count; ++i)
        jobs[i] = k::PostJob{m.logits() + (size_t)i * 4 * kLowRes * kLowRes, m.iou() + (size_t)i * 4, steps[i].mask,
                             rs[i].original.width, rs[i].original.height, rs[i].resized.width, rs[i].resized.height};
    m.masks_on_device(jobs.data(), count);
    return m.completion();
}

// A pass whose enqueue failed takes only ITS OWN requests with it: the failure is kept as a sticky error that the next
// dlimg_amd_synchronize returns -- the callers whose requests were dropped got a success code when they queued them
// (possibly on other threads), synchronize is where they learn that a mask will not arrive.
void note_failed_pass(EnvironmentImpl& env, int images, const char* what) {
    std::lock_guard<std::mutex> lock(env.step_error_mutex);
    env.dropped_steps += images;
    if (env.step_error.empty()) env.step_error = what;
}

// Hands one planned pass to its lane: to the lane's enqueue thread (LaneWorker, environment.hpp), or enqueued right here
// with DLIMGEDIT_STEP_WORKERS=0.  Notes the pass in env.step_passes either way.  pending_mutex held by the caller.
void run_device_steps(EnvironmentImpl& env, int lane, EnvironmentImpl::PendingStep const* steps, int count) {
    auto ticket = std::make_shared<EnvironmentImpl::StepTicket>();
    if ((int)env.step_passes.size() < env.lane_count(0)) env.step_passes.resize(env.lane_count(0));
    if (!env.use_step_workers) {
        try {
            ticket->done = enqueue_device_steps(env, lane, steps, count, &ticket->overflow);
            ticket->state.store(1, std::memory_order_release);
        } catch (std::exception const& ex) {
            note_failed_pass(env, count, ex.what());
            throw;
        }
        env.step_passes[lane].push_back(EnvironmentImpl::StepPass{ticket, count});
        return;
    }
    env.step_passes[lane].push_back(EnvironmentImpl::StepPass{ticket, count});
    std::vector<EnvironmentImpl::PendingStep> owned(steps, steps + count);
    EnvironmentImpl* e = &env;
    env.lane_worker(0, lane).post([e, lane, ticket, owned = std::move(owned)] {
        try {
            ticket->done = enqueue_device_steps(*e, lane, owned.data(), (int)owned.size(), &ticket->overflow);
            ticket->state.store(1, std::memory_order_release);
        } catch (std::exception const& ex) {
            note_failed_pass(*e, (int)owned.size(), ex.what());
            ticket->state.store(2, std::memory_order_release);
        } catch (...) {
            note_failed_pass(*e, (int)owned.size(), "unknown error");
            ticket->state.store(2, std::memory_order_release);
        }
    });
}

// Forgets the passes that have finished and returns the lanes requests are spread over.
// lanes the step queue deals its passes to, and its pass width: the environment's settings, or the model's defaults
// (environment.hpp: ViT-B four images per pass on three lanes, larger models two images on every lane)
// (defaults by model width, measured on MI355X with the official 20-step block and with 50-step blocks: ViT-B four images
// per pass on three of its four lanes (r05); ViT-H THREE per pass on TWO of its three lanes (r06: 168.3-168.8 -> 171.1-171.2
// images/s on one box, 166.6-167.0 -> 169.9-170.6 on another; 2 x 2: +0.7 %, 4 / 5 / 6 / 10 per pass and one or three lanes:
// no better than the old two per pass on every lane) -- three images give its N = 1280 stream writers 240 tiles of the 256
// CUs where two give 160; ViT-L FOUR per pass on two lanes (256 tiles for its N = 1024 stream writers: 333.6-335.8 -> 342.5-346.0))
int step_queue_lanes(EnvironmentImpl& env) {
    const int lanes = std::max(1, env.effective_lane_count(0));
    int want = env.step_lanes;
    const int width = env.lane(0, 0).geometry().embed_dim;
    if (want < 0) want = width <= 768 ? 3 : 2;
    return want > 0 ? std::min(lanes, want) : lanes;
}
int step_queue_width(EnvironmentImpl& env) {
    if (env.coalesce > 0) return env.coalesce;
    const int width = env.lane(0, 0).geometry().embed_dim;
    return width >= 1280 ? 3 : 4;
}

int retire_device_steps(EnvironmentImpl& env) {
    const int lanes = step_queue_lanes(env);
    if ((int)env.step_passes.size() < env.lane_count(0)) env.step_passes.resize(env.lane_count(0));
    for (size_t l = 0; l < env.step_passes.size(); ++l) {
        auto& q = env.step_passes[l];
        while (!q.empty()) {
            EnvironmentImpl::StepTicket& t = *q.front().ticket;
            const int state = t.state.load(std::memory_order_acquire);
            if (state == 0) break;                                     // still with the lane's enqueue thread
            if (state == 1 && !env.lane(0, (int)l).poll_and_recycle(t.done)) break;
            t.done = nullptr;                                          // recycled (or never recorded: state 2)
            // the pass has run: its own f16-range report, and only its own (process() / process_batch() callers on the same
            // lane read theirs themselves; SamModel hands every pass another slot of the lane's ring)
            if (state == 1 && t.overflow && *t.overflow != 0) {
                std::lock_guard<std::mutex> errors(env.step_error_mutex);
                env.overflowed_steps += q.front().images;
            }
            q.pop_front();
        }
    }
    return lanes;
}

// The queue's state as the planner sees it (step_queue.hpp); `lanes` = the lanes requests are spread over
void queue_state(EnvironmentImpl& env, int lanes, StepQueueState& st) {
    st.cursor = env.step_cursor % lanes;
    for (int l = 0; l < lanes; ++l) {
        int images = 0;
        for (auto const& pass : env.step_passes[l]) images += pass.images;
        st.passes_in_flight.push_back((int)env.step_passes[l].size());
        st.images_in_flight.push_back(images);
    }
}

// Launches what is waiting (environment.hpp, PendingStep) as step_queue.hpp plans it.  pending_mutex held by the caller.
// [Holding requests back until a whole wave of lanes x coalesce had arrived was measured and gained nothing on a burst of
// 20 requests, while it delays the first launch.  Under rocprofv3's kernel trace hipEventQuery reports every pass as
// finished: the planner's tie-break keeps the lanes taking turns there.]
void flush_device_steps(EnvironmentImpl& env, bool all) {
    if (env.pending.empty()) return;
    const int lanes = retire_device_steps(env);
    StepQueueState st;
    queue_state(env, lanes, st);
    const std::vector<StepPlanPass> plan = plan_device_steps(st, (int)env.pending.size(), step_queue_width(env), env.step_depth, all);
    env.step_cursor = st.cursor;
    // A pass that fails to enqueue takes only ITS OWN requests with it (note_failed_pass): the passes launched before it
    // stay launched, the requests behind it stay queued for the next call.  With enqueue threads nothing fails here.
    size_t done = 0;
    try {
        for (StepPlanPass const& pass : plan) {
            run_device_steps(env, pass.lane, env.pending.data() + done, pass.images);
            done += (size_t)pass.images;
        }
    } catch (std::exception const&) {
        size_t lost = 0;
        size_t at = 0;
        for (StepPlanPass const& pass : plan) {          // the pass that threw is the first one not counted in `done`
            if (at == done) { lost = (size_t)pass.images; break; }
            at += (size_t)pass.images;
        }
        env.pending.erase(env.pending.begin(), env.pending.begin() + std::min(env.pending.size(), done + lost));
        throw;
    }
    env.pending.erase(env.pending.begin(), env.pending.begin() + done);
}

// a call that is a batch already: the lane with the fewest passes in flight, whatever its depth
int lane_for_batch(EnvironmentImpl& env) {
    const int lanes = retire_device_steps(env);
    StepQueueState st;
    queue_state(env, lanes, st);
    const std::vector<StepPlanPass> plan = plan_device_steps(st, 1, 1, 1 << 30, false);
    env.step_cursor = st.cursor;
    return plan.empty() ? 0 : plan[0].lane;
}

}  // namespace
}  // namespace dlimg

DLIMG_API int dlimg_amd_encode_and_mask(dlimg_Environment env, dlimg_ImageView const* dev_images, int count,
                                        int const* points, uint8_t* const* dev_masks) {
    return guarded([&] {
        DLIMG_ASSERT(dev_images != nullptr && points != nullptr && dev_masks != nullptr && count > 0);
        EnvironmentImpl& e = impl(env);
        for (int i = 0; i < count; ++i) {
            check_image(dev_images[i]);
            DLIMG_ASSERT(dev_masks[i] != nullptr);
            if (std::max(dev_images[i].width, dev_images[i].height) != kImageSize)
                throw Exception("device-resident images must have their longest side at 1024 pixels");
        }
        std::lock_guard<std::mutex> lock(e.pending_mutex);
        if (count >= step_queue_width(e)) {
            // a call that is a batch already runs as it is (behind whatever single requests were waiting)
            flush_device_steps(e, true);
            std::vector<EnvironmentImpl::PendingStep> steps(count);
            for (int i = 0; i < count; ++i)
                steps[i] = EnvironmentImpl::PendingStep{dev_images[i], points[i * 2], points[i * 2 + 1], dev_masks[i]};
            run_device_steps(e, lane_for_batch(e), steps.data(), count);
            return;
        }
        for (int i = 0; i < count; ++i)
            e.pending.push_back(EnvironmentImpl::PendingStep{dev_images[i], points[i * 2], points[i * 2 + 1], dev_masks[i]});
        flush_device_steps(e, false);
    });
}

DLIMG_API int dlimg_amd_get_segmentation_masks_device(dlimg_Segmentation const* segs, int count, int const* points,
                                                      int const* regions, int root_device, uint8_t* dev_out,
                                                      size_t* out_offsets) {
    return guarded([&] {
        DLIMG_ASSERT(segs != nullptr && count >= 0);
        std::vector<SegmentationImpl const*> s(count);
        for (int i = 0; i < count; ++i) {
            DLIMG_ASSERT(segs[i] != nullptr);
            s[i] = &impl(segs[i]);
        }
        SegmentationImpl::compute_mask_batch_device(s.data(), count, points, regions, root_device, dev_out, out_offsets);
    });
}

DLIMG_API int dlimg_amd_synchronize(dlimg_Environment env) {
    return guarded([&] {
        EnvironmentImpl& e = impl(env);
        std::string flush_error;
        {
            std::lock_guard<std::mutex> lock(e.pending_mutex);
            // keep dealing out what is queued even if one pass fails: every request either runs or is counted as dropped
            for (int guard = 0; !e.pending.empty() && guard < 1024; ++guard) {
                try {
                    flush_device_steps(e, true);
                } catch (std::exception const&) {
                }                                   // recorded in e.step_error (note_failed_pass)
            }
            e.drain_step_workers();                 // (the workers never take pending_mutex)
        }
        for_each_lane(e, [](SamModel& m) { m.synchronize(); });
        // an encoder pass whose activations left the f16 range says so in ITS flag, which the pass's ticket owns
        // (retire_device_steps); the asynchronous entry point learns it here, like every other failure of a queued pass
        std::lock_guard<std::mutex> lock(e.pending_mutex);
        retire_device_steps(e);
        std::lock_guard<std::mutex> errors(e.step_error_mutex);
        std::string msg;
        if (!e.step_error.empty())
            msg = std::to_string(e.dropped_steps) + " queued request(s) were dropped because their pass failed: " + e.step_error;
        if (e.overflowed_steps > 0)
            msg += std::string(msg.empty() ? "" : "; ") + std::to_string(e.overflowed_steps) + " request(s) ran in an image encoder "
                   "pass that produced non-finite values (an activation left the f16 range): their masks are not valid";
        e.step_error.clear();
        e.dropped_steps = 0;
        e.overflowed_steps = 0;
        if (!msg.empty()) throw Exception("dlimg_amd_encode_and_mask: " + msg);
    });
}

DLIMG_API int dlimg_amd_queue_config(dlimg_Environment env, int* out) {
    return guarded([&] {
        DLIMG_ASSERT(out != nullptr);
        EnvironmentImpl& e = impl(env);
        out[0] = step_queue_width(e);
        out[1] = e.step_depth;
        out[2] = e.lane_count(0);
        out[3] = step_queue_lanes(e);
        LaneBoard const* board = e.lane(0, 0).board();
        out[4] = board ? (int)std::min<long>(board->passes(), 0x7fffffff) : 0;
        out[5] = board ? (int)std::min<long>(board->alone_passes(), 0x7fffffff) : 0;
    });
}

DLIMG_API int dlimg_amd_image_memory(void const* pixels, size_t bytes, int* out_pinned) {
    return guarded([&] {
        DLIMG_ASSERT(out_pinned != nullptr);
        *out_pinned = image_memory_is_pinned(pixels, bytes) ? 1 : 0;
    });
}

DLIMG_API int dlimg_amd_set_profiling(dlimg_Environment env, int enabled) {
    return guarded([&] {
        EnvironmentImpl& e = impl(env);
        {
            std::lock_guard<std::mutex> lock(e.pending_mutex);
            flush_device_steps(e, true);
            e.drain_step_workers();
        }
        // drain everything; mode 1 pins requests to lane 0 while the clocks run (every kernel alone on the chip),
        // mode 2 leaves the lanes as they are (the regime the throughput figure is measured in)
        for_each_lane(e, [](SamModel& m) { m.synchronize(); });
        e.set_single_lane(enabled == 1);
        for_each_lane(e, [&](SamModel& m) { m.set_profiling(enabled == 2 || (enabled == 1 && &m == &e.lane(0, 0))); });
    });
}

DLIMG_API int dlimg_amd_take_stage_stats(dlimg_Environment env, double* out_ms, double* out_work, long* out_launches) {
    static_assert(ST_COUNT == DLIMG_AMD_STAGE_COUNT, "stage table out of sync with the public header");
    return guarded([&] {
        StageStats total;
        for_each_lane(impl(env), [&](SamModel& m) {
            StageStats s = m.take_stats();
            for (int i = 0; i < ST_COUNT; ++i) {
                total.ms[i] += s.ms[i];
                total.work[i] += s.work[i];
                total.launches[i] += s.launches[i];
            }
        });
        for (int i = 0; i < ST_COUNT; ++i) {
            if (out_ms) out_ms[i] = total.ms[i];
            if (out_work) out_work[i] = total.work[i];
            if (out_launches) out_launches[i] = total.launches[i];
        }
    });
}

DLIMG_API int dlimg_amd_lane_count(dlimg_Environment env) {
    int n = 0;
    guarded([&] { n = impl(env).lane_count(0); });
    return n;
}

DLIMG_API int dlimg_amd_replica_count(dlimg_Environment env) {
    int n = 0;
    guarded([&] { n = impl(env).replica_count(); });
    return n;
}

DLIMG_API int dlimg_amd_segmentation_device(dlimg_Segmentation seg, int* out_replica, int* out_device) {
    return guarded([&] {
        SegmentationImpl& s = impl(seg);
        if (out_replica) *out_replica = s.replica();
        if (out_device) *out_device = s.environment().device_of(s.replica());
    });
}

// ---- pre / post-processing of segment_objects (BiRefNet) -------------------------

DLIMG_API int dlimg_amd_birefnet_prepare_image(uint8_t const* pixels, int width, int height, int stride, int channels,
                                               float const* mean, float const* std, float* out_nchw) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(pixels && mean && std && out_nchw && width > 0 && height > 0);
        const int C = channel_bytes(channels);
        if (C < 3) throw Exception("prepare_image: needs an image with at least three channels");
        DLIMG_ASSERT(stride >= width * C);
        Upload<uint8_t> src(pixels, (size_t)stride * height);
        DeviceBuffer<float> dst((size_t)3 * width * height);
        k::birefnet_prepare_image(src.get(), width, height, stride, C, mean, std, dst.get(), nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_nchw, dst.get(), (size_t)3 * width * height);
    });
}

DLIMG_API int dlimg_amd_birefnet_process_mask(float const* logits, int width, int height, uint8_t* out_mask) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(logits && out_mask && width > 0 && height > 0);
        Upload<float> src(logits, (size_t)width * height);
        DeviceBuffer<uint8_t> dst((size_t)width * height);
        k::birefnet_process_mask(src.get(), width, height, dst.get(), nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_mask, dst.get(), (size_t)width * height);
    });
}

DLIMG_API int dlimg_amd_resize_mask(uint8_t const* mask, int width, int height, int stride, int out_w, int out_h,
                                    uint8_t* out_mask) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(mask && out_mask && width > 0 && height > 0 && out_w > 0 && out_h > 0 && stride >= width);
        AxisTable tx = make_axis_table(width, out_w, ResizeFilter::box), ty = make_axis_table(height, out_h, ResizeFilter::box);
        float lut[256];
        for (int i = 0; i < 256; ++i) lut[i] = float(i) / 255.0f;        // STBIR_COLORSPACE_LINEAR decode
        Upload<uint8_t> src(mask, (size_t)stride * height);
        Upload<int> xf(tx.first.data(), tx.first.size()), xc(tx.count.data(), tx.count.size());
        Upload<int> yf(ty.first.data(), ty.first.size()), yc(ty.count.data(), ty.count.size());
        Upload<float> xk(tx.coef.data(), tx.coef.size()), yk(ty.coef.data(), ty.coef.size()), dlut(lut, 256);
        DeviceBuffer<float> tmp((size_t)height * out_w);
        DeviceBuffer<uint8_t> dst((size_t)out_w * out_h);
        k::ResizeAxis ax{xf.get(), xc.get(), xk.get(), tx.taps, out_w}, ay{yf.get(), yc.get(), yk.get(), ty.taps, out_h};
        k::resize_srgb(src.get(), width, height, stride, 1, ax, ay, dlut.get(), nullptr, tmp.get(), dst.get(), nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_mask, dst.get(), (size_t)out_w * out_h);
    });
}

}  // extern "C"

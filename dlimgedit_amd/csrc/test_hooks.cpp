// Test and benchmark hooks (include/dlimgedit/dlimgedit_amd_test.h): single kernels behind plain host buffers for the
// parity tests, host-logic checks that need no GPU, and the kernels-alone timing loops of tools/ and bench.py's
// `hbm_kernels` leg.  NOT part of the product library: python -m dlimgedit_amd.build links this file into
// lib/libdlimgedit_test.so (same objects as the product plus this one) and into the tuning library only.
#include "ext_common.hpp"
#include "step_queue.hpp"
#include "mask_pieces.hpp"
#include "resize_tables.hpp"

#include <dlimgedit/dlimgedit_amd_test.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dlimg {
namespace {

using namespace extapi;

// dlimg_amd_test_force_gemm_tile() forces a tile configuration of kernels/gemm.hip in the single-kernel hooks below
// wherever it fits the problem.  An explicit call on this library's own state: nothing can steer the product's tile choice.
std::atomic<int> g_forced_test_tile{-1};
std::atomic<int> g_forced_consumer_tile{-1};     // LayerNorm-folded consumers only; -1: as g_forced_test_tile
void apply_forced_tile(k::GemmArgs& g) {
    const int consumer = g_forced_consumer_tile.load(std::memory_order_relaxed);
    const int t = (g.ln_stats && consumer >= 0) ? consumer : g_forced_test_tile.load(std::memory_order_relaxed);
    if (t < 0 || g.tile >= 0) return;
    if (k::gemm_tile_fits(g, t)) g.tile = t;
}

}  // namespace
}  // namespace dlimg

using namespace dlimg;

extern "C" {

DLIMG_API int dlimg_amd_test_mask_pieces(int count, long long const* mask_bytes, long long extra_bytes, long long* out_piece_end,
                                         int piece_capacity, long long* out_copies, int copy_capacity, int* out_pieces) {
    int copies = -1;
    const int rc = guarded([&] {
        DLIMG_ASSERT(count >= 0 && mask_bytes && out_piece_end && out_copies && out_pieces);
        std::vector<size_t> sizes(count);
        size_t total = 0;
        for (int i = 0; i < count; ++i) {
            sizes[i] = (size_t)mask_bytes[i];
            total += padded_mask_bytes(sizes[i]);
        }
        total += (size_t)extra_bytes;
        const std::vector<size_t> ends = mask_piece_ends(total);
        if ((int)ends.size() > piece_capacity) throw Exception("test_mask_pieces: the piece array is too small");
        MaskCursor cursor;
        size_t begin = 0;
        int n = 0;
        for (size_t i = 0; i < ends.size(); ++i) {
            out_piece_end[i] = (long long)ends[i];
            for (MaskCopy const& c : mask_copies_in_piece(sizes, begin, ends[i], cursor)) {
                if (n >= copy_capacity) throw Exception("test_mask_pieces: the copy array is too small");
                out_copies[n * 5 + 0] = (long long)i;
                out_copies[n * 5 + 1] = c.mask;
                out_copies[n * 5 + 2] = (long long)c.staging_offset;
                out_copies[n * 5 + 3] = (long long)c.mask_offset;
                out_copies[n * 5 + 4] = (long long)c.bytes;
                ++n;
            }
            begin = ends[i];
        }
        *out_pieces = (int)ends.size();
        copies = n;
    });
    return rc == 0 ? copies : -1;
}

// LaneWorker (environment.hpp) without a GPU: `tasks` tasks that each sleep `sleep_us` and then note their index; drain()
// must return only after all of them, in posting order; one more task posted afterwards must have run by the time the
// destructor returns.  out_order: tasks + 1 entries.  Returns the number of tasks that ran.
DLIMG_API int dlimg_amd_test_lane_worker(int tasks, int sleep_us, int* out_order) {
    int ran = -1;
    const int rc = guarded([&] {
        DLIMG_ASSERT(tasks >= 0 && out_order != nullptr);
        std::vector<int> order;
        std::mutex m;
        int after_drain = -1;
        {
            LaneWorker w;
            for (int i = 0; i < tasks; ++i)
                w.post([&, i] {
                    if (sleep_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
                    std::lock_guard<std::mutex> lock(m);
                    order.push_back(i);
                });
            w.drain();
            {
                std::lock_guard<std::mutex> lock(m);
                after_drain = (int)order.size();
            }
            w.post([&] {
                if (sleep_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
                std::lock_guard<std::mutex> lock(m);
                order.push_back(tasks);
            });
        }                                        // the destructor finishes what is queued
        if (after_drain != tasks) throw Exception("LaneWorker::drain returned with tasks outstanding");
        for (size_t i = 0; i < order.size(); ++i) out_order[i] = order[i];
        ran = (int)order.size();
    });
    return rc == 0 ? ran : -1;
}

// "0-3,8,10-11" -> CPU indices (the parser behind bind_thread_near_device, environment.cpp); returns their number (-1: error)
DLIMG_API int dlimg_amd_test_parse_cpu_list(char const* text, int* out_cpus, int capacity) {
    int count = -1;
    const int rc = guarded([&] {
        DLIMG_ASSERT(text != nullptr && out_cpus != nullptr && capacity >= 0);
        const std::vector<int> cpus = parse_cpu_list(text);
        if ((int)cpus.size() > capacity) throw Exception("test_parse_cpu_list: the output array is too small");
        for (size_t i = 0; i < cpus.size(); ++i) out_cpus[i] = cpus[i];
        count = (int)cpus.size();
    });
    return rc == 0 ? count : -1;
}

DLIMG_API int dlimg_amd_test_plan_steps(int lanes, int* passes_in_flight, int* images_in_flight, int* cursor, int pending, int width,
                                        int depth, int all, int* out_lane, int* out_images, int capacity) {
    int planned = -1;
    const int rc = guarded([&] {
        DLIMG_ASSERT(lanes > 0 && passes_in_flight && images_in_flight && cursor && capacity >= 0);
        StepQueueState st;
        st.passes_in_flight.assign(passes_in_flight, passes_in_flight + lanes);
        st.images_in_flight.assign(images_in_flight, images_in_flight + lanes);
        st.cursor = *cursor % lanes;
        const std::vector<StepPlanPass> plan = plan_device_steps(st, pending, width, depth, all != 0);
        if ((int)plan.size() > capacity) throw Exception("test_plan_steps: the output arrays are too small");
        for (size_t i = 0; i < plan.size(); ++i) {
            out_lane[i] = plan[i].lane;
            out_images[i] = plan[i].images;
        }
        std::copy(st.passes_in_flight.begin(), st.passes_in_flight.end(), passes_in_flight);
        std::copy(st.images_in_flight.begin(), st.images_in_flight.end(), images_in_flight);
        *cursor = st.cursor;
        planned = (int)plan.size();
    });
    return rc == 0 ? planned : -1;
}

DLIMG_API int dlimg_amd_test_preprocess(uint8_t const* pixels, int width, int height, int stride, int channels,
                                        uint16_t* out_patches) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(pixels != nullptr && out_patches != nullptr && height > 0 && stride > 0);
        Upload<uint8_t> img(pixels, (size_t)stride * height);
        DeviceBuffer<half_t> out((size_t)kTokens * kPatchK);
        k::preprocess(img.get(), width, height, stride, channels, out.get(), nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(reinterpret_cast<half_t*>(out_patches), out.get(), (size_t)kTokens * kPatchK);
    });
}

DLIMG_API int dlimg_amd_test_postprocess(float const* planes, int n_planes, float const* iou, int out_w, int out_h,
                                         uint8_t* out_mask) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(planes != nullptr && out_mask != nullptr && n_planes >= 1 && (!iou || n_planes == 4));
        DLIMG_ASSERT(out_w > 0 && out_h > 0);
        Upload<float> src(planes, (size_t)n_planes * kLowRes * kLowRes);
        Upload<float> sel(iou, 4);
        DeviceBuffer<uint8_t> dst((size_t)out_w * out_h);
        ResizeLongestSide rs;
        rs.set(Extent{out_w, out_h});
        k::PostJob job{src.get(), iou ? sel.get() : nullptr, dst.get(), out_w, out_h, rs.resized.width, rs.resized.height};
        k::postprocess_masks(&job, 1, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_mask, dst.get(), (size_t)out_w * out_h);
    });
}

DLIMG_API int dlimg_amd_test_postprocess_batch(float const* planes, int n_masks, int out_w, int out_h, uint8_t* out_masks) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(planes != nullptr && out_masks != nullptr && n_masks >= 1 && n_masks <= 16 && out_w > 0 && out_h > 0);
        const size_t px = (size_t)out_w * out_h;
        Upload<float> src(planes, (size_t)n_masks * kLowRes * kLowRes);
        DeviceBuffer<uint8_t> dst(px * n_masks);
        HIP_CHECK(hipMemset(dst.get(), 0x5a, px * n_masks));                  // nothing but what the kernel writes counts
        ResizeLongestSide rs;
        rs.set(Extent{out_w, out_h});
        std::vector<k::PostJob> jobs;
        for (int i = 0; i < n_masks; ++i)
            jobs.push_back(k::PostJob{src.get() + (size_t)i * kLowRes * kLowRes, nullptr, dst.get() + i * px, out_w, out_h,
                                      rs.resized.width, rs.resized.height});
        k::postprocess_masks(jobs.data(), n_masks, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_masks, dst.get(), px * n_masks);
    });
}

DLIMG_API int dlimg_amd_test_force_gemm_tile(int tile) {
    g_forced_test_tile.store(tile < 0 ? -1 : tile, std::memory_order_relaxed);
    g_forced_consumer_tile.store(-1, std::memory_order_relaxed);
    return 0;
}

DLIMG_API int dlimg_amd_test_force_gemm_consumer_tile(int tile) {
    g_forced_consumer_tile.store(tile < 0 ? -1 : tile, std::memory_order_relaxed);
    return 0;
}

DLIMG_API int dlimg_amd_test_gemm(int M, int N, int K, uint16_t const* A, uint16_t const* W, float const* bias,
                                  float const* resid, int resid_rows, int act, float* out_f32, uint16_t* out_f16) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(M > 0 && N > 0 && K > 0 && A != nullptr && W != nullptr);
        Upload<half_t> a(reinterpret_cast<half_t const*>(A), (size_t)M * K);
        Upload<half_t> w(reinterpret_cast<half_t const*>(W), (size_t)N * K);
        Upload<float> b(bias, N);
        Upload<float> r(resid, resid ? (size_t)resid_rows * N : 0);
        DeviceBuffer<float> o32(out_f32 ? (size_t)M * N : 0);
        DeviceBuffer<half_t> o16(out_f16 ? (size_t)M * N : 0);
        k::GemmArgs g;
        g.A = a.get(); g.lda = K; g.W = w.get(); g.ldw = K; g.bias = bias ? b.get() : nullptr;
        g.resid = resid ? r.get() : nullptr; g.ldr = N; g.resid_mod = resid ? resid_rows : 1;
        g.out_f32 = out_f32 ? o32.get() : nullptr; g.ldc32 = N;
        g.out_h = out_f16 ? o16.get() : nullptr; g.ldc16 = N;
        g.M = M; g.N = N; g.K = K; g.act = act;
        apply_forced_tile(g);
        k::gemm(g, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_f32, o32.get(), out_f32 ? (size_t)M * N : 0);
        download(reinterpret_cast<half_t*>(out_f16), o16.get(), out_f16 ? (size_t)M * N : 0);
    });
}

DLIMG_API int dlimg_amd_test_gemm_ln(int M, int D, int K1, int N, uint16_t const* A1, uint16_t const* W1,
                                     float const* bias1, float const* resid, uint16_t const* Wg, float const* colsum,
                                     float const* bias2, float eps, int act, float* out_x, uint16_t* out_xh,
                                     float* out_y) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(M > 0 && D > 0 && K1 > 0 && N > 0);
        DLIMG_ASSERT(A1 && W1 && Wg && colsum && out_x && out_xh && out_y);
        Upload<half_t> a1(reinterpret_cast<half_t const*>(A1), (size_t)M * K1);
        Upload<half_t> w1(reinterpret_cast<half_t const*>(W1), (size_t)D * K1);
        Upload<half_t> wg(reinterpret_cast<half_t const*>(Wg), (size_t)N * D);
        Upload<float> b1(bias1, bias1 ? D : 0), r(resid, resid ? (size_t)M * D : 0), cs(colsum, N), b2(bias2, bias2 ? N : 0);
        DeviceBuffer<float> x((size_t)M * D), y((size_t)M * N), stats((size_t)M * 24 * 2);
        DeviceBuffer<half_t> xh((size_t)M * D);
        k::GemmArgs g;      // producer: writes the stream, its f16 copy and the per-tile row statistics
        g.A = a1.get(); g.lda = K1; g.W = w1.get(); g.ldw = K1; g.bias = bias1 ? b1.get() : nullptr;
        g.resid = resid ? r.get() : nullptr; g.ldr = D; g.resid_mod = M;
        g.out_f32 = x.get(); g.ldc32 = D; g.out_h = xh.get(); g.ldc16 = D; g.stats_out = stats.get();
        g.M = M; g.N = D; g.K = K1;
        apply_forced_tile(g);
        const int groups = D / k::gemm_choose_tile(g);
        k::gemm(g, nullptr);
        g = k::GemmArgs{};  // consumer: LayerNorm folded in
        g.A = xh.get(); g.lda = D; g.W = wg.get(); g.ldw = D; g.bias = bias2 ? b2.get() : nullptr;
        g.ln_stats = stats.get(); g.ln_groups = groups; g.ln_colsum = cs.get(); g.ln_eps = eps;
        g.out_f32 = y.get(); g.ldc32 = N; g.M = M; g.N = N; g.K = D; g.act = act;
        apply_forced_tile(g);
        k::gemm(g, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_x, x.get(), (size_t)M * D);
        download(reinterpret_cast<half_t*>(out_xh), xh.get(), (size_t)M * D);
        download(out_y, y.get(), (size_t)M * N);
    });
}

// One stream-writing GEMM of the encoder, x = A.W^T + bias + resid with the per-tile row statistics, in either
// representation of the stream: fp32 + f16 copy (pair == 0: resid_hi / resid_lo are summed to the fp32 residual on the host
// side of the kernel, out_x and out_hi are written) or the f16 pair (pair == 1: out_hi / out_lo).  stats: M * 24 * 2 floats.
DLIMG_API int dlimg_amd_test_gemm_stream(int M, int D, int K, uint16_t const* A, uint16_t const* W, float const* bias,
                                         uint16_t const* resid_hi, uint16_t const* resid_lo, int pair, float* out_x,
                                         uint16_t* out_hi, uint16_t* out_lo, float* out_stats) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(M > 0 && D > 0 && K > 0 && A && W && out_hi && out_stats);
        DLIMG_ASSERT((resid_hi != nullptr) == (resid_lo != nullptr));
        DLIMG_ASSERT(pair ? out_lo != nullptr : out_x != nullptr);
        const size_t n = (size_t)M * D;
        Upload<half_t> a(reinterpret_cast<half_t const*>(A), (size_t)M * K);
        Upload<half_t> w(reinterpret_cast<half_t const*>(W), (size_t)D * K);
        Upload<float> b(bias, bias ? D : 0);
        Upload<half_t> rh(reinterpret_cast<half_t const*>(resid_hi), resid_hi ? n : 0);
        Upload<half_t> rl(reinterpret_cast<half_t const*>(resid_lo), resid_lo ? n : 0);
        std::vector<float> sum(resid_hi && !pair ? n : 0);
        for (size_t i = 0; i < sum.size(); ++i)
            sum[i] = (float)reinterpret_cast<half_t const*>(resid_hi)[i] + (float)reinterpret_cast<half_t const*>(resid_lo)[i];
        Upload<float> r(sum.data(), sum.size());
        DeviceBuffer<float> x(pair ? 0 : n), stats((size_t)M * 24 * 2);
        DeviceBuffer<half_t> hi(n), lo(pair ? n : 0);
        HIP_CHECK(hipMemset(stats.get(), 0, (size_t)M * 24 * 2 * sizeof(float)));
        k::GemmArgs g;
        g.A = a.get(); g.lda = K; g.W = w.get(); g.ldw = K; g.bias = bias ? b.get() : nullptr; g.resid_mod = M;
        if (resid_hi && pair) { g.resid_h = rh.get(); g.resid_l = rl.get(); g.ldrs = D; }
        else if (resid_hi) { g.resid = r.get(); g.ldr = D; }
        if (pair) g.out_l = lo.get();
        else { g.out_f32 = x.get(); g.ldc32 = D; }
        g.out_h = hi.get(); g.ldc16 = D; g.stats_out = stats.get();
        g.M = M; g.N = D; g.K = K; g.shared_gpu = true;
        apply_forced_tile(g);
        k::gemm(g, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_x, x.get(), pair ? 0 : n);
        download(reinterpret_cast<half_t*>(out_hi), hi.get(), n);
        download(reinterpret_cast<half_t*>(out_lo), lo.get(), pair ? n : 0);
        download(out_stats, stats.get(), (size_t)M * 24 * 2);
    });
}

DLIMG_API int dlimg_amd_test_layernorm(float const* x, float const* w, float const* b, float eps, int rows, int dim,
                                       int act, float* out_f32, uint16_t* out_f16) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(rows > 0 && dim > 0 && x && w && b);
        const size_t n = (size_t)rows * dim;
        Upload<float> dx(x, n), dw(w, dim), db(b, dim);
        DeviceBuffer<float> o32(out_f32 ? n : 0);
        DeviceBuffer<half_t> o16(out_f16 ? n : 0);
        k::layernorm(dx.get(), dw.get(), db.get(), eps, rows, dim, act, out_f32 ? o32.get() : nullptr,
                     out_f16 ? o16.get() : nullptr, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_f32, o32.get(), out_f32 ? n : 0);
        download(reinterpret_cast<half_t*>(out_f16), o16.get(), out_f16 ? n : 0);
    });
}

DLIMG_API int dlimg_amd_test_attention(int global, uint16_t const* qkv, float const* qkv_bias, float const* rel_h,
                                       float const* rel_w, int batch, int heads, int hd, uint16_t* out) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(batch > 0 && heads > 0 && qkv && rel_h && rel_w && out);
        const int D = heads * hd;
        const int span = global ? 64 : 14;
        const size_t rows = (size_t)batch * kTokens;
        Upload<half_t> dq(reinterpret_cast<half_t const*>(qkv), rows * 3 * D);
        Upload<float> db(qkv_bias, qkv_bias ? (size_t)3 * D : 0);
        Upload<float> dh(rel_h, (size_t)(2 * span - 1) * hd), dw(rel_w, (size_t)(2 * span - 1) * hd);
        DeviceBuffer<half_t> o(rows * D);
        HIP_CHECK(hipMemset(o.get(), 0, rows * D * sizeof(half_t)));
        HIP_CHECK(hipDeviceSynchronize());
        // a stream of the calling thread's own (not the process-wide null stream): calls from several host threads then
        // overlap on the GPU as the execution lanes' kernels do (tests/test_gpu_concurrency.py stress test)
        struct OwnStream {
            hipStream_t s = nullptr;
            ~OwnStream() { if (s) (void)hipStreamDestroy(s); }
        };
        thread_local OwnStream own;
        if (!own.s) HIP_CHECK(hipStreamCreateWithFlags(&own.s, hipStreamNonBlocking));
        if (global) {
            // the kernel's contract (kernels.hpp): q columns and rel-pos tables arrive pre-scaled -- here on the host, as
            // SamModel does with the weights that produce them
            const size_t n = (size_t)(2 * span - 1) * hd;
            const float qs = k::attention_global_q_scale(hd), rs = k::attention_global_rel_scale(hd);
            std::vector<half_t> hq(reinterpret_cast<half_t const*>(qkv), reinterpret_cast<half_t const*>(qkv) + rows * 3 * D);
            for (size_t r = 0; r < rows; ++r)
                for (int c = 0; c < D; ++c) hq[r * 3 * D + c] = (half_t)((float)hq[r * 3 * D + c] * qs);
            std::vector<half_t> hh(n), hw(n);
            for (size_t i = 0; i < n; ++i) {
                hh[i] = (half_t)(rel_h[i] * rs);
                hw[i] = (half_t)(rel_w[i] * rs);
            }
            Upload<half_t> dqs(hq.data(), hq.size()), dh16(hh.data(), n), dw16(hw.data(), n);
            HIP_CHECK(hipDeviceSynchronize());
            k::attention_global(dqs.get(), dh16.get(), dw16.get(), o.get(), batch, heads, hd, own.s);
            HIP_CHECK(hipStreamSynchronize(own.s));      // the uploads go out of scope below
        } else {
            DLIMG_ASSERT(qkv_bias != nullptr);
            // the kernel takes the padding values and the tables as f16, as SamModel converts them when it loads the weights
            const size_t n = (size_t)(2 * span - 1) * hd;
            DeviceBuffer<half_t> db16((size_t)3 * D), dh16(n), dw16(n);
            k::cast_f16(db.get(), db16.get(), (size_t)3 * D, own.s);
            k::cast_f16(dh.get(), dh16.get(), n, own.s);
            k::cast_f16(dw.get(), dw16.get(), n, own.s);
            k::attention_window(dq.get(), db16.get(), dh16.get(), dw16.get(), o.get(), batch, heads, hd, own.s);
            HIP_CHECK(hipStreamSynchronize(own.s));
        }
        HIP_CHECK(hipStreamSynchronize(own.s));
        download(reinterpret_cast<half_t*>(out), o.get(), rows * D);
    });
}

DLIMG_API int dlimg_amd_test_resize(uint8_t const* pixels, int width, int height, int stride, int channels, int out_w,
                                    int out_h, uint8_t* out_pixels) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(pixels && out_pixels && width > 0 && height > 0 && out_w > 0 && out_h > 0);
        const int C = channel_bytes(channels);
        AxisTable tx = make_axis_table(width, out_w), ty = make_axis_table(height, out_h);
        float lut[256];
        srgb_decode_table(lut);
        Upload<uint8_t> src(pixels, (size_t)stride * height);
        Upload<int> xf(tx.first.data(), tx.first.size()), xc(tx.count.data(), tx.count.size());
        Upload<int> yf(ty.first.data(), ty.first.size()), yc(ty.count.data(), ty.count.size());
        Upload<float> xk(tx.coef.data(), tx.coef.size()), yk(ty.coef.data(), ty.coef.size()), dlut(lut, 256);
        Upload<uint32_t> enc(kSrgbEncodeTab4, 104);
        DeviceBuffer<float> tmp((size_t)height * out_w * C);
        DeviceBuffer<uint8_t> dst((size_t)out_w * out_h * C);
        k::ResizeAxis ax{xf.get(), xc.get(), xk.get(), tx.taps, out_w}, ay{yf.get(), yc.get(), yk.get(), ty.taps, out_h};
        k::resize_srgb(src.get(), width, height, stride, C, ax, ay, dlut.get(), enc.get(), tmp.get(), dst.get(), nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        download(out_pixels, dst.get(), (size_t)out_w * out_h * C);
    });
}

DLIMG_API int dlimg_amd_bench_prepost(int batch, int iters, int working_set_mb, double* out_pre_ms, double* out_post_ms) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(batch > 0 && batch <= 16 && iters > 0 && working_set_mb >= 0 && working_set_mb <= 8192 && out_pre_ms && out_post_ms);
        // Successive launches rotate over a ring of DISTINCT input and output sets whose footprint is `working_set_mb`
        // (default 768 MB, three times the 256 MB Infinity Cache): by the time a set comes round again nothing of it is
        // cache-resident, so bytes / time is an HBM rate.  [Re-running one 160 MiB set back to back, as this hook did
        // before, measures the Infinity Cache: 6.35 TB/s "HBM" where a float4 copy from HBM reaches 6.29.]
        const size_t want = (size_t)(working_set_mb > 0 ? working_set_mb : 768) << 20;
        const size_t img_bytes = (size_t)kImageSize * kImageSize * 4, patch_elems = (size_t)kTokens * kPatchK;
        const size_t mask_bytes = (size_t)kImageSize * kImageSize, logit_elems = (size_t)4 * kLowRes * kLowRes;
        const size_t pre_set = batch * (img_bytes + patch_elems * 2), post_set = batch * (logit_elems * 4 + mask_bytes);
        const int pre_sets = (int)std::max<size_t>(1, (want + pre_set - 1) / pre_set);
        const int post_sets = (int)std::max<size_t>(1, (want + post_set - 1) / post_set);
        std::vector<uint8_t> himg(img_bytes);
        uint32_t seed = 99u;
        for (auto& v : himg) { seed = seed * 1664525u + 1013904223u; v = (uint8_t)(seed >> 24); }
        std::vector<float> hlog(logit_elems);
        for (auto& v : hlog) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; }
        std::vector<float> hiou = {0.1f, 0.7f, 0.5f, 0.3f};
        DeviceBuffer<uint8_t> imgs((size_t)pre_sets * batch * img_bytes), masks((size_t)post_sets * batch * mask_bytes);
        DeviceBuffer<half_t> patches((size_t)pre_sets * batch * patch_elems);
        DeviceBuffer<float> logits((size_t)post_sets * batch * logit_elems), iou((size_t)post_sets * batch * 4);
        HIP_CHECK(hipMemcpy(imgs.get(), himg.data(), img_bytes, hipMemcpyHostToDevice));
        for (size_t i = 1; i < (size_t)pre_sets * batch; ++i)
            HIP_CHECK(hipMemcpy(imgs.get() + i * img_bytes, imgs.get(), img_bytes, hipMemcpyDeviceToDevice));
        HIP_CHECK(hipMemcpy(logits.get(), hlog.data(), logit_elems * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)post_sets * batch; ++i) {
            if (i) HIP_CHECK(hipMemcpy(logits.get() + i * logit_elems, logits.get(), logit_elems * 4, hipMemcpyDeviceToDevice));
            HIP_CHECK(hipMemcpy(iou.get() + i * 4, hiou.data(), 16, hipMemcpyHostToDevice));
        }
        std::vector<std::vector<k::PreImage>> pre(pre_sets, std::vector<k::PreImage>(batch));
        std::vector<std::vector<k::PostJob>> post(post_sets, std::vector<k::PostJob>(batch));
        for (int sidx = 0; sidx < pre_sets; ++sidx)
            for (int i = 0; i < batch; ++i) {
                const size_t n = (size_t)sidx * batch + i;
                pre[sidx][i] = k::PreImage{imgs.get() + n * img_bytes, kImageSize, kImageSize, kImageSize * 4, 4, patches.get() + n * patch_elems};
            }
        for (int sidx = 0; sidx < post_sets; ++sidx)
            for (int i = 0; i < batch; ++i) {
                const size_t n = (size_t)sidx * batch + i;
                post[sidx][i] = k::PostJob{logits.get() + n * logit_elems, iou.get() + n * 4, masks.get() + n * mask_bytes,
                                           kImageSize, kImageSize, kImageSize, kImageSize};
            }
        hipStream_t st;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        auto time = [&](auto&& launch, int sets) {
            int at = 0;
            for (int i = 0; i < std::max(3, sets); ++i) launch(at++ % sets);      // one whole turn of the ring: everything touched once
            HIP_CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch(at++ % sets);
            HIP_CHECK(hipEventRecord(e1, st));
            HIP_CHECK(hipStreamSynchronize(st));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            return (double)ms / iters;
        };
        *out_pre_ms = time([&](int sidx) { k::preprocess_batch(pre[sidx].data(), batch, st); }, pre_sets);
        *out_post_ms = time([&](int sidx) { k::postprocess_masks(post[sidx].data(), batch, st); }, post_sets);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
    });
}

DLIMG_API int dlimg_amd_bench_attention(int global, int batch, int heads, int hd, int iters, double* out_ms) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(batch > 0 && batch <= 16 && heads > 0 && heads <= 32 && (hd == 64 || hd == 80) && iters > 0 && out_ms);
        const int D = heads * hd, span = global ? 64 : 14;
        const size_t rows = (size_t)batch * kTokens, nrel = (size_t)(2 * span - 1) * hd;
        std::vector<half_t> hq(rows * 3 * D);
        std::vector<float> hb(3 * D), hrel(nrel);
        uint32_t seed = 777u;
        auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hq) v = (half_t)(rnd() * 1.5f);
        for (auto& v : hb) v = rnd() * 0.2f;
        for (auto& v : hrel) v = rnd() * 0.3f;
        if (global) {
            // the global kernel's contract (kernels.hpp): q columns times log2(e) / sqrt(hd), rel-pos tables times sqrt(hd), as
            // SamWeights prepares them -- unscaled operands give scores ~6 x hotter than the product's, i.e. another frequency
            // of the lazy-maximum raise and other timings than the product's
            const float qs = k::attention_global_q_scale(hd), rs = k::attention_global_rel_scale(hd);
            for (size_t r = 0; r < rows; ++r)
                for (int c = 0; c < D; ++c) hq[r * 3 * D + c] = (half_t)((float)hq[r * 3 * D + c] * qs);
            for (auto& v : hrel) v *= rs;
        }
        Upload<half_t> dq(hq.data(), hq.size());
        Upload<float> db(hb.data(), hb.size()), dh(hrel.data(), nrel), dw(hrel.data(), nrel);
        DeviceBuffer<half_t> o(rows * D), dh16(nrel), dw16(nrel), db16((size_t)3 * D);
        hipStream_t st;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        k::cast_f16(dh.get(), dh16.get(), nrel, st);
        k::cast_f16(dw.get(), dw16.get(), nrel, st);
        k::cast_f16(db.get(), db16.get(), (size_t)3 * D, st);
        auto launch = [&] {
            if (global) k::attention_global(dq.get(), dh16.get(), dw16.get(), o.get(), batch, heads, hd, st);
            else k::attention_window(dq.get(), db16.get(), dh16.get(), dw16.get(), o.get(), batch, heads, hd, st);
        };
        for (int i = 0; i < 3; ++i) launch();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch();
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *out_ms = ms / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
    });
}

DLIMG_API int dlimg_amd_bench_gemm(int M, int N, int K, int act, int flavour, int iters, double* out_ms) {
    return dlimg_amd_bench_gemm_streams(M, N, K, act, flavour, -1, 0, 1, iters, out_ms);
}

DLIMG_API int dlimg_amd_bench_gemm_streams(int M, int N, int K, int act, int flavour, int tile, int shared, int streams,
                                           int iters, double* out_ms) {
    return dlimg_amd_bench_gemm_stamps(M, N, K, act, flavour, tile, shared, streams, iters, out_ms, nullptr, 0);
}

DLIMG_API int dlimg_amd_bench_gemm_stamps(int M, int N, int K, int act, int flavour, int tile, int shared, int streams,
                                          int iters, double* out_ms, unsigned long long* out_stamps, int max_groups) {
    return guarded([&] {
        require_gpu();
        DLIMG_ASSERT(M > 0 && N > 0 && K > 0 && iters > 0 && out_ms && streams >= 1 && streams <= 8);
        std::vector<half_t> ha((size_t)M * K), hw((size_t)N * K);
        uint32_t seed = 12345u;
        auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
#ifdef DLIMG_TUNING
        const bool zeros = std::getenv("DLIMGEDIT_BENCH_ZERO") != nullptr;     // clock experiment (tuning build): all-zero operands
#else
        const bool zeros = false;
#endif
        for (auto& v : ha) v = zeros ? (half_t)0.f : (half_t)rnd();
        for (auto& v : hw) v = zeros ? (half_t)0.f : (half_t)(rnd() * 0.05f);
        Upload<half_t> a(ha.data(), ha.size()), w(hw.data(), hw.size());
        DeviceBuffer<half_t> o((size_t)M * N);
        DeviceBuffer<float> o32, colsum, bias, stats;
        DeviceBuffer<half_t> pair_in, pair_out;
        k::GemmArgs g;
        g.A = a.get(); g.lda = K; g.W = w.get(); g.ldw = K; g.out_h = o.get(); g.ldc16 = N;
        g.M = M; g.N = N; g.K = K; g.act = act; g.tile = tile; g.shared_gpu = shared != 0;
        apply_forced_tile(g);
        // flavour 0: f16 output only; 1: LayerNorm folded in; 2: residual-stream writer (bias + fp32 residual in
        // place); 3: the same plus the f16 copy of the stream and its row statistics; 4: f16 output with bias;
        // 5: the encoder's stream writer (bias + the stream as an f16 pair in, pair out, row statistics); 6: 5 without statistics
        DLIMG_ASSERT(flavour >= 0 && flavour <= 6);
        std::vector<float> cs(N, 0.5f);
        bias.reserve(N);
        HIP_CHECK(hipMemcpy(bias.get(), cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
        if (flavour == 1) {
            DLIMG_ASSERT(K % 128 == 0);
            const int groups = K / 128;              // as left by a producer with 128-column tiles
            std::vector<float> st((size_t)groups * M * 2);
            for (size_t i = 0; i < st.size(); i += 2) { st[i] = rnd(); st[i + 1] = 128.f; }
            stats.reserve(st.size()); colsum.reserve(N);
            HIP_CHECK(hipMemcpy(stats.get(), st.data(), st.size() * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(colsum.get(), cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
            g.ln_stats = stats.get(); g.ln_groups = groups; g.ln_colsum = colsum.get(); g.ln_eps = 1e-6f;
            g.bias = bias.get();
        } else if (flavour == 4) {
            g.bias = bias.get();
        } else if (flavour >= 5) {
            pair_in.reserve((size_t)M * N * 2); pair_out.reserve((size_t)M * N);
            HIP_CHECK(hipMemset(pair_in.get(), 0, (size_t)M * N * 4));
            g.bias = bias.get(); g.resid_h = pair_in.get(); g.resid_l = pair_in.get() + (size_t)M * N; g.ldrs = N; g.resid_mod = M;
            g.out_l = pair_out.get();
            if (flavour == 5) { stats.reserve((size_t)M * 24 * 2); g.stats_out = stats.get(); }
        } else if (flavour >= 2) {
            o32.reserve((size_t)M * N);
            HIP_CHECK(hipMemset(o32.get(), 0, (size_t)M * N * 4));
            g.bias = bias.get(); g.resid = o32.get(); g.ldr = N; g.resid_mod = M; g.out_f32 = o32.get(); g.ldc32 = N;
            g.out_h = nullptr;
            if (flavour == 3) {
                stats.reserve((size_t)M * 24 * 2);
                g.out_h = o.get(); g.stats_out = stats.get();
            }
        }
        // `streams` concurrent copies of the problem (own outputs, shared operands), launched round-robin: the regime
        // of the execution lanes, where kernels of different images share the chip
        DeviceBuffer<unsigned long long> stamps;
#ifndef DLIMG_TUNING
        if (out_stamps && max_groups > 0)
            throw Exception("in-kernel stamps exist only in the tuning build (python -m dlimgedit_amd.build --tuning, DLIMGEDIT_TUNING_LIB=1)");
#endif
        if (out_stamps && max_groups > 0) {
            stamps.reserve((size_t)max_groups * 4);
            HIP_CHECK(hipMemset(stamps.get(), 0, (size_t)max_groups * 4 * sizeof(unsigned long long)));
            g.stamps = stamps.get();             // stream 0's kernels; the last launch's values remain
        }
        std::vector<hipStream_t> ss(streams);
        std::vector<k::GemmArgs> gs(streams, g);
        for (int i = 1; i < streams; ++i) gs[i].stamps = nullptr;
        std::vector<DeviceBuffer<half_t>> outs(streams);
        std::vector<DeviceBuffer<float>> outs32(streams), stat_bufs(streams);
        for (int i = 0; i < streams; ++i) {
            HIP_CHECK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
            if (i == 0) continue;
            if (g.out_h) { outs[i].reserve((size_t)M * N); gs[i].out_h = outs[i].get(); }
            if (g.out_f32) {
                outs32[i].reserve((size_t)M * N);
                HIP_CHECK(hipMemset(outs32[i].get(), 0, (size_t)M * N * 4));
                gs[i].out_f32 = outs32[i].get(); gs[i].resid = outs32[i].get();
            }
            if (g.stats_out) { stat_bufs[i].reserve((size_t)M * 24 * 2); gs[i].stats_out = stat_bufs[i].get(); }
        }
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3 * streams; ++i) k::gemm(gs[i % streams], ss[i % streams]);
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<hipEvent_t> fork(1), join(streams);
        HIP_CHECK(hipEventCreateWithFlags(&fork[0], hipEventDisableTiming));
        for (auto& e : join) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(e0, ss[0]));
        HIP_CHECK(hipEventRecord(fork[0], ss[0]));
        for (int i = 1; i < streams; ++i) HIP_CHECK(hipStreamWaitEvent(ss[i], fork[0], 0));
        for (int i = 0; i < iters * streams; ++i) k::gemm(gs[i % streams], ss[i % streams]);
        for (int i = 1; i < streams; ++i) {
            HIP_CHECK(hipEventRecord(join[i], ss[i]));
            HIP_CHECK(hipStreamWaitEvent(ss[0], join[i], 0));
        }
        HIP_CHECK(hipEventRecord(e1, ss[0]));
        HIP_CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *out_ms = ms / (iters * streams);        // per GEMM, aggregate over the streams
        if (out_stamps && max_groups > 0)
            HIP_CHECK(hipMemcpy(out_stamps, stamps.get(), (size_t)max_groups * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipEventDestroy(fork[0]);
        for (auto e : join) (void)hipEventDestroy(e);
        for (auto st : ss) (void)hipStreamDestroy(st);
    });
}

}  // extern "C"

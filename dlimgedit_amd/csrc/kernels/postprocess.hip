// K16  mask post-processing: low-res logits -> binary mask at the original image resolution.
//
// Fuses what the reference does in two places:
//   * in the decoder ONNX graph (SamOnnxModel.mask_postprocessing, exported by
//     /root/reference/script/export_models.py:29-43): bilinear 256->1024 (align_corners=False),
//     crop to the pre-padding size, bilinear to (H, W);
//   * write_mask_image (/root/reference/src/segmentation.cpp:108-116): `> 0 ? 255 : 0`;
//   * and, in single-mask mode, SamOnnxModel.select_masks (argmax of iou + (n_pts-2.5)*[1000,0,0,0]).
// Neither 4 MiB fp32 intermediate is materialised: each output pixel evaluates the two stages
// analytically (up to 4 stage-1 samples of 4 taps each).
//
// Every multiply/add is an explicitly rounded fp32 operation (__fmul_rn/__fadd_rn, no fma
// contraction) in the order of oracle/sam_oracle.py:bilinear_resize, so the thresholded mask is
// bit-identical to the oracle's for identical logits.
#include "device_common.hpp"
#include "kernels.hpp"

// bit-exactness contract with the oracle: no mul+add contraction anywhere in this file
#pragma clang fp contract(off)

namespace dlimg {
namespace {

constexpr int LOW = 256;
constexpr int FULL = 1024;
constexpr int MAX_JOBS = 16;

struct JobPack { k::PostJob j[MAX_JOBS]; };

struct Tap { int i0, i1; float w0, w1; };

// half-pixel bilinear source coordinates, align_corners=False (oracle: _lin_coeffs)
DLIMG_DEVICE Tap make_tap(int dst, float scale, int in_size) {
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    src = fmaxf(src, 0.f);
    Tap t;
    t.i0 = min((int)src, in_size - 1);
    t.i1 = min(t.i0 + 1, in_size - 1);
    float l1 = __fsub_rn(src, (float)t.i0);
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    t.w1 = l1;
    t.w0 = __fsub_rn(1.0f, l1);
    return t;
}

DLIMG_DEVICE float lerp2(float a00, float a01, float a10, float a11, const Tap& ty, const Tap& tx) {
    const float top = __fadd_rn(__fmul_rn(a00, tx.w0), __fmul_rn(a01, tx.w1));
    const float bot = __fadd_rn(__fmul_rn(a10, tx.w0), __fmul_rn(a11, tx.w1));
    return __fadd_rn(__fmul_rn(ty.w0, top), __fmul_rn(ty.w1, bot));
}

// value of the 1024x1024 upsampled plane at (Y, X)
DLIMG_DEVICE float stage1(const float* __restrict__ low, int Y, int X) {
    const Tap ty = make_tap(Y, 0.25f, LOW), tx = make_tap(X, 0.25f, LOW);
    const float* r0 = low + ty.i0 * LOW;
    const float* r1 = low + ty.i1 * LOW;
    return lerp2(r0[tx.i0], r0[tx.i1], r1[tx.i0], r1[tx.i1], ty, tx);
}

// Fast form for masks whose second stage is the identity (longest side 1024: BASELINE configs 1-4).  The first stage is an
// exact 4x up-sampling, so inside the plane the taps of output pixel 4j + p are the same for every j: columns (j-1, j)
// with weights (0.375, 0.625) and (0.125, 0.875) for p = 0, 1, columns (j, j+1) with (0.875, 0.125) and (0.625, 0.375) for
// p = 2, 3 -- the values make_tap computes (0.25 * (x + 0.5) - 0.5 is exact in fp32), rows alike.  A thread therefore
// produces a 4 x 4 block of the mask from a 3 x 3 neighbourhood of logits: 9 loads and ~110 rounded operations for 16
// pixels (the per-pixel form: 64 loads, ~350), the same multiplications and additions in the same order, so the mask
// is bit-identical.  Blocks on the border of the logit plane (clamped taps) and ragged right / bottom edges take the
// per-pixel path.  16 masks per launch: 38 us (0.55 TB/s, VALU- and L1-bound) -> 17 us (1.2 TB/s, latency-bound).
__global__ __launch_bounds__(256) void postprocess_identity_kernel(JobPack pack) {
    const k::PostJob job = pack.j[blockIdx.y];
    const float* low = job.src;
    if (job.select_iou) {
        float best = __fadd_rn(job.select_iou[0], __fmul_rn(-0.5f, 1000.0f));
        int bi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (job.select_iou[i] > best) { best = job.select_iou[i]; bi = i; }
        low += (size_t)bi * LOW * LOW;
    }
    const int W = job.out_w, H = job.out_h;
    const int bw = (W + 3) >> 2, bh = (H + 3) >> 2;
    // a thread takes FOUR vertically adjacent 4 x 4 blocks (16 rows x 4 columns of the mask): a quarter of the waves, and the
    // 18 logits of its 6 x 3 neighbourhood are requested together instead of 9 per wave (r04: 16 masks per launch, inputs
    // and outputs rotating through 768 MB so that they come from HBM: 21.5 us -> see DESIGN section 6)
    const int bh4 = (bh + 3) >> 2;
    const long total = (long)bw * bh4;
    const float wa[4] = {0.375f, 0.125f, 0.875f, 0.625f};      // weight of the first tap for p = 0..3
    const float wb[4] = {0.625f, 0.875f, 0.125f, 0.375f};      // weight of the second tap
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int by4 = (int)(g / bw), bx = (int)(g % bw);
        const int ox0 = bx * 4;
        const bool cols_inside = bx >= 1 && bx <= LOW - 2 && ox0 + 4 <= W && (W & 3) == 0 && (((uintptr_t)job.dst) & 3) == 0;
        const bool all_interior = cols_inside && by4 >= 1 && by4 * 4 + 3 <= LOW - 2 && by4 * 16 + 16 <= H;
        if (all_interior) {
            // logit rows by0 - 1 .. by0 + 4 (six), columns bx - 1 .. bx + 1: horizontal interpolation once per row
            const int by0 = by4 * 4;
            float hz[6][4];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float* row = low + (by0 - 1 + r) * LOW + (bx - 1);
                const float v0 = row[0], v1 = row[1], v2 = row[2];
#pragma unroll
                for (int p = 0; p < 4; ++p) hz[r][p] = __fadd_rn(__fmul_rn(p < 2 ? v0 : v1, wa[p]), __fmul_rn(p < 2 ? v1 : v2, wb[p]));
            }
            uint8_t* dst = job.dst + (size_t)(by0 * 4) * W + ox0;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r0 = sub + (q < 2 ? 0 : 1);
                    uint32_t packed = 0;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float val = __fadd_rn(__fmul_rn(wa[q], hz[r0][p]), __fmul_rn(wb[q], hz[r0 + 1][p]));
                        if (val > 0.f) packed |= 0xffu << (8 * p);
                    }
                    *reinterpret_cast<uint32_t*>(dst + (size_t)(sub * 4 + q) * W) = packed;
                }
            continue;
        }
        for (int sub = 0; sub < 4; ++sub) {
            const int by = by4 * 4 + sub;
            if (by >= bh) break;
            const int oy0 = by * 4;
            uint8_t* dst = job.dst + (size_t)oy0 * W + ox0;
            const bool interior = by >= 1 && by <= LOW - 2 && bx >= 1 && bx <= LOW - 2 && oy0 + 4 <= H && ox0 + 4 <= W &&
                                  (((uintptr_t)dst | (uintptr_t)W) & 3) == 0;
            if (interior) {
                float v[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[r][c] = low[(by - 1 + r) * LOW + (bx - 1 + c)];
                float hz[3][4];                  // horizontal interpolation of the three rows at the four x positions
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int c0 = p < 2 ? 0 : 1;
                        hz[r][p] = __fadd_rn(__fmul_rn(v[r][c0], wa[p]), __fmul_rn(v[r][c0 + 1], wb[p]));
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r0 = q < 2 ? 0 : 1;
                    uint32_t packed = 0;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float val = __fadd_rn(__fmul_rn(wa[q], hz[r0][p]), __fmul_rn(wb[q], hz[r0 + 1][p]));
                        if (val > 0.f) packed |= 0xffu << (8 * p);
                    }
                    *reinterpret_cast<uint32_t*>(dst + (size_t)q * W) = packed;
                }
            } else {
                for (int q = 0; q < 4 && oy0 + q < H; ++q)
                    for (int p = 0; p < 4 && ox0 + p < W; ++p) dst[(size_t)q * W + p] = stage1(low, oy0 + q, ox0 + p) > 0.f ? 255 : 0;
            }
        }
    }
}

__global__ __launch_bounds__(256) void postprocess_kernel(JobPack pack) {
    const k::PostJob job = pack.j[blockIdx.y];
    const float* low = job.src;
    if (job.select_iou) {
        // SamOnnxModel.select_masks with num_points = 2: score = iou + (2 - 2.5) * [1000, 0, 0, 0]
        float best = __fadd_rn(job.select_iou[0], __fmul_rn(-0.5f, 1000.0f));
        int bi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (job.select_iou[i] > best) { best = job.select_iou[i]; bi = i; }
        low += (size_t)bi * LOW * LOW;
    }
    const int W = job.out_w, H = job.out_h;
    const int groups_per_row = (W + 3) >> 2;
    const bool identity2 = (job.pre_w == W && job.pre_h == H);
    const float sy = (float)job.pre_h / (float)H, sx = (float)job.pre_w / (float)W;
    const long total = (long)H * groups_per_row;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int oy = (int)(g / groups_per_row);
        const int ox0 = (int)(g % groups_per_row) * 4;
        uint32_t packed = 0;
        Tap t2y;
        if (!identity2) t2y = make_tap(oy, sy, job.pre_h);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ox = ox0 + i;
            if (ox >= W) break;
            float v;
            if (identity2) {
                v = stage1(low, oy, ox);
            } else {
                const Tap t2x = make_tap(ox, sx, job.pre_w);
                v = lerp2(stage1(low, t2y.i0, t2x.i0), stage1(low, t2y.i0, t2x.i1), stage1(low, t2y.i1, t2x.i0),
                          stage1(low, t2y.i1, t2x.i1), t2y, t2x);
            }
            if (v > 0.f) packed |= 0xffu << (8 * i);
        }
        uint8_t* dst = job.dst + (size_t)oy * W + ox0;
        if (ox0 + 4 <= W && (((uintptr_t)dst) & 3) == 0) {
            *reinterpret_cast<uint32_t*>(dst) = packed;
        } else {
            for (int i = 0; i < 4 && ox0 + i < W; ++i) dst[i] = (uint8_t)(packed >> (8 * i));
        }
    }
}

}  // namespace

namespace k {

void postprocess_masks(const PostJob* jobs, int count, hipStream_t s) {
    for (int base = 0; base < count; base += MAX_JOBS) {
        const int n = count - base < MAX_JOBS ? count - base : MAX_JOBS;
        JobPack pack{};
        long max_groups = 1, max_blocks16 = 1;
        bool all_identity = true;
        for (int i = 0; i < n; ++i) {
            const PostJob& j = jobs[base + i];
            if (j.out_w <= 0 || j.out_h <= 0 || j.pre_w <= 0 || j.pre_h <= 0 || j.pre_w > FULL || j.pre_h > FULL)
                throw_error("postprocess_masks: invalid extent");
            if (!j.src || !j.dst) throw_error("postprocess_masks: null buffer");
            pack.j[i] = j;
            long g = (long)j.out_h * ((j.out_w + 3) / 4);
            if (g > max_groups) max_groups = g;
            long b16 = (long)((((j.out_h + 3) / 4) + 3) / 4) * ((j.out_w + 3) / 4);      // 16 x 4 pixels per thread
            if (b16 > max_blocks16) max_blocks16 = b16;
            all_identity = all_identity && j.pre_w == j.out_w && j.pre_h == j.out_h;
        }
        // second stage = identity for every job of the launch: 4 x 4 pixels per thread.  Only for launches of several
        // masks: with one or two masks the 65 536 threads per mask of that form do not fill the chip and the launch takes
        // 5.4 us instead of 3.9; at 16 masks 17 us instead of 38.
        if (all_identity && n >= 4) {
            long blocks = (max_blocks16 + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(postprocess_identity_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, pack);
            continue;
        }
        long blocks = (max_groups + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, pack);
    }
}

}  // namespace k
}  // namespace dlimg

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
// exact 4x up-sampling, so the taps of output pixel 4j + p are the same for every j: columns (j-1, j) with weights
// (0.375, 0.625) and (0.125, 0.875) for p = 0, 1, columns (j, j+1) with (0.875, 0.125) and (0.625, 0.375) for p = 2, 3 --
// the values make_tap computes (0.25 * (x + 0.5) - 0.5 is exact in fp32), rows alike; at the right / bottom border the
// clamped index gives the same sum of two products, at the left / top border (pixels 0, 1) make_tap clamps the
// COORDINATE: taps (0, 1) with weights (1, 0), which the first lane / first row band select.  Same multiplications and
// additions in the same order as the per-pixel form, so the mask is bit-identical (tests: test_postprocess_bit_exact).
//
// Against the HBM roofline (r05): a lane owns 16 consecutive pixels of 8 rows.  It reads four logits per logit row as ONE
// 16-byte load -- a wave reads whole 1 KB logit rows -- and gets its two neighbours from the adjacent lanes; it writes
// 16 bytes per pixel row, so a wave-instruction stores one whole 1 KB row of the mask (r04: 4 bytes per lane, 256 B per
// instruction, three scalar loads per row, 0.10 of the HBM peak at 16 masks per launch).  A workgroup = 4 waves = 32
// consecutive pixel rows; the workgroups of a mask are consecutive ids of ONE XCD (xcd_remap), whose L2 serves the two
// logit rows neighbouring bands share.  Ragged extents (W < 1024 or H not a multiple of 8) compute the same values and
// only store what is inside.
constexpr int ID_ROWS = 8;                       // pixel rows per lane (two logit rows + one above, one below)
__global__ __launch_bounds__(256) void postprocess_identity_kernel(JobPack pack, int bands_per_mask) {
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const k::PostJob job = pack.j[logical / bands_per_mask];
    const int band = (logical % bands_per_mask) * 4 + (threadIdx.x >> 6);      // band of ID_ROWS pixel rows
    const int lane = lane_id();
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
    const int oy0 = band * ID_ROWS, ox0 = lane * 16;
    if (oy0 >= H) return;                        // (wave-uniform)
    constexpr int LR = ID_ROWS / 4;              // logit rows of the band
    const int ly0 = band * LR;
    // horizontal pass: hz[r][x] = value of logit row ly0 - 1 + r at the lane's 16 pixel columns
    const float wa[4] = {0.375f, 0.125f, 0.875f, 0.625f};      // weight of the first tap for p = 0..3
    const float wb[4] = {0.625f, 0.875f, 0.125f, 0.375f};      // weight of the second tap
    const bool first = lane == 0, last = lane == 63;
    float hz[LR + 2][16];
#pragma unroll
    for (int r = 0; r < LR + 2; ++r) {
        int ly = ly0 - 1 + r;
        ly = ly < 0 ? 0 : (ly > LOW - 1 ? LOW - 1 : ly);
        const float4_t c = *reinterpret_cast<const float4_t*>(low + ly * LOW + lane * 4);
        float left = __shfl_up(c[3], 1, 64), right = __shfl_down(c[0], 1, 64);
        if (last) right = c[3];                  // clamped index 256 -> 255
        const float v[6] = {left, c[0], c[1], c[2], c[3], right};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float a = v[j + (p < 2 ? 0 : 1)], bq = v[j + (p < 2 ? 1 : 2)], fa = wa[p], fb = wb[p];
                if (j == 0 && p < 2 && first) { a = c[0]; bq = c[1]; fa = 1.0f; fb = 0.0f; }     // pixels 0, 1: taps (0, 1), weights (1, 0)
                hz[r][j * 4 + p] = __fadd_rn(__fmul_rn(a, fa), __fmul_rn(bq, fb));
            }
    }
    const bool wide = (W & 15) == 0 && (((uintptr_t)job.dst) & 15) == 0 && ox0 + 16 <= W;
#pragma unroll
    for (int y = 0; y < ID_ROWS; ++y) {
        const int oy = oy0 + y;
        if (oy >= H) break;
        const int q = y & 3, r0 = (y >> 2) + (q < 2 ? 0 : 1);      // rows (ly - 1, ly) for q = 0, 1, (ly, ly + 1) for q = 2, 3
        // pixel rows 0, 1 (first band only; y < 2 is a compile-time fact in the unrolled loop): taps (0, 1), weights (1, 0)
        const bool top_rows = y < 2 && band == 0;
        const float fa = top_rows ? 1.0f : wa[q], fb = top_rows ? 0.0f : wb[q];
        uint32_t packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const float top = top_rows ? hz[1][x] : hz[r0][x], bot = top_rows ? hz[2][x] : hz[r0 + 1][x];
            const float val = __fadd_rn(__fmul_rn(fa, top), __fmul_rn(fb, bot));
            if (val > 0.f) packed[x >> 2] |= 0xffu << (8 * (x & 3));
        }
        uint8_t* dst = job.dst + (size_t)oy * W + ox0;
        if (wide) {
            *reinterpret_cast<uint4_t*>(dst) = uint4_t{packed[0], packed[1], packed[2], packed[3]};
        } else {
            for (int x = 0; x < 16 && ox0 + x < W; ++x) dst[x] = (uint8_t)(packed[x >> 2] >> (8 * (x & 3)));
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
        long max_groups = 1;
        bool all_identity = true;
        for (int i = 0; i < n; ++i) {
            const PostJob& j = jobs[base + i];
            if (j.out_w <= 0 || j.out_h <= 0 || j.pre_w <= 0 || j.pre_h <= 0 || j.pre_w > FULL || j.pre_h > FULL)
                throw_error("postprocess_masks: invalid extent");
            if (!j.src || !j.dst) throw_error("postprocess_masks: null buffer");
            pack.j[i] = j;
            long g = (long)j.out_h * ((j.out_w + 3) / 4);
            if (g > max_groups) max_groups = g;
            all_identity = all_identity && j.pre_w == j.out_w && j.pre_h == j.out_h;
        }
        // second stage = identity for every job of the launch: the exact-4x form, 16 x 8 pixels per lane.  From three masks
        // per launch on (MI355X, us per launch for 1 / 2 / 3 / 4 / 8 / 16 masks: per-pixel form 4.1 / 5.8 / 8.1 / 9.8 / 17.7 /
        // 34.8, this form 6.0 / 6.4 / 6.5 / 7.0 / 8.0 / 11.7: it has the longer dependent chain and the higher floor)
        if (all_identity && n >= 3) {
            int max_h = 1;
            for (int i = 0; i < n; ++i) max_h = jobs[base + i].out_h > max_h ? jobs[base + i].out_h : max_h;
            const int bands = ((max_h + ID_ROWS - 1) / ID_ROWS + 3) / 4;          // workgroups (4 bands each) per mask
            hipLaunchKernelGGL(postprocess_identity_kernel, dim3((unsigned)(bands * n)), dim3(256), 0, s, pack, bands);
            continue;
        }
        long blocks = (max_groups + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, pack);
    }
}

}  // namespace k
}  // namespace dlimg

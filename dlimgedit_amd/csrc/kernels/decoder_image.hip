// Image-side kernels of the SAM mask decoder that are more than a plain GEMM: the image positions' half of a two-way
// block in ONE launch, and the up-scaling path with the mask product in ONE launch.  In the reference all of this runs
// inside the decoder ONNX graph behind Session::operator() (/root/reference/src/segmentation.cpp:154-158); the published
// definition is SAM's TwoWayAttentionBlock (cross_attn_image_to_token + norm4) and MaskDecoder.output_upscaling.
//
// Both kernels keep the 16x16x32 MFMA's operands in the layout their producer leaves them in: the sum over k of a GEMM
// does not care in which order k is visited, so lane group g of a wave supplies whatever eight k-values it already holds,
// as long as the weight fragment is read with the same permutation.  No transposition through LDS between the stages.
#include "device_common.hpp"
#include "kernels.hpp"

namespace dlimg {
namespace {

constexpr int TOK = 7;
constexpr int DIM = 256;
constexpr int INNER = 128;
constexpr int NTOK_IMG = 4096;

typedef float f32x4 __attribute__((ext_vector_type(4)));

DLIMG_DEVICE f32x4 mfma16(half8_t a, half8_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// sum over the four lanes l, l^16, l^32, l^48 (the four k-groups that share an output row)
DLIMG_DEVICE float sum_over_groups(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------
// keys <- LayerNorm(keys + attention(image -> tokens) Wo + bo), with the f16 copy the next projection reads.
//   attention: an image position attends to the 7 tokens of its prompt, 8 heads x 16 (q: f16 [rows][ldq], token k / v:
//              fp32 [P][7][128]); lane (row m = lane % 16, group g = lane / 16) does heads 2g and 2g + 1 of its row, so its
//              32 outputs ARE the A fragments of the projection for k = 32 g + 8 kk .. + 7, kk = 0 .. 3
//   projection: 128 -> 256 on v_mfma_f32_16x16x32_f16, Wo (f16 [256][128]) in LDS for the workgroup's 64 rows
//   epilogue  : + bias + residual (the keys, fp32), LayerNorm over the 256 columns of a row (64 values in the lane, the
//               rest in the three other groups), fp32 and f16 out
// Replaces image_to_token_attention + the 128 -> 256 GEMM + decoder_keys_norm: three launches and a 1 MB + 4 MB + 4 MB
// round trip through HBM per prompt become one launch that reads q and the keys once and writes the keys once.
constexpr int IU_ROWS = 64;
constexpr int IU_WSTRIDE = INNER + 8;        // halves per row of Wo in LDS: 272 B, so the 16 rows of a fragment read spread over the banks
constexpr size_t IU_LDS = (size_t)DIM * IU_WSTRIDE * 2 + 2 * TOK * INNER * 4 + 3 * DIM * 4;

struct ImageUpdate {
    const half_t* q; int ldq;
    const float* tk; const float* tv;
    const half_t* W; const float* bias;
    const float* ln_w; const float* ln_b; float eps;
    float* keys; half_t* keys_h;
};

__global__ __launch_bounds__(256) void image_update_kernel(ImageUpdate a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* w_lds = reinterpret_cast<half_t*>(smem);                                   // [256][IU_WSTRIDE]
    float* sk = reinterpret_cast<float*>(smem + (size_t)DIM * IU_WSTRIDE * 2);         // [7][128]
    float* sv = sk + TOK * INNER;
    float* cb = sv + TOK * INNER;                                                      // bias | ln_w | ln_b
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * IU_ROWS;
    const int p = (int)(row0 / NTOK_IMG);
    const size_t row = row0 + wave * 16 + m;

    // everything the workgroup reads is requested before the first wait
    half8_t wreg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ch = tid + 256 * i;
        wreg[i] = *reinterpret_cast<const half8_t*>(a.W + (size_t)(ch >> 4) * INNER + (ch & 15) * 8);
    }
    float4_t kvreg[2];
    if (tid < TOK * INNER / 4) {
        kvreg[0] = reinterpret_cast<const float4_t*>(a.tk + (size_t)p * TOK * INNER)[tid];
        kvreg[1] = reinterpret_cast<const float4_t*>(a.tv + (size_t)p * TOK * INNER)[tid];
    }
    const float c0 = a.bias[tid], c1 = a.ln_w[tid], c2 = a.ln_b[tid];
    const half_t* qr = a.q + row * a.ldq + 32 * g;
    half8_t q8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q8[i] = *reinterpret_cast<const half8_t*>(qr + 8 * i);
    f32x4 res[16];
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) res[jt] = *reinterpret_cast<const f32x4*>(a.keys + row * DIM + jt * 16 + 4 * g);

#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ch = tid + 256 * i;
        *reinterpret_cast<half8_t*>(w_lds + (size_t)(ch >> 4) * IU_WSTRIDE + (ch & 15) * 8) = wreg[i];
    }
    if (tid < TOK * INNER / 4) {
        reinterpret_cast<float4_t*>(sk)[tid] = kvreg[0];
        reinterpret_cast<float4_t*>(sv)[tid] = kvreg[1];
    }
    cb[tid] = c0;
    cb[DIM + tid] = c1;
    cb[2 * DIM + tid] = c2;
    __syncthreads();

    // attention of this lane's row over the 7 tokens, heads 2g and 2g + 1
    half8_t afrag[4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int col = (2 * g + hh) * 16;
        float qv[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) { qv[e] = (float)q8[2 * hh][e]; qv[8 + e] = (float)q8[2 * hh + 1][e]; }
        float s[TOK], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            float d = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const float4_t kk = reinterpret_cast<const float4_t*>(sk + j * INNER + col)[e4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d = fmaf(qv[4 * e4 + e], kk[e], d);
            }
            s[j] = d * 0.25f;                                     // 16^-0.5
            mx = fmaxf(mx, s[j]);
        }
        float l = 0.f, o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const float pj = expf(s[j] - mx);
            l += pj;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const float4_t vv = reinterpret_cast<const float4_t*>(sv + j * INNER + col)[e4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * e4 + e] = fmaf(pj, vv[e], o[4 * e4 + e]);
            }
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            afrag[2 * hh][e] = (half_t)(o[e] * inv);
            afrag[2 * hh + 1][e] = (half_t)(o[8 + e] * inv);
        }
    }

    // projection: acc[jt][r] = out[row m][column jt * 16 + 4 g + r]
    f32x4 acc[16];
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const half_t* wl = w_lds + (size_t)m * IU_WSTRIDE + 32 * g;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) {
            const half8_t b = *reinterpret_cast<const half8_t*>(wl + (size_t)jt * 16 * IU_WSTRIDE + 8 * kk);
            acc[jt] = mfma16(b, afrag[kk], acc[jt]);
        }

    // + bias + residual, LayerNorm over the row
    float s1 = 0.f;
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
        const float4_t bv = reinterpret_cast<const float4_t*>(cb)[jt * 4 + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[jt][r] = acc[jt][r] + bv[r] + res[jt][r];
            s1 += acc[jt][r];
        }
    }
    const float mean = sum_over_groups(s1) / (float)DIM;
    float s2 = 0.f;
#pragma unroll
    for (int jt = 0; jt < 16; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[jt][r] -= mean;
            s2 = fmaf(acc[jt][r], acc[jt][r], s2);
        }
    const float rstd = 1.0f / sqrtf(sum_over_groups(s2) / (float)DIM + a.eps);
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
        const float4_t wv = reinterpret_cast<const float4_t*>(cb + DIM)[jt * 4 + g];
        const float4_t bv = reinterpret_cast<const float4_t*>(cb + 2 * DIM)[jt * 4 + g];
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = acc[jt][r] * rstd * wv[r] + bv[r];
        store16_result(a.keys + row * DIM + jt * 16 + 4 * g, y);
        *reinterpret_cast<half4_t*>(a.keys_h + row * DIM + jt * 16 + 4 * g) =
            half4_t{(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
    }
}


// ---------------------------------------------------------------------------------------------
// Up-scaling path and mask product in one launch:
//   logits[p][mk][Y][X] = hyper[p][mk] . GELU(ConvT2(GELU(LN2d(ConvT1(keys)))))[pixel]
// Both transposed convolutions (kernel 2, stride 2) are GEMMs over the image positions whose output columns are the four
// sub-pixels times the output channels (SamWeights: rows of W1 = s1 * 64 + co, rows of W2 = s2 * 32 + c2).  A wave takes 16
// image positions through all of it:
//   stage A  [16][256] x W1^T -> sub-pixels x 64 channels per position (v_mfma_f32_16x16x32_f16, W1 in LDS); a workgroup
//            does two of the four first-stage sub-pixels (half of W1), its twin the other two: twice the workgroups, each
//            with half of the work and of the weights to stage (one prompt: 128 workgroups, 20 -> 13 us)
//   stage B  + bias, LayerNorm over the 64 channels of a sub-pixel (16 values in the lane, the rest in the three other lane
//            groups), GELU, f16: in the accumulator layout these ARE the A fragments of stage C, with the k-permutation
//            k = (2 kk + i / 4) * 16 + 4 g + i % 4 that the W2 fragments are read with
//   stage C  per sub-pixel s1: [16][64] x W2^T -> 4 sub-pixels s2 x 32 channels, + bias, GELU, and the product with the four
//            hyper vectors of the prompt; lane group g ends up with mask g of every pixel (a 4 x 4 transpose-and-add over
//            the lane groups: three exchanges per pixel) and stores it
// Replaces two GEMMs, a LayerNorm launch and mask_logits, and with them 4 + 4 + 2 + 2 + 8 + 8 MB of HBM traffic per prompt
// for intermediate results (up1 fp32, up1 f16, up2 fp32): the kernel reads the keys (2 MB) and writes the logits (1 MB).
constexpr int UP_ROWS = 64;
constexpr int UP_W1STRIDE = DIM + 8;          // halves per row of W1 in LDS (528 B)
constexpr int UP_W2STRIDE = 64 + 8;           // halves per row of W2 in LDS (144 B)
constexpr int UP_CONST = 256 + 64 + 64 + 128 + 128;      // b1 | ln_w | ln_b | b2 | hyper
constexpr int UP_W1ROWS = 128;                // a workgroup takes two of the four first-stage sub-pixels: half of W1
constexpr size_t UP_LDS = (size_t)UP_W1ROWS * UP_W1STRIDE * 2 + (size_t)128 * UP_W2STRIDE * 2 + UP_CONST * 4;

struct Upscale {
    const half_t* keys_h;                      // [P*4096][256]
    const half_t* W1; const float* b1;         // [256][256], [256]
    const float* ln_w; const float* ln_b; float eps;
    const half_t* W2; const float* b2;         // [128][64], [128]
    const float* hyper;                        // [P][4][32]
    float* logits;                             // [P][4][256][256]
    int rows_per_block;
};

__global__ __launch_bounds__(256) void upscale_logits_kernel(Upscale a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* w1 = reinterpret_cast<half_t*>(smem);
    half_t* w2 = w1 + (size_t)UP_W1ROWS * UP_W1STRIDE;
    float* cst = reinterpret_cast<float*>(w2 + (size_t)128 * UP_W2STRIDE);
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int half = blockIdx.x & 1;                                    // first-stage sub-pixels 2 * half and 2 * half + 1
    const size_t row0 = (size_t)(blockIdx.x >> 1) * a.rows_per_block;  // rows_per_block: a multiple of 64 that divides 4096
    const int p = (int)(row0 / NTOK_IMG);
    size_t row = row0 + wave * 16 + m;

    // the wave's own 16 rows of the keys, as A fragments: k = 32 kk + 8 g .. + 7
    half8_t a1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a1[kk] = *reinterpret_cast<const half8_t*>(a.keys_h + row * DIM + 32 * kk + 8 * g);
    // weights and constants -> LDS (this half of W1: 128 rows = 4096 chunks of 16 bytes, 16 per thread)
    {
        half8_t wr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = tid + 256 * i;
            wr[i] = *reinterpret_cast<const half8_t*>(a.W1 + (size_t)(half * UP_W1ROWS + (ch >> 5)) * DIM + (ch & 31) * 8);
        }
        {
            half8_t w2r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = tid + 256 * i;
                w2r[i] = *reinterpret_cast<const half8_t*>(a.W2 + (size_t)(ch >> 3) * 64 + (ch & 7) * 8);
            }
            const float v_b1 = a.b1[tid];
            const float v_ln = tid < 64 ? a.ln_w[tid] : (tid < 128 ? a.ln_b[tid - 64] : a.b2[tid - 128]);
            const float v_hy = tid < 128 ? a.hyper[(size_t)p * 128 + tid] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = tid + 256 * i;
                *reinterpret_cast<half8_t*>(w2 + (size_t)(ch >> 3) * UP_W2STRIDE + (ch & 7) * 8) = w2r[i];
            }
            cst[tid] = v_b1;
            cst[256 + tid] = v_ln;                 // ln_w | ln_b | b2 are contiguous
            if (tid < 128) cst[512 + tid] = v_hy;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = tid + 256 * i;
            *reinterpret_cast<half8_t*>(w1 + (size_t)(ch >> 5) * UP_W1STRIDE + (ch & 31) * 8) = wr[i];
        }
    }
    __syncthreads();

    // with many prompts a workgroup takes several groups of 64 rows, so that the 150 KB of weights are staged once for all
#pragma unroll 1
    for (int rb = 0; rb < a.rows_per_block; rb += UP_ROWS) {
    // the LDS reads below do not depend on rb: without this the compiler hoists all of them out of the loop into registers
    // it does not have (474 spills)
    int not_invariant = 0;
    asm volatile("" : "+v"(not_invariant));
    const half_t* w1i = w1 + not_invariant;
    const half_t* w2i = w2 + not_invariant;
    const float* c_b1 = cst + not_invariant;
    const float* c_lw = c_b1 + 256;
    const float* c_lb = c_b1 + 320;
    const float* c_b2 = c_b1 + 384;
    const float* c_hy = c_b1 + 512;
    if (rb > 0) {
        row = row0 + rb + wave * 16 + m;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a1[kk] = *reinterpret_cast<const half8_t*>(a.keys_h + row * DIM + 32 * kk + 8 * g);
    }
    // stage A: acc1[jt][r] = up1[row m][column half * 128 + jt * 16 + 4 g + r], column = s1 * 64 + co
    f32x4 acc1[8];
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) acc1[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const half_t* wl = w1i + (size_t)m * UP_W1STRIDE + 8 * g;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int jt = 0; jt < 8; ++jt) {
                const half8_t b = *reinterpret_cast<const half8_t*>(wl + (size_t)jt * 16 * UP_W1STRIDE + 32 * kk);
                acc1[jt] = mfma16(b, a1[kk], acc1[jt]);
            }
    }

    const int tok = (int)(row % NTOK_IMG), ty = tok >> 6, tx = tok & 63;
    float* out = a.logits + ((size_t)p * 4 + g) * 65536;
    const half_t* w2l = w2i + (size_t)m * UP_W2STRIDE + 4 * g;
#pragma unroll
    for (int s1l = 0; s1l < 2; ++s1l) {
        const int s1 = 2 * half + s1l;
        // stage B: LayerNorm2d over the 64 channels of sub-pixel s1, GELU, f16 -> A fragments of stage C
        f32x4 v[4];
        float sum = 0.f;
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
            const float4_t bv = reinterpret_cast<const float4_t*>(c_b1)[(4 * s1 + jl) * 4 + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[jl][r] = acc1[4 * s1l + jl][r] + bv[r];
                sum += v[jl][r];
            }
        }
        const float mean = sum_over_groups(sum) / 64.0f;
        float sq = 0.f;
#pragma unroll
        for (int jl = 0; jl < 4; ++jl)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[jl][r] -= mean;
                sq = fmaf(v[jl][r], v[jl][r], sq);
            }
        const float rstd = 1.0f / sqrtf(sum_over_groups(sq) / 64.0f + a.eps);
        half8_t a2[2];
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
            const float4_t wv = reinterpret_cast<const float4_t*>(c_lw)[jl * 4 + g];
            const float4_t bv = reinterpret_cast<const float4_t*>(c_lb)[jl * 4 + g];
            float4_t y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = v[jl][r] * rstd * wv[r] + bv[r];
            y = gelu4(y);
#pragma unroll
            for (int r = 0; r < 4; ++r) a2[jl >> 1][(jl & 1) * 4 + r] = (half_t)y[r];
        }
        // stage C: acc2[jt2][r] = up2[(row, s1)][column jt2 * 16 + 4 g + r], column = s2 * 32 + c2
        f32x4 acc2[8];
#pragma unroll
        for (int jt2 = 0; jt2 < 8; ++jt2) acc2[jt2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt2 = 0; jt2 < 8; ++jt2) {
                const half_t* src = w2l + (size_t)jt2 * 16 * UP_W2STRIDE + 32 * kk;
                const half4_t lo = *reinterpret_cast<const half4_t*>(src), hi = *reinterpret_cast<const half4_t*>(src + 16);
                const half8_t b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc2[jt2] = mfma16(b, a2[kk], acc2[jt2]);
            }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float4_t u[2];
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                const float4_t bv = reinterpret_cast<const float4_t*>(c_b2)[(2 * s2 + jl) * 4 + g];
                u[jl] = gelu4(float4_t{acc2[2 * s2 + jl][0] + bv[0], acc2[2 * s2 + jl][1] + bv[1], acc2[2 * s2 + jl][2] + bv[2],
                                       acc2[2 * s2 + jl][3] + bv[3]});
            }
            float part[4];
#pragma unroll
            for (int mk = 0; mk < 4; ++mk) {
                float d = 0.f;
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) {
                    const float4_t hv = reinterpret_cast<const float4_t*>(c_hy)[mk * 8 + jl * 4 + g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) d = fmaf(u[jl][r], hv[r], d);
                }
                part[mk] = d;
            }
            // sum over the four lane groups, group g keeping mask g: exchange with g ^ 2, then with g ^ 1
            const bool hi_pair = (g & 2) != 0, odd = (g & 1) != 0;
            const float k0 = (hi_pair ? part[2] : part[0]) + __shfl_xor(hi_pair ? part[0] : part[2], 32, 64);
            const float k1 = (hi_pair ? part[3] : part[1]) + __shfl_xor(hi_pair ? part[1] : part[3], 32, 64);
            const float total = (odd ? k1 : k0) + __shfl_xor(odd ? k0 : k1, 16, 64);
            const int Y = 4 * ty + 2 * (s1 >> 1) + (s2 >> 1), X = 4 * tx + 2 * (s1 & 1) + (s2 & 1);
            out[Y * 256 + X] = total;
        }
    }
    }
}

}  // namespace

namespace k {

void image_update(const half_t* q, int ldq, const float* tk, const float* tv, const half_t* W, const float* bias,
                  const float* ln_w, const float* ln_b, float eps, float* keys, half_t* keys_h, int P, hipStream_t s) {
    if (P <= 0) return;
    if (ldq % 8 || (((uintptr_t)q | (uintptr_t)W | (uintptr_t)keys | (uintptr_t)keys_h) & 15))
        throw_error("image_update: operands must be 16-byte aligned");
    static k::LdsOptIn opt_in;
    opt_in.ensure((const void*)image_update_kernel, IU_LDS, "image_update: the device refuses the kernel's LDS size");
    ImageUpdate a{q, ldq, tk, tv, W, bias, ln_w, ln_b, eps, keys, keys_h};
    hipLaunchKernelGGL(image_update_kernel, dim3(P * NTOK_IMG / IU_ROWS), dim3(256), IU_LDS, s, a);
}


void upscale_logits(const half_t* keys_h, const half_t* W1, const float* b1, const float* ln_w, const float* ln_b, float eps,
                    const half_t* W2, const float* b2, const float* hyper, float* logits, int P, hipStream_t s) {
    if (P <= 0) return;
    if (((uintptr_t)keys_h | (uintptr_t)W1 | (uintptr_t)W2) & 15) throw_error("upscale_logits: operands must be 16-byte aligned");
    static k::LdsOptIn opt_in;
    opt_in.ensure((const void*)upscale_logits_kernel, UP_LDS, "upscale_logits: the device refuses the kernel's LDS size");
    // two workgroups per group of rows (one per pair of first-stage sub-pixels), one workgroup per CU (89 KB of LDS): the
    // smallest row groups that still fit the chip in one round
    const int rows_per_block = 2 * P * (NTOK_IMG / 64) <= 256 ? 64 : (2 * P * (NTOK_IMG / 128) <= 256 ? 128 : 256);
    Upscale a{keys_h, W1, b1, ln_w, ln_b, eps, W2, b2, hyper, logits, rows_per_block};
    hipLaunchKernelGGL(upscale_logits_kernel, dim3(2 * P * NTOK_IMG / rows_per_block), dim3(256), UP_LDS, s, a);
}

}  // namespace k
}  // namespace dlimg

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
        *reinterpret_cast<f32x4*>(a.keys + row * DIM + jt * 16 + 4 * g) = y;
        *reinterpret_cast<half4_t*>(a.keys_h + row * DIM + jt * 16 + 4 * g) =
            half4_t{(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
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

}  // namespace k
}  // namespace dlimg

// Token-side kernels of the SAM prompt encoder / mask decoder (everything that is NOT a big GEMM).
// In the reference all of this runs inside the decoder ONNX graph behind Session::operator()
// (/root/reference/src/segmentation.cpp:154-158); the published definition is SAM's PromptEncoder,
// TwoWayTransformer and MaskDecoder.  The token side is 7 tokens per prompt: latency-bound fp32
// VALU work, kept in fp32 end to end.
#include "device_common.hpp"
#include "kernels.hpp"

namespace dlimg {
namespace {

constexpr int TOK = 7;        // iou token + 4 mask tokens + 2 prompt tokens
constexpr int DIM = 256;
constexpr int INNER = 128;    // cross-attention width (downsample 2)
constexpr int HEADS = 8;
constexpr int NTOK_IMG = 4096;

// ---------------------------------------------------------------------------------------------
// Prompt encoder: SamOnnxModel._embed_points applied to the two packed points
// (segmentation.cpp:135-152 packs them; labels 1/-1 for a point, 2/3 for a box).
__global__ __launch_bounds__(256) void prompt_tokens_kernel(const float* __restrict__ coords,
                                                            const float* __restrict__ labels,
                                                            const float* __restrict__ gauss,
                                                            const float* __restrict__ point_embed,
                                                            const float* __restrict__ not_a_point,
                                                            const float* __restrict__ iou_token,
                                                            const float* __restrict__ mask_tokens,
                                                            float* __restrict__ tokens) {
    const int p = blockIdx.x, c = threadIdx.x;
    float* t = tokens + (size_t)p * TOK * DIM;
    t[c] = iou_token[c];
#pragma unroll
    for (int m = 0; m < 4; ++m) t[(1 + m) * DIM + c] = mask_tokens[m * DIM + c];
    const int kf = c & 127;
    for (int i = 0; i < 2; ++i) {
        const float x = (coords[(p * 2 + i) * 2 + 0] + 0.5f) / 1024.0f;
        const float y = (coords[(p * 2 + i) * 2 + 1] + 0.5f) / 1024.0f;
        float v = __fadd_rn(__fmul_rn(2.0f * x - 1.0f, gauss[kf]), __fmul_rn(2.0f * y - 1.0f, gauss[128 + kf]));
        v = 6.283185307179586f * v;
        float e = c < 128 ? sinf(v) : cosf(v);
        const float lab = labels[p * 2 + i];
        if (lab == -1.0f) e = not_a_point[c];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
            if (lab == (float)k4) e += point_embed[k4 * DIM + c];
        t[(5 + i) * DIM + c] = e;
    }
}

// ---------------------------------------------------------------------------------------------
// Y[r,n] = act((X[r,:]+X2[r,:]) . W[n,:] + b[n]) + R[r,n]; one wave per output column.
constexpr int RCHUNK = 8;
__global__ __launch_bounds__(256) void token_linear_kernel(const float* __restrict__ X, const float* __restrict__ X2,
                                                           const float* __restrict__ W, const float* __restrict__ b,
                                                           const float* R, float* Y, int rows, int K, int N, int relu) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    const float* wr = W + (size_t)n * K;
    for (int r0 = 0; r0 < rows; r0 += RCHUNK) {
        float acc[RCHUNK];
#pragma unroll
        for (int r = 0; r < RCHUNK; ++r) acc[r] = 0.f;
        for (int kk = lane; kk < K; kk += 64) {
            const float w = wr[kk];
#pragma unroll
            for (int r = 0; r < RCHUNK; ++r) {
                if (r0 + r < rows) {
                    float x = X[(size_t)(r0 + r) * K + kk];
                    if (X2) x += X2[(size_t)(r0 + r) * K + kk];
                    acc[r] = fmaf(x, w, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RCHUNK; ++r) {
            float v = wave_sum(acc[r]);
            if (lane == 0 && r0 + r < rows) {
                v += b ? b[n] : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                if (R) v += R[(size_t)(r0 + r) * N + n];
                Y[(size_t)(r0 + r) * N + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Self-attention among the 7 tokens (8 heads x 32).
__global__ __launch_bounds__(256) void token_self_attention_kernel(const float* __restrict__ q,
                                                                   const float* __restrict__ kx,
                                                                   const float* __restrict__ v,
                                                                   float* __restrict__ out) {
    __shared__ float sq[TOK * DIM], sk[TOK * DIM], sv[TOK * DIM];
    const int p = blockIdx.x, c = threadIdx.x;
    for (int t = 0; t < TOK; ++t) {
        sq[t * DIM + c] = q[((size_t)p * TOK + t) * DIM + c];
        sk[t * DIM + c] = kx[((size_t)p * TOK + t) * DIM + c];
        sv[t * DIM + c] = v[((size_t)p * TOK + t) * DIM + c];
    }
    __syncthreads();
    const int h0 = (c >> 5) * 32;
    const float scale = 0.17677669529663687f;   // 32^-0.5
    for (int t = 0; t < TOK; ++t) {
        float s[TOK];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            float d = 0.f;
            for (int e = 0; e < 32; ++e) d = fmaf(sq[t * DIM + h0 + e], sk[j * DIM + h0 + e], d);
            s[j] = d * scale;
            m = fmaxf(m, s[j]);
        }
        float l = 0.f, o = 0.f;
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            float pj = expf(s[j] - m);
            l += pj;
            o = fmaf(pj, sv[j * DIM + c], o);
        }
        out[((size_t)p * TOK + t) * DIM + c] = o / l;
    }
}

// ---------------------------------------------------------------------------------------------
// Tokens attend to the 4096 image positions (8 heads x 16).  One workgroup per (prompt, head);
// thread = (query t, key lane kl of 32); per-thread online softmax, then a 32-lane combine.
__global__ __launch_bounds__(256) void token_to_image_kernel(const float* __restrict__ q, const half_t* __restrict__ K,
                                                             int ldk, const half_t* __restrict__ V, int ldv,
                                                             float* __restrict__ out) {
    const int p = blockIdx.x / HEADS, h = blockIdx.x % HEADS;
    const int t = threadIdx.x >> 5, kl = threadIdx.x & 31;
    const bool active = t < TOK;
    const int tq = active ? t : 0;
    float qv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) qv[e] = q[((size_t)p * TOK + tq) * INNER + h * 16 + e] * 0.25f;   // 16^-0.5
    float m = -INFINITY, l = 0.f, o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    const half_t* kb = K + (size_t)p * NTOK_IMG * ldk + h * 16;
    const half_t* vb = V + (size_t)p * NTOK_IMG * ldv + h * 16;
    for (int j = kl; j < NTOK_IMG; j += 32) {
        half8_t k0 = *reinterpret_cast<const half8_t*>(kb + (size_t)j * ldk);
        half8_t k1 = *reinterpret_cast<const half8_t*>(kb + (size_t)j * ldk + 8);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qv[e], (float)k0[e], s);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qv[8 + e], (float)k1[e], s);
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), pj = expf(s - mn);
        half8_t v0 = *reinterpret_cast<const half8_t*>(vb + (size_t)j * ldv);
        half8_t v1 = *reinterpret_cast<const half8_t*>(vb + (size_t)j * ldv + 8);
        l = l * a + pj;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, (float)v0[e], o[e] * a);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[8 + e] = fmaf(pj, (float)v1[e], o[8 + e] * a);
        m = mn;
    }
    // combine the 32 key lanes of this query (lanes of one 32-lane half)
    float M = m;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    const float w = expf(m - M);
    l *= w;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float x = o[e] * w;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        o[e] = x;
    }
    if (active && kl == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) out[((size_t)p * TOK + t) * INNER + h * 16 + e] = o[e] / l;
    }
}

// ---------------------------------------------------------------------------------------------
// Image positions attend to the 7 tokens.  Thread = (image token, head).
__global__ __launch_bounds__(256) void image_to_token_kernel(const half_t* __restrict__ q, int ldq,
                                                             const float* __restrict__ kt, const float* __restrict__ vt,
                                                             half_t* __restrict__ out) {
    __shared__ float sk[TOK * INNER], sv[TOK * INNER];
    const size_t gidx = (size_t)blockIdx.x * 256 + threadIdx.x;     // (prompt*4096 + token)*8 + head
    const int p = (int)(gidx / (NTOK_IMG * HEADS));
    for (int i = threadIdx.x; i < TOK * INNER; i += 256) {
        sk[i] = kt[(size_t)p * TOK * INNER + i];
        sv[i] = vt[(size_t)p * TOK * INNER + i];
    }
    __syncthreads();
    const int h = (int)(gidx % HEADS);
    const size_t row = gidx / HEADS;
    const half_t* qr = q + row * ldq + h * 16;
    half8_t q0 = *reinterpret_cast<const half8_t*>(qr), q1 = *reinterpret_cast<const half8_t*>(qr + 8);
    float qv[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) { qv[e] = (float)q0[e]; qv[8 + e] = (float)q1[e]; }
    float s[TOK], m = -INFINITY;
#pragma unroll
    for (int j = 0; j < TOK; ++j) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) d = fmaf(qv[e], sk[j * INNER + h * 16 + e], d);
        s[j] = d * 0.25f;
        m = fmaxf(m, s[j]);
    }
    float l = 0.f, o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < TOK; ++j) {
        const float pj = expf(s[j] - m);
        l += pj;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = fmaf(pj, sv[j * INNER + h * 16 + e], o[e]);
    }
    const float inv = 1.0f / l;
    half8_t o0, o1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o0[e] = (half_t)(o[e] * inv); o1[e] = (half_t)(o[8 + e] * inv); }
    half_t* orow = out + row * INNER + h * 16;
    *reinterpret_cast<half8_t*>(orow) = o0;
    *reinterpret_cast<half8_t*>(orow + 8) = o1;
}

// ---------------------------------------------------------------------------------------------
// Hyper-network MLPs + IoU head.  grid (P, 5): y = 0..3 mask token MLPs (-> 32), y = 4 IoU head (-> 4).
DLIMG_DEVICE void mlp_layer(const float* x, const float* W, const float* b, float* y, int K, int N, bool relu, int wave,
                            int lane) {
    for (int n = wave; n < N; n += 4) {
        float a = 0.f;
        for (int kk = lane; kk < K; kk += 64) a = fmaf(x[kk], W[(size_t)n * K + kk], a);
        a = wave_sum(a);
        if (lane == 0) {
            a += b[n];
            y[n] = relu ? fmaxf(a, 0.f) : a;
        }
    }
}

__global__ __launch_bounds__(256) void output_heads_kernel(const float* __restrict__ queries, k::HeadWeights hw,
                                                           float* __restrict__ hyper, float* __restrict__ iou) {
    __shared__ float x0[DIM], x1[DIM], x2[DIM];
    const int p = blockIdx.x, mi = blockIdx.y;
    const int tok = mi < 4 ? 1 + mi : 0;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    x0[threadIdx.x] = queries[((size_t)p * TOK + tok) * DIM + threadIdx.x];
    __syncthreads();
    mlp_layer(x0, hw.w[mi][0], hw.b[mi][0], x1, DIM, DIM, true, wave, lane);
    __syncthreads();
    mlp_layer(x1, hw.w[mi][1], hw.b[mi][1], x2, DIM, DIM, true, wave, lane);
    __syncthreads();
    const int nout = mi < 4 ? 32 : 4;
    float* dst = mi < 4 ? hyper + ((size_t)p * 4 + mi) * 32 : iou + (size_t)p * 4;
    mlp_layer(x2, hw.w[mi][2], hw.b[mi][2], dst, DIM, nout, false, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// logits[p][m][Y][X] = hyper[p][m] . up[pixel]; `up` rows are in quad order (see kernels.hpp).
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ up, const float* __restrict__ hyper,
                                                          float* __restrict__ logits) {
    __shared__ float sh[4 * 32];
    const size_t gidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int p = (int)(gidx >> 16);
    if (threadIdx.x < 128) sh[threadIdx.x] = hyper[(size_t)p * 128 + threadIdx.x];
    __syncthreads();
    const int q = (int)(gidx & 65535);
    const int s2 = q & 3, s1 = (q >> 2) & 3, tok = q >> 4;
    const int y = tok >> 6, x = tok & 63;
    const int Y = 4 * y + 2 * (s1 >> 1) + (s2 >> 1);
    const int X = 4 * x + 2 * (s1 & 1) + (s2 & 1);
    const float4_t* src = reinterpret_cast<const float4_t*>(up + gidx * 32);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
        const float4_t u = src[c4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m] = fmaf(u[e], sh[m * 32 + c4 * 4 + e], acc[m]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) logits[(((size_t)p * 4 + m) * 256 + Y) * 256 + X] = acc[m];
}

}  // namespace

namespace k {

void prompt_tokens(const float* coords, const float* labels, const float* gauss, const float* point_embed,
                   const float* not_a_point, const float* iou_token, const float* mask_tokens, float* tokens, int P,
                   hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(prompt_tokens_kernel, dim3(P), dim3(256), 0, s, coords, labels, gauss, point_embed, not_a_point,
                       iou_token, mask_tokens, tokens);
}

void token_linear(const float* X, const float* X2, const float* W, const float* b, const float* R, float* Y, int rows,
                  int K, int N, int relu, hipStream_t s) {
    if (rows <= 0 || N <= 0) return;
    if (K <= 0) throw_error("token_linear: K must be positive");
    hipLaunchKernelGGL(token_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, X, X2, W, b, R, Y, rows, K, N, relu);
}

void token_self_attention(const float* q, const float* kx, const float* v, float* out, int P, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(token_self_attention_kernel, dim3(P), dim3(256), 0, s, q, kx, v, out);
}

void token_to_image_attention(const float* q, const half_t* K, int ldk, const half_t* V, int ldv, float* out, int P,
                              hipStream_t s) {
    if (P <= 0) return;
    if (ldk % 8 || ldv % 8 || (((uintptr_t)K | (uintptr_t)V) & 15))
        throw_error("token_to_image_attention: K/V rows must be 16-byte aligned");
    hipLaunchKernelGGL(token_to_image_kernel, dim3(P * HEADS), dim3(256), 0, s, q, K, ldk, V, ldv, out);
}

void image_to_token_attention(const half_t* q, int ldq, const float* kt, const float* vt, half_t* out, int P,
                              hipStream_t s) {
    if (P <= 0) return;
    if (ldq % 8 || (((uintptr_t)q | (uintptr_t)out) & 15))
        throw_error("image_to_token_attention: q rows must be 16-byte aligned");
    hipLaunchKernelGGL(image_to_token_kernel, dim3(P * NTOK_IMG * HEADS / 256), dim3(256), 0, s, q, ldq, kt, vt, out);
}

void output_heads(const float* queries, const HeadWeights& hw, float* hyper, float* iou, int P, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(output_heads_kernel, dim3(P, 5), dim3(256), 0, s, queries, hw, hyper, iou);
}

void mask_logits(const float* up, const float* hyper, float* logits, int P, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(mask_logits_kernel, dim3(P * 65536 / 256), dim3(256), 0, s, up, hyper, logits);
}

}  // namespace k
}  // namespace dlimg

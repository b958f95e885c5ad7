// Token-side kernels of the SAM prompt encoder / mask decoder (everything that is NOT a big GEMM).
// In the reference all of this runs inside the decoder ONNX graph behind Session::operator()
// (/root/reference/src/segmentation.cpp:154-158); the published definition is SAM's PromptEncoder,
// TwoWayTransformer and MaskDecoder.  The token side is 7 tokens per prompt: latency-bound fp32
// VALU work, kept in fp32 end to end.
#include "device_common.hpp"
#include "kernels.hpp"

#include <mutex>

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
DLIMG_DEVICE void prompt_tokens_block(int p, const float* __restrict__ coords, const float* __restrict__ labels,
                                      const float* __restrict__ gauss, const float* __restrict__ point_embed,
                                      const float* __restrict__ not_a_point, const float* __restrict__ iou_token,
                                      const float* __restrict__ mask_tokens, float* __restrict__ tokens,
                                      float* __restrict__ tokens_copy) {
    const int c = threadIdx.x;
    float* t = tokens + (size_t)p * TOK * DIM;
    float* t2 = tokens_copy + (size_t)p * TOK * DIM;      // the decoder's running queries start as a copy
    t[c] = t2[c] = iou_token[c];
#pragma unroll
    for (int m = 0; m < 4; ++m) t[(1 + m) * DIM + c] = t2[(1 + m) * DIM + c] = mask_tokens[m * DIM + c];
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
        t[(5 + i) * DIM + c] = t2[(5 + i) * DIM + c] = e;
    }
}

constexpr int RCHUNK = 8;      // rows a wave accumulates at a time in the token linears

// ---------------------------------------------------------------------------------------------
// Tokens attend to the 4096 image positions (8 heads x 16), in two steps so that the 4096 keys of a head are spread
// over 8 workgroups (one workgroup per head streams 0.5 MB through a single CU and takes 44 us):
//   partial: workgroup = (prompt, head, key group of 512); a thread takes 2 keys (requested up front), then query
//            by query scores them, does the softmax against the wave's maximum (one exponential per score, no
//            rescale) and its part of P.V; butterfly inside each wave -> per-wave (max, sum, output[16]) in `part`
//   merge  : token_merge_out_kernel folds the 8 group partials in a fixed order and applies the output projection
constexpr int T2I_GROUPS = 8;                                  // key groups per head
constexpr int T2I_THREADS = 256;
constexpr int T2I_KEYS = NTOK_IMG / T2I_GROUPS / T2I_THREADS;  // keys per thread
constexpr int T2I_WAVES = T2I_THREADS / 64;
constexpr int T2I_PARTS = T2I_GROUPS;                          // partial triples per (prompt, head, query)

__global__ __launch_bounds__(T2I_THREADS) void token_to_image_partial_kernel(const float* __restrict__ q,
                                                                             const half_t* __restrict__ K, int ldk,
                                                                             const half_t* __restrict__ V, int ldv,
                                                                             float* __restrict__ part) {
    __shared__ float sq[TOK * 16];
    __shared__ float wpart[TOK][T2I_WAVES][18];
    const int grp = blockIdx.x % T2I_GROUPS, h = (blockIdx.x / T2I_GROUPS) % HEADS, p = blockIdx.x / (T2I_GROUPS * HEADS);
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    if (tid < TOK * 16) sq[tid] = q[((size_t)p * TOK + tid / 16) * INNER + h * 16 + (tid & 15)] * 0.25f;   // 16^-0.5
    const size_t key0 = (size_t)p * NTOK_IMG + (size_t)grp * (NTOK_IMG / T2I_GROUPS);
    const half_t* kb = K + key0 * ldk + h * 16;
    const half_t* vb = V + key0 * ldv + h * 16;
    half8_t kreg[T2I_KEYS][2], vreg[T2I_KEYS][2];
#pragma unroll
    for (int i = 0; i < T2I_KEYS; ++i) {
        const size_t j = (size_t)i * T2I_THREADS + tid;
        kreg[i][0] = *reinterpret_cast<const half8_t*>(kb + j * ldk);
        kreg[i][1] = *reinterpret_cast<const half8_t*>(kb + j * ldk + 8);
        vreg[i][0] = *reinterpret_cast<const half8_t*>(vb + j * ldv);
        vreg[i][1] = *reinterpret_cast<const half8_t*>(vb + j * ldv + 8);
    }
    __syncthreads();
    float* dst = part + ((((size_t)p * HEADS + h) * TOK) * T2I_PARTS + grp) * 18;
#pragma unroll 1
    for (int t = 0; t < TOK; ++t) {
        // keep K / V as the f16 they arrived in: otherwise the conversions to float are hoisted out of the query loop
#pragma unroll
        for (int i = 0; i < T2I_KEYS; ++i)
            asm volatile("" : "+v"(kreg[i][0]), "+v"(kreg[i][1]), "+v"(vreg[i][0]), "+v"(vreg[i][1]));
        float sc[T2I_KEYS];
#pragma unroll
        for (int i = 0; i < T2I_KEYS; ++i) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s = fmaf(sq[t * 16 + e], (float)kreg[i][e >> 3][e & 7], s);
            sc[i] = s;
        }
        float m = sc[0];
#pragma unroll
        for (int i = 1; i < T2I_KEYS; ++i) m = fmaxf(m, sc[i]);
        const float M = wave_max(m);
        float l = 0.f, o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int i = 0; i < T2I_KEYS; ++i) {
            const float pj = __expf(sc[i] - M);
            l += pj;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = fmaf(pj, (float)vreg[i][e >> 3][e & 7], o[e]);
        }
        const float ls = wave_sum(l);
        float* d = wpart[t][wave];
        if (lane == 0) { d[0] = M; d[1] = ls; }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = wave_sum(o[e]);
            if (lane == 0) d[2 + e] = x;
        }
    }
    // the four waves' partials of every query are folded here, in wave order: one triple per workgroup leaves
    __syncthreads();
    if (tid < TOK * 18) {
        const int t = tid / 18, e = tid % 18;
        float M = wpart[t][0][0];
#pragma unroll
        for (int w = 1; w < T2I_WAVES; ++w) M = fmaxf(M, wpart[t][w][0]);
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < T2I_WAVES; ++w) acc += (e == 0 ? 0.f : wpart[t][w][e]) * __expf(wpart[t][w][0] - M);
        dst[(size_t)t * T2I_PARTS * 18 + e] = e == 0 ? M : acc;
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
// One wave per output neuron at a time: the 64 lanes read one 1-KB weight row with a single 16-byte load each
// (a thread per neuron reads 64 rows per instruction, one line each), multiply with the activations in LDS and fold
// with the DPP wave sum; 16 waves share the 256 neurons of a layer.
constexpr int HEAD_THREADS = 1024;
static_assert(DIM == 64 * 4, "one float4 of the weight row per lane");

template <int N_OUT>
DLIMG_DEVICE void head_layer(const float* x /*LDS*/, const float* __restrict__ w, const float* __restrict__ b, int n_out,
                             bool relu, float* y) {
    constexpr int WAVES = HEAD_THREADS / 64, ROWS = (N_OUT + WAVES - 1) / WAVES;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const float4_t v = reinterpret_cast<const float4_t*>(x)[lane];
    float4_t wr[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {             // every row of this wave is requested before the first is used
        const int n = wave + r * WAVES;
        wr[r] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (n < n_out) wr[r] = reinterpret_cast<const float4_t*>(w + (size_t)n * DIM)[lane];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int n = wave + r * WAVES;
        const float acc = wave_sum((v[0] * wr[r][0] + v[1] * wr[r][1]) + (v[2] * wr[r][2] + v[3] * wr[r][3]));
        if (lane == 0 && n < n_out) {
            const float out = acc + b[n];
            y[n] = relu ? fmaxf(out, 0.f) : out;
        }
    }
}

__global__ __launch_bounds__(HEAD_THREADS) void output_heads_kernel(k::TokenRows queries, k::HeadWeights hw,
                                                                    float* __restrict__ hyper, float* __restrict__ iou) {
    __shared__ __attribute__((aligned(16))) float x0[DIM], x1[DIM], x2[DIM];
    __shared__ float2_t stat[1];
    const int p = blockIdx.x, mi = blockIdx.y;
    const int tok = mi < 4 ? 1 + mi : 0;
    const size_t row = (size_t)p * TOK + tok;
    // the token after norm_final_attn: statistics of its row by wave 0, normalised while it is staged
    if (queries.ln_w && threadIdx.x < 64) {
        const float4_t v = reinterpret_cast<const float4_t*>(queries.x + row * DIM)[threadIdx.x];
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / DIM);
        const float4_t d = v - mean;
        const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / DIM);
        if (threadIdx.x == 0) stat[0] = float2_t{mean, 1.0f / sqrtf(var + queries.eps)};
    }
    __syncthreads();
    if (threadIdx.x < DIM) {
        float v = queries.x[row * DIM + threadIdx.x];
        if (queries.ln_w) v = (v - stat[0][0]) * stat[0][1] * queries.ln_w[threadIdx.x] + queries.ln_b[threadIdx.x];
        x0[threadIdx.x] = v;
    }
    __syncthreads();
    head_layer<DIM>(x0, hw.w[mi][0], hw.b[mi][0], DIM, true, x1);
    __syncthreads();
    head_layer<DIM>(x1, hw.w[mi][1], hw.b[mi][1], DIM, true, x2);
    __syncthreads();
    float* dst = mi < 4 ? hyper + ((size_t)p * 4 + mi) * 32 : iou + (size_t)p * 4;
    head_layer<32>(x2, hw.w[mi][2], hw.b[mi][2], mi < 4 ? 32 : 4, false, dst);
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


// ---------------------------------------------------------------------------------------------
// Token-side linear layers, several per launch, with the LayerNorm in front of them applied on the fly.
//
// The token side is a chain of tiny dependent steps (7 tokens per prompt); each launch costs 5-9 us whatever it computes
// (dispatch + a cold weight row per wave), so what matters is how few launches the chain takes.  Every LayerNorm of the
// two-way blocks is therefore evaluated by its CONSUMERS: a linear layer (or a residual read) names the un-normalised
// rows and the LayerNorm's parameters, every workgroup computes the rows' statistics itself (two-pass, fp32, as
// layernorm_vec_kernel) and normalises while it reads.  Layers that consume the same step run in one launch (q / k / v).
// (A lane-per-row form with the rows staged in LDS and wave-uniform weight loads was built as well: 18.8 us per launch
// at one prompt against 9.4 us for this one, 27 against 35 us at five prompts -- scalar weight loads and LDS reads
// share one counter and serialise.  Not kept.)
constexpr int TL_MAX_ROWS = 112;              // 16 prompts x 7 tokens per launch
constexpr int TL_MAX_OPS = 5;
constexpr int TL_PROMPT_SLICE = 2;           // prompts per workgroup of the fused attention-output kernels
constexpr int TL_ROW_SLICE = 14;            // rows per workgroup of token_linears_kernel (two prompts)

struct LinJob { k::TokenLinear op[TL_MAX_OPS]; int count; int rows; };

// (mean, rstd) of every row of a TokenRows matrix with a LayerNorm: rows dealt to the four waves
DLIMG_DEVICE void token_row_stats(const k::TokenRows& m, int rows, float2_t* stat /*LDS [rows]*/, int row0 = 0) {
    const int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = row0 + wave; r < rows; r += nw) {
        const float4_t v = reinterpret_cast<const float4_t*>(m.x + (size_t)r * DIM)[lane];
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / DIM);
        const float4_t d = v - mean;
        const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / DIM);
        if (lane == 0) stat[r] = float2_t{mean, 1.0f / sqrtf(var + m.eps)};
    }
}
// 4 consecutive columns (4 * c4 ..) of row r
DLIMG_DEVICE float4_t token_row_load4(const k::TokenRows& m, const float2_t* stat, int r, int c4) {
    float4_t v = reinterpret_cast<const float4_t*>(m.x + (size_t)r * DIM)[c4];
    if (m.ln_w) {
        const float2_t st = stat[r];
        v = (v - st[0]) * st[1] * reinterpret_cast<const float4_t*>(m.ln_w)[c4] + reinterpret_cast<const float4_t*>(m.ln_b)[c4];
    }
    if (m.add) v += reinterpret_cast<const float4_t*>(m.add + (size_t)r * DIM)[c4];
    return v;
}
DLIMG_DEVICE float token_row_load1(const k::TokenRows& m, const float2_t* stat, int r, int c) {
    float v = m.x[(size_t)r * DIM + c];
    if (m.ln_w) v = (v - stat[r][0]) * stat[r][1] * m.ln_w[c] + m.ln_b[c];
    if (m.add) v += m.add[(size_t)r * DIM + c];
    return v;
}

// One output column per wave: Y[r][n] = act(in[r] . W[n] + b[n]) + resid[r][n] for all rows.
// lds_in: rows of the input already in LDS ([rows][K] fp32; K <= 256) or null (then `op.in` is read from memory)
// w_first: the wave's first 64 float4 of its weight row, requested by the caller before its own prologue (the row is
// cold in the caches: its latency then runs behind the statistics / attention work instead of after it)
DLIMG_DEVICE float4_t token_weight_prefetch(const k::TokenLinear& op, int first_col) {
    const int lane = lane_id();
    const int n = first_col + (threadIdx.x >> 6);
    if (n >= op.N || lane >= (op.K >> 2)) return float4_t{0.f, 0.f, 0.f, 0.f};
    return reinterpret_cast<const float4_t*>(op.W + (size_t)n * op.K)[lane];
}
DLIMG_DEVICE void token_linear_columns(const k::TokenLinear& op, int rows, int first_col, const float* lds_in,
                                       const float2_t* stat_in, const float2_t* stat_res, float4_t w_first, int row0 = 0) {
    const int lane = lane_id();
    const int n = first_col + (threadIdx.x >> 6);
    if (n >= op.N) return;
    const int K4 = op.K >> 2;
    const float4_t* wr = reinterpret_cast<const float4_t*>(op.W + (size_t)n * op.K);
    for (int r0 = row0; r0 < rows; r0 += RCHUNK) {
        float acc[RCHUNK];
#pragma unroll
        for (int r = 0; r < RCHUNK; ++r) acc[r] = 0.f;
        for (int k4 = lane; k4 < K4; k4 += 64) {
            const float4_t w = k4 == lane ? w_first : wr[k4];
#pragma unroll
            for (int r = 0; r < RCHUNK; ++r) {
                if (r0 + r < rows) {
                    float4_t x;
                    if (lds_in) x = reinterpret_cast<const float4_t*>(lds_in + (size_t)(r0 + r - row0) * op.K)[k4];
                    else if (op.K == DIM) x = token_row_load4(op.in, stat_in, r0 + r, k4);
                    else x = reinterpret_cast<const float4_t*>(op.in.x + (size_t)(r0 + r) * op.K)[k4];
                    acc[r] = fmaf(x[0], w[0], fmaf(x[1], w[1], fmaf(x[2], w[2], fmaf(x[3], w[3], acc[r]))));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RCHUNK; ++r) {
            float v = wave_sum(acc[r]);
            if (lane == 0 && r0 + r < rows) {
                v += op.b ? op.b[n] : 0.f;
                if (op.relu) v = fmaxf(v, 0.f);
                if (op.resid.x) v += token_row_load1(op.resid, stat_res, r0 + r, n);
                op.Y[(size_t)(r0 + r) * op.N + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void token_linears_kernel(LinJob job) {
    __shared__ float2_t stat_in[TL_MAX_ROWS], stat_res[TL_MAX_ROWS];
    int o = 0, first = blockIdx.x * 4;
    while (o + 1 < job.count && first >= job.op[o].N) { first -= job.op[o].N; ++o; }      // N is a multiple of 4
    const k::TokenLinear& op = job.op[o];
    const float4_t w_first = token_weight_prefetch(op, first);
    // rows are dealt to blockIdx.y in slices of TL_ROW_SLICE: more prompts are more workgroups, not longer ones
    const int row0 = blockIdx.y * TL_ROW_SLICE, row1 = min(job.rows, row0 + TL_ROW_SLICE);
    const bool ln_in = op.in.ln_w && op.K == DIM, ln_res = op.resid.x && op.resid.ln_w;
    if (ln_in) token_row_stats(op.in, row1, stat_in, row0);
    if (ln_res) token_row_stats(op.resid, row1, stat_res, row0);
    if (ln_in || ln_res) __syncthreads();
    token_linear_columns(op, row1, first, nullptr, stat_in, stat_res, w_first, row0);
}

// Deep layers (the token MLP's second linear, K = 2048): one column per wave like token_linears_kernel, but the input
// rows are staged in LDS by the whole workgroup and the wave's weight row is requested in full before that, so no
// load sits inside the accumulation loop (that loop, 8 dependent trips to L2, made this launch 20.5 us for one prompt).
// Same order of additions as token_linear_columns: results are bit-identical.
constexpr int TLD_MAX_K = 2048;
__global__ __launch_bounds__(256) void token_linear_deep_kernel(k::TokenLinear op, int rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [row slice][K]
    __shared__ float2_t stat_res[TL_MAX_ROWS];
    const int lane = lane_id();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int K4 = op.K >> 2, trips = K4 >> 6;
    const int row0 = blockIdx.y * TL_ROW_SLICE, row1 = min(rows, row0 + TL_ROW_SLICE);
    float4_t w[TLD_MAX_K / 256];
#pragma unroll
    for (int t = 0; t < TLD_MAX_K / 256; ++t)
        w[t] = (t < trips && n < op.N) ? reinterpret_cast<const float4_t*>(op.W + (size_t)n * op.K)[lane + 64 * t]
                                       : float4_t{0.f, 0.f, 0.f, 0.f};
    const float4_t* src = reinterpret_cast<const float4_t*>(op.in.x + (size_t)row0 * op.K);
    float4_t* dst = reinterpret_cast<float4_t*>(lds);
    for (int i = threadIdx.x; i < (row1 - row0) * K4; i += 256) dst[i] = src[i];
    if (op.resid.x && op.resid.ln_w) token_row_stats(op.resid, row1, stat_res, row0);
    __syncthreads();
    if (n >= op.N) return;
    for (int r = row0; r < row1; ++r) {
        const float4_t* xr = dst + (size_t)(r - row0) * K4;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < TLD_MAX_K / 256; ++t)
            if (t < trips) {
                const float4_t x = xr[lane + 64 * t];
                acc = fmaf(x[0], w[t][0], fmaf(x[1], w[t][1], fmaf(x[2], w[t][2], fmaf(x[3], w[t][3], acc))));
            }
        float v = wave_sum(acc);
        if (lane == 0) {
            v += op.b ? op.b[n] : 0.f;
            if (op.relu) v = fmaxf(v, 0.f);
            if (op.resid.x) v += token_row_load1(op.resid, stat_res, r, n);
            op.Y[(size_t)r * op.N + n] = v;
        }
    }
}

// Self-attention among the 7 tokens of every prompt (8 heads x 32), recomputed by every workgroup into LDS, followed by
// the output projection (one column per wave) with bias and residual: one launch instead of two.
// Thread c owns channel c of q, k and v of a prompt (21 registers); a head is 32 adjacent lanes, so a score is one product
// per lane summed over the half wave (DPP inside the rows of 16, one lane exchange across them).  [The first version had
// every lane walk all 32 channels of its head out of LDS: 3136 LDS reads per thread and prompt, 12-14 us per launch.]
DLIMG_DEVICE float sum_over_32_lanes(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v + __shfl_xor(v, 16, 64);
}
__global__ __launch_bounds__(256) void token_self_attn_out_kernel(const float* __restrict__ q, const float* __restrict__ kx,
                                                                  const float* __restrict__ v, k::TokenLinear op, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // prompts are dealt to blockIdx.y in pairs (rows row0 .. row1 of the token matrix; LDS rows are local)
    const int p0 = blockIdx.y * TL_PROMPT_SLICE, p1 = min(P, p0 + TL_PROMPT_SLICE);
    const int row0 = p0 * TOK, row1 = p1 * TOK;
    float* att = lds;                                   // [rows of this slice][256]
    float2_t* stat_res = reinterpret_cast<float2_t*>(att + (size_t)TL_PROMPT_SLICE * TOK * DIM) - row0;     // indexed by global row
    const float4_t w_first = token_weight_prefetch(op, blockIdx.x * 4);
    const int c = threadIdx.x;
    const float scale = 0.17677669529663687f;           // 32^-0.5
    for (int p = p0; p < p1; ++p) {
        float rq[TOK], rk[TOK], rv[TOK];
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
            rq[t] = q[((size_t)p * TOK + t) * DIM + c] * scale;
            rk[t] = kx[((size_t)p * TOK + t) * DIM + c];
            rv[t] = v[((size_t)p * TOK + t) * DIM + c];
        }
#pragma unroll
        for (int t = 0; t < TOK; ++t) {
            float s[TOK];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < TOK; ++j) {
                s[j] = sum_over_32_lanes(rq[t] * rk[j]);
                m = fmaxf(m, s[j]);
            }
            float l = 0.f, o = 0.f;
#pragma unroll
            for (int j = 0; j < TOK; ++j) {
                const float pj = expf(s[j] - m);
                l += pj;
                o = fmaf(pj, rv[j], o);
            }
            att[((size_t)(p - p0) * TOK + t) * DIM + c] = o / l;
        }
    }
    if (op.resid.x && op.resid.ln_w) token_row_stats(op.resid, row1, stat_res, row0);
    __syncthreads();
    token_linear_columns(op, row1, blockIdx.x * 4, att, nullptr, stat_res, w_first, row0);
}

// The per-key-group partials of the token-to-image attention folded (fixed order) by every workgroup into LDS, followed
// by the output projection (K = 128) with bias and residual.
__global__ __launch_bounds__(256) void token_merge_out_kernel(const float* __restrict__ part, k::TokenLinear op, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* att = lds;                                   // [P * 7][128]
    float2_t* stat_res = reinterpret_cast<float2_t*>(att + (size_t)P * TOK * INNER);
    const float4_t w_first = token_weight_prefetch(op, blockIdx.x * 4);
    const int p0 = blockIdx.y * TL_PROMPT_SLICE, p1 = min(P, p0 + TL_PROMPT_SLICE);
    const int row0 = p0 * TOK, row1 = p1 * TOK;
    const int total = (p1 - p0) * TOK * INNER;
    for (int idx = threadIdx.x; idx < total; idx += 256) {       // ((p * TOK + t) * HEADS + h) * 16 + e, p local
        const int e = idx & 15, h = (idx >> 4) % HEADS, t = (idx / (16 * HEADS)) % TOK, p = p0 + idx / (16 * HEADS * TOK);
        const float* src = part + ((((size_t)p * HEADS + h) * TOK + t) * T2I_PARTS) * 18;
        float M = src[0];
#pragma unroll
        for (int w = 1; w < T2I_PARTS; ++w) M = fmaxf(M, src[w * 18]);
        float ls = 0.f, os = 0.f;
#pragma unroll
        for (int w = 0; w < T2I_PARTS; ++w) {
            const float f = __expf(src[w * 18] - M);
            ls += src[w * 18 + 1] * f;
            os += src[w * 18 + 2 + e] * f;
        }
        att[idx] = os / ls;
    }
    if (op.resid.x && op.resid.ln_w) token_row_stats(op.resid, row1, stat_res, row0);
    __syncthreads();
    token_linear_columns(op, row1, blockIdx.x * 4, att, nullptr, stat_res, w_first, row0);
}

// ---------------------------------------------------------------------------------------------
// Image side, start of a decode: keys = embedding + no_mask_embed (has_mask_input == 0, segmentation.cpp:43-45) as fp32
// and f16 (the A operand of the image-side projections), for all prompts.
// One launch starts a decode: workgroups 0 .. P-1 build the prompts' tokens (prompt_tokens_block), the others the keys.
// The two have nothing to do with each other except that both are the first step of their chain -- and every launch of
// the decoder costs its 5-9 us of dependent latency.
struct DecoderStart {
    const float* coords; const float* labels; const float* gauss; const float* point_embed; const float* not_a_point;
    const float* iou_token; const float* mask_tokens; float* tokens; float* tokens_copy;
    const float* const* emb; const float* no_mask; float* keys; half_t* keys_h; size_t n4_per_prompt; int P;
};
__global__ __launch_bounds__(256) void decoder_start_kernel(DecoderStart a) {
    if ((int)blockIdx.x < a.P) {
        prompt_tokens_block(blockIdx.x, a.coords, a.labels, a.gauss, a.point_embed, a.not_a_point, a.iou_token, a.mask_tokens,
                            a.tokens, a.tokens_copy);
        return;
    }
    const float* const* __restrict__ emb = a.emb;
    const float* __restrict__ no_mask = a.no_mask;
    float* __restrict__ keys = a.keys;
    half_t* __restrict__ keys_h = a.keys_h;
    const size_t n4_per_prompt = a.n4_per_prompt;
    const int P = a.P;
    const size_t total = n4_per_prompt * P;
    const size_t nblk = gridDim.x - P;
    for (size_t i = (blockIdx.x - P) * (size_t)blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
        const size_t p = i / n4_per_prompt, j = i % n4_per_prompt;
        float4_t v = reinterpret_cast<const float4_t*>(emb[p])[j];
        v += reinterpret_cast<const float4_t*>(no_mask)[j % (DIM / 4)];
        reinterpret_cast<float4_t*>(keys)[i] = v;
        reinterpret_cast<half4_t*>(keys_h)[i] = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    }
}

// LayerNorm of the keys (norm4 of a two-way block) in place, with the f16 form the image-side GEMMs consume.
__global__ __launch_bounds__(256) void decoder_keys_norm_kernel(float* __restrict__ keys, const float* __restrict__ w,
                                                                const float* __restrict__ b, float eps,
                                                                half_t* __restrict__ keys_h, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = lane_id();
    float4_t v = reinterpret_cast<const float4_t*>(keys + (size_t)row * DIM)[lane];
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)DIM;
    v -= mean;
    const float rstd = 1.0f / sqrtf(wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) / (float)DIM + eps);
    const float4_t y = v * rstd * reinterpret_cast<const float4_t*>(w)[lane] + reinterpret_cast<const float4_t*>(b)[lane];
    reinterpret_cast<float4_t*>(keys + (size_t)row * DIM)[lane] = y;
    reinterpret_cast<half4_t*>(keys_h + (size_t)row * DIM)[lane] = half4_t{(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
}

}  // namespace

namespace k {

void decoder_start(const float* coords, const float* labels, const float* gauss, const float* point_embed,
                   const float* not_a_point, const float* iou_token, const float* mask_tokens, float* tokens,
                   float* tokens_copy, const float* const* emb_dev, const float* no_mask, float* keys, half_t* keys_h, int P,
                   hipStream_t s) {
    if (P <= 0) return;
    const size_t n4 = (size_t)NTOK_IMG * DIM / 4;
    const size_t total = n4 * P;
    const int key_blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    DecoderStart a{coords, labels, gauss, point_embed, not_a_point, iou_token, mask_tokens, tokens, tokens_copy,
                   emb_dev, no_mask, keys, keys_h, n4, P};
    hipLaunchKernelGGL(decoder_start_kernel, dim3(P + key_blocks), dim3(256), 0, s, a);
}

size_t token_to_image_scratch_floats(int P) { return (size_t)P * HEADS * TOK * T2I_PARTS * 18; }

void image_to_token_attention(const half_t* q, int ldq, const float* kt, const float* vt, half_t* out, int P,
                              hipStream_t s) {
    if (P <= 0) return;
    if (ldq % 8 || (((uintptr_t)q | (uintptr_t)out) & 15))
        throw_error("image_to_token_attention: q rows must be 16-byte aligned");
    hipLaunchKernelGGL(image_to_token_kernel, dim3(P * NTOK_IMG * HEADS / 256), dim3(256), 0, s, q, ldq, kt, vt, out);
}

void token_linears(const TokenLinear* ops, int count, int rows, hipStream_t s) {
    if (count <= 0 || rows <= 0) return;
    if (count > TL_MAX_OPS || rows > TL_MAX_ROWS) throw_error("token_linears: too many layers or rows for one launch");
    LinJob job;
    job.count = count;
    job.rows = rows;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (ops[i].K <= 0 || ops[i].K % 4 || ops[i].N <= 0 || ops[i].N % 4)
            throw_error("token_linears: K and N must be positive multiples of 4");
        if ((ops[i].in.ln_w || ops[i].in.add) && ops[i].K != DIM)
            throw_error("token_linears: on-the-fly LayerNorm / addition needs 256-wide input rows");
        job.op[i] = ops[i];
        blocks += ops[i].N / 4;
    }
    const int slices = (rows + TL_ROW_SLICE - 1) / TL_ROW_SLICE;
    if (count == 1 && ops[0].K > DIM && ops[0].K % 256 == 0 && ops[0].K <= TLD_MAX_K && !ops[0].in.ln_w && !ops[0].in.add) {
        static k::LdsOptIn opt_in;
        opt_in.ensure((const void*)token_linear_deep_kernel, (size_t)TL_ROW_SLICE * TLD_MAX_K * 4,
                      "token_linears: the device refuses the kernel's LDS size");
        const size_t lds = (size_t)std::min(rows, TL_ROW_SLICE) * ops[0].K * 4;
        hipLaunchKernelGGL(token_linear_deep_kernel, dim3(ops[0].N / 4, slices), dim3(256), lds, s, ops[0], rows);
        return;
    }
    hipLaunchKernelGGL(token_linears_kernel, dim3(blocks, slices), dim3(256), 0, s, job);
}

void token_self_attention_out(const float* q, const float* kx, const float* v, const TokenLinear& out, int P, hipStream_t s) {
    if (P <= 0) return;
    if (P * TOK > TL_MAX_ROWS || out.K != DIM || out.N % 4) throw_error("token_self_attention_out: unsupported shape");
    const size_t lds = (size_t)TL_PROMPT_SLICE * TOK * (DIM * 4 + 8);
    static k::LdsOptIn opt_in;
    opt_in.ensure((const void*)token_self_attn_out_kernel, 160 * 1024, "token_self_attention_out: the device refuses the kernel's LDS size");
    hipLaunchKernelGGL(token_self_attn_out_kernel, dim3(out.N / 4, (P + TL_PROMPT_SLICE - 1) / TL_PROMPT_SLICE), dim3(256), lds, s, q, kx,
                       v, out, P);
}

void token_merge_out(const float* scratch, const TokenLinear& out, int P, hipStream_t s) {
    if (P <= 0) return;
    if (P * TOK > TL_MAX_ROWS || out.K != INNER || out.N % 4) throw_error("token_merge_out: unsupported shape");
    const size_t lds = (size_t)P * TOK * INNER * 4 + (size_t)P * TOK * 8;
    static k::LdsOptIn opt_in;
    opt_in.ensure((const void*)token_merge_out_kernel, 160 * 1024, "token_merge_out: the device refuses the kernel's LDS size");
    hipLaunchKernelGGL(token_merge_out_kernel, dim3(out.N / 4, (P + TL_PROMPT_SLICE - 1) / TL_PROMPT_SLICE), dim3(256), lds, s, scratch, out,
                       P);
}

void token_to_image_partials(const float* q, const half_t* K, int ldk, const half_t* V, int ldv, float* scratch, int P,
                             hipStream_t s) {
    if (P <= 0) return;
    if (ldk % 8 || ldv % 8 || (((uintptr_t)K | (uintptr_t)V) & 15))
        throw_error("token_to_image_attention: K/V rows must be 16-byte aligned");
    hipLaunchKernelGGL(token_to_image_partial_kernel, dim3(P * HEADS * T2I_GROUPS), dim3(T2I_THREADS), 0, s, q, K, ldk, V,
                       ldv, scratch);
}

void decoder_keys_norm(float* keys, const float* w, const float* b, float eps, half_t* keys_h, int P, hipStream_t s) {
    if (P <= 0) return;
    const int rows = P * NTOK_IMG;
    hipLaunchKernelGGL(decoder_keys_norm_kernel, dim3(rows / 4), dim3(256), 0, s, keys, w, b, eps, keys_h, rows);
}

void output_heads(const TokenRows& queries, const HeadWeights& hw, float* hyper, float* iou, int P, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(output_heads_kernel, dim3(P, 5), dim3(HEAD_THREADS), 0, s, queries, hw, hyper, iou);
}

void mask_logits(const float* up, const float* hyper, float* logits, int P, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(mask_logits_kernel, dim3(P * 65536 / 256), dim3(256), 0, s, up, hyper, logits);
}

}  // namespace k
}  // namespace dlimg

// Token-side kernels of the SAM prompt encoder / mask decoder (everything that is NOT a big GEMM).
// In the reference all of this runs inside the decoder ONNX graph behind Session::operator()
// (/root/reference/src/segmentation.cpp:154-158); the published definition is SAM's PromptEncoder,
// TwoWayTransformer and MaskDecoder.  The token side is 7 tokens per prompt: latency-bound fp32
// VALU work, kept in fp32 end to end.
#include "device_common.hpp"
#include "kernels.hpp"

#include <mutex>
#include <type_traits>

namespace dlimg {
namespace {

#include "gemm_f16_tile.inc"

constexpr int TOK = 7;        // iou token + 4 mask tokens + 2 prompt tokens
constexpr int DIM = 256;
constexpr int INNER = 128;    // cross-attention width (downsample 2)
constexpr int HEADS = 8;
constexpr int NTOK_IMG = 4096;

// ---------------------------------------------------------------------------------------------
// Prompt encoder: SamOnnxModel._embed_points applied to the two packed points
// (segmentation.cpp:135-152 packs them; labels 1/-1 for a point, 2/3 for a box).
// Column c of the 7 token rows of prompt p: iou token, 4 mask tokens, the 2 prompt points.
DLIMG_DEVICE void prompt_token_column(const k::DecoderPrompts& pr, int p, int c, const float* __restrict__ gauss,
                                      const float* __restrict__ point_embed, const float* __restrict__ not_a_point,
                                      const float* __restrict__ iou_token, const float* __restrict__ mask_tokens,
                                      float (&out)[7]) {
    out[0] = iou_token[c];
#pragma unroll
    for (int m = 0; m < 4; ++m) out[1 + m] = mask_tokens[m * DIM + c];
    const int kf = c & 127;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float x = (pr.coords[(p * 2 + i) * 2 + 0] + 0.5f) / 1024.0f;
        const float y = (pr.coords[(p * 2 + i) * 2 + 1] + 0.5f) / 1024.0f;
        float v = __fadd_rn(__fmul_rn(2.0f * x - 1.0f, gauss[kf]), __fmul_rn(2.0f * y - 1.0f, gauss[128 + kf]));
        v = 6.283185307179586f * v;
        float e = c < 128 ? sinf(v) : cosf(v);
        const float lab = pr.labels[p * 2 + i];
        if (lab == -1.0f) e = not_a_point[c];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
            if (lab == (float)k4) e += point_embed[k4 * DIM + c];
        out[5 + i] = e;
    }
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
constexpr int RCHUNK = 8;               // rows a wave accumulates at a time in the token linears
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

// Rows [row0, row0 + rows) of a token matrix as its consumers see them (LayerNorm, addend) into LDS [rows][256].  What a
// launch of the token side costs is its chain of dependent round trips to L2 (1-2 us each), so every load of the step is
// issued before the first wait.  blockDim.x == 256: thread = column; statistics a wave per row (token_row_stats' arithmetic).
template <int MAXR>
DLIMG_DEVICE void stage_token_rows(const k::TokenRows& m, int row0, int rows, float* x_lds, float2_t* stat_lds) {
    const int c = threadIdx.x, lane = lane_id(), wave = c >> 6;
    float xv[MAXR], av[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        xv[r] = r < rows ? m.x[(size_t)(row0 + r) * DIM + c] : 0.f;
        av[r] = (m.add && r < rows) ? m.add[(size_t)(row0 + r) * DIM + c] : 0.f;
    }
    const float lw = m.ln_w ? m.ln_w[c] : 1.f, lb = m.ln_w ? m.ln_b[c] : 0.f;
    if (m.ln_w) {
        constexpr int PER_WAVE = (MAXR + 3) / 4;
        float4_t sv[PER_WAVE];
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int r = wave + 4 * i;
            sv[i] = r < rows ? reinterpret_cast<const float4_t*>(m.x + (size_t)(row0 + r) * DIM)[lane] : float4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int r = wave + 4 * i;
            const float4_t v = sv[i];
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / DIM);
            const float4_t d = v - mean;
            const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / DIM);
            if (lane == 0 && r < rows) stat_lds[r] = float2_t{mean, 1.0f / sqrtf(var + m.eps)};
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
        if (r < rows) {
            float v = xv[r];
            if (m.ln_w) v = (v - stat_lds[r][0]) * stat_lds[r][1] * lw + lb;
            x_lds[r * DIM + c] = v + av[r];
        }
    __syncthreads();
}

// One output column per wave: Y[r][n] = act(in[r] . W[n] + b[n]) + resid[r][n] for all rows of a slice (<= 16).
// lds_in: rows of the input already in LDS ([rows][K] fp32; K <= 256) or null (then `op.in` is read from memory).
// TokenColumn: what the wave needs from memory for its column, requested by the caller BEFORE its own prologue so that the
// latency runs behind the statistics / attention work instead of after it: the first 64 float4 of the weight row (cold in
// the caches), the bias, and -- lane L for row row0 + L, which is the lane that finishes and stores that row -- the
// residual entry with its LayerNorm scale / shift.  [Before r03 lane 0 fetched bias and residual after the sums: one more
// round trip to L2 per call, which a workgroup that runs several columns per wave pays several times over.]
struct TokenColumn { float4_t w; float bias, rx, radd, rlw, rlb; };
static_assert(TL_ROW_SLICE <= 16, "lane L of a wave finishes row L of the slice");
DLIMG_DEVICE TokenColumn token_column_prefetch(const k::TokenLinear& op, int first_col, int row0, int row1) {
    const int lane = lane_id();
    const int n = first_col + (threadIdx.x >> 6);
    TokenColumn c{float4_t{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, 0.f, 1.f, 0.f};
    if (n >= op.N) return c;
    if (lane < (op.K >> 2)) c.w = reinterpret_cast<const float4_t*>(op.W + (size_t)n * op.K)[lane];
    if (op.b) c.bias = op.b[n];
    if (op.resid.x) {
        const int r = min(row0 + lane, row1 - 1);
        c.rx = op.resid.x[(size_t)r * DIM + n];
        if (op.resid.add) c.radd = op.resid.add[(size_t)r * DIM + n];
        if (op.resid.ln_w) { c.rlw = op.resid.ln_w[n]; c.rlb = op.resid.ln_b[n]; }
    }
    return c;
}
// NR = rows of the slice, 7 or 14 (whole prompts).
// The accumulate loop keeps the round-2 shape -- a run-time row bound, one guarded step per row, chunks of 8 -- and the whole
// library is built with -fno-slp-vectorize (dlimgedit_amd/build.py).  With the row count a compile-time constant and the
// seven rows' loads in straight-line code, the SLP vectoriser paired the rows (2,1), (4,3), (6,5) into chains of
// v_pk_fma_f32 with op_sel between v_mov shuffles, and THAT code got ONE output element in about 10^4 decodes wrong (error
// 0.02-0.5, always the low half of a pair: token row 2, 4 or 6) as soon as four host threads kept the execution lanes busy,
// and never from one thread.  Found by tools/decoder_stress.py (every workspace of the token side compared bit for bit with
// the serial answer); bisected in a second build by swapping single pieces: the epilogue, system-scope loads of the rows,
// a second barrier, lane-0 stores and lower occupancy made no difference, the loop shape did (0 in 320 000 decodes), and so
// did the compiler pass alone: the straight-line source built with -fno-slp-vectorize is clean too (0 in 240 000, the
// vectorised build of the same source 7 in 60 000 on the same box).  Every wait the ISA manual asks for is in the failing
// code; whether it is a gap in the compiler's hazard table for packed fp32 on gfx950 or an erratum is not known.  The
// tuning build keeps the failing form (-DDLIMG_STRAIGHT_ROWS, build it WITHOUT -fno-slp-vectorize to see it fail);
// tests/test_gpu_concurrency.py keeps the stress in the suite.
template <int NR>
DLIMG_DEVICE void token_linear_columns(const k::TokenLinear& op, int first_col, const float* lds_in, const float2_t* stat_in,
                                       const float2_t* stat_res, const TokenColumn& col, int row0) {
    static_assert(NR % TOK == 0 && NR <= 16, "slices are whole prompts");
    const int lane = lane_id();
    const int n = first_col + (threadIdx.x >> 6);
    if (n >= op.N) return;
    const int K4 = op.K >> 2;
    const float4_t* wr = reinterpret_cast<const float4_t*>(op.W + (size_t)n * op.K);
    float mine = 0.f;                       // the finished sum of row row0 + lane
#if defined(DLIMG_TUNING) && defined(DLIMG_STRAIGHT_ROWS)
    // the failing form, kept in the tuning build for the hunt (tools/decoder_stress.py with DLIMGEDIT_TUNING_LIB)
#pragma unroll
    for (int c0 = 0; c0 < NR; c0 += TOK) {
        float acc[TOK];
#pragma unroll
        for (int r = 0; r < TOK; ++r) acc[r] = 0.f;
        for (int k4 = lane; k4 < K4; k4 += 64) {
            const float4_t w = k4 == lane ? col.w : wr[k4];
            float4_t x[TOK];
#pragma unroll
            for (int r = 0; r < TOK; ++r) {
                if (lds_in) x[r] = reinterpret_cast<const float4_t*>(lds_in + (size_t)(c0 + r) * op.K)[k4];
                else if (op.K == DIM) x[r] = token_row_load4(op.in, stat_in, row0 + c0 + r, k4);
                else x[r] = reinterpret_cast<const float4_t*>(op.in.x + (size_t)(row0 + c0 + r) * op.K)[k4];
            }
#pragma unroll
            for (int r = 0; r < TOK; ++r)
                acc[r] = fmaf(x[r][0], w[0], fmaf(x[r][1], w[1], fmaf(x[r][2], w[2], fmaf(x[r][3], w[3], acc[r]))));
        }
#pragma unroll
        for (int r = 0; r < TOK; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == c0 + r) mine = v;
        }
    }
#else
    const int rows = row0 + NR;
    for (int r0 = row0; r0 < rows; r0 += RCHUNK) {
        float acc[RCHUNK];
#pragma unroll
        for (int r = 0; r < RCHUNK; ++r) acc[r] = 0.f;
        for (int k4 = lane; k4 < K4; k4 += 64) {
            const float4_t w = k4 == lane ? col.w : wr[k4];
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
            const float v = wave_sum(acc[r]);
            if (lane == r0 - row0 + r) mine = v;
        }
    }
#endif
    if (lane < NR) {
        const int r = row0 + lane;
        float v = mine + col.bias;
        if (op.relu) v = fmaxf(v, 0.f);
        if (op.resid.x) {
            float x = col.rx;
            if (op.resid.ln_w) x = (x - stat_res[r][0]) * stat_res[r][1] * col.rlw + col.rlb;
            v += x + col.radd;
        }
        op.Y[(size_t)r * op.N + n] = v;
    }
}
// the body once for each slice size
#define DLIMG_FOR_SLICE_ROWS(count, ...)                                   \
    if ((count) == TL_ROW_SLICE) { constexpr int NR = TL_ROW_SLICE; __VA_ARGS__ } \
    else { constexpr int NR = TOK; __VA_ARGS__ }

#if defined(DLIMG_TUNING) && defined(DLIMG_LOW_OCCUPANCY)
__attribute__((amdgpu_waves_per_eu(1, 4)))
#endif
__global__ __launch_bounds__(256) void token_linears_kernel(LinJob job) {
    __shared__ float2_t stat_in[TL_MAX_ROWS], stat_res[TL_MAX_ROWS];
    int o = 0, first = blockIdx.x * 4;
    while (o + 1 < job.count && first >= job.op[o].N) { first -= job.op[o].N; ++o; }      // N is a multiple of 4
    const k::TokenLinear& op = job.op[o];
    // rows are dealt to blockIdx.y in slices of TL_ROW_SLICE: more prompts are more workgroups, not longer ones
    const int row0 = blockIdx.y * TL_ROW_SLICE, row1 = min(job.rows, row0 + TL_ROW_SLICE);
    const TokenColumn w_first = token_column_prefetch(op, first, row0, row1);
    const bool ln_in = op.in.ln_w && op.K == DIM, ln_res = op.resid.x && op.resid.ln_w;
    if (ln_in) token_row_stats(op.in, row1, stat_in, row0);
    if (ln_res) token_row_stats(op.resid, row1, stat_res, row0);
    if (ln_in || ln_res) __syncthreads();
    DLIMG_FOR_SLICE_ROWS(row1 - row0, token_linear_columns<NR>(op, first, nullptr, stat_in, stat_res, w_first, row0);)
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
    k::TokenLinear tail = op;          // bias and residual of the wave's column, requested with the weights (TokenColumn)
    tail.K = 0;
    const TokenColumn col = token_column_prefetch(tail, blockIdx.x * 4, row0, row1);
    const float4_t* src = reinterpret_cast<const float4_t*>(op.in.x + (size_t)row0 * op.K);
    float4_t* dst = reinterpret_cast<float4_t*>(lds);
    for (int i = threadIdx.x; i < (row1 - row0) * K4; i += 256) dst[i] = src[i];
    if (op.resid.x && op.resid.ln_w) token_row_stats(op.resid, row1, stat_res, row0);
    __syncthreads();
    if (n >= op.N) return;
    float mine = 0.f;
    for (int r = row0; r < row1; ++r) {
        const float4_t* xr = dst + (size_t)(r - row0) * K4;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < TLD_MAX_K / 256; ++t)
            if (t < trips) {
                const float4_t x = xr[lane + 64 * t];
                acc = fmaf(x[0], w[t][0], fmaf(x[1], w[t][1], fmaf(x[2], w[t][2], fmaf(x[3], w[t][3], acc))));
            }
        const float v = wave_sum(acc);
        if (lane == r - row0) mine = v;
    }
    const int r = row0 + lane;
    if (r < row1) {
        float v = mine + col.bias;
        if (op.relu) v = fmaxf(v, 0.f);
        if (op.resid.x) {
            float x = col.rx;
            if (op.resid.ln_w) x = (x - stat_res[r][0]) * stat_res[r][1] * col.rlw + col.rlb;
            v += x + col.radd;
        }
        op.Y[(size_t)r * op.N + n] = v;
    }
}

// Tokens attend to the 4096 image positions (8 heads x 16), in two steps so that the 4096 keys of a head are spread
// over 8 workgroups (one workgroup per head streams 0.5 MB through a single CU and takes 44 us):
//   partial: workgroup = (prompt, head, key group of 512); a thread takes 2 keys (requested up front), then query
//            by query scores them, does the softmax against the wave's maximum (one exponential per score, no
//            rescale) and its part of P.V; butterfly inside each wave -> per-wave (max, sum, output[16]) in `part`
//   merge  : the consumers (token_merge_linear_kernel, output_heads_kernel) fold the 8 group partials in a fixed order
//            and apply the output projection themselves
// The query projection (LayerNorm + positional part on the fly, 256 -> 128) can ride in this launch: a workgroup needs
// the 16 columns of its head for 7 tokens only and computes them with the code (and bits) of token_linears_kernel.
constexpr int T2I_GROUPS = 8;                                  // key groups per head
constexpr int T2I_THREADS = 256;
constexpr int T2I_KEYS = NTOK_IMG / T2I_GROUPS / T2I_THREADS;  // keys per thread
constexpr int T2I_WAVES = T2I_THREADS / 64;
constexpr int T2I_PARTS = T2I_GROUPS;                          // partial triples per (prompt, head, query)

__global__ __launch_bounds__(T2I_THREADS) void token_to_image_partial_kernel(const float* __restrict__ q, k::TokenLinear qp,
                                                                             const half_t* __restrict__ K, int ldk,
                                                                             const half_t* __restrict__ V, int ldv,
                                                                             float* __restrict__ part) {
    __shared__ float sq[TOK * 16];
    __shared__ float qrows[TOK * INNER];
    __shared__ __attribute__((aligned(16))) float xrows[TOK * DIM];
    __shared__ float2_t qstat[TOK];
    __shared__ float wpart[TOK][T2I_WAVES][18];
    const int grp = blockIdx.x % T2I_GROUPS, h = (blockIdx.x / T2I_GROUPS) % HEADS, p = blockIdx.x / (T2I_GROUPS * HEADS);
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
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
    if (qp.W) {
        // q = qp(rows of prompt p) for the 16 columns of head h, 4 columns (one per wave) at a time
        const int row0 = p * TOK;
        TokenColumn wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = token_column_prefetch(qp, h * 16 + 4 * i, row0, row0 + TOK);
        stage_token_rows<TOK>(qp.in, row0, TOK, xrows, qstat);
        qp.Y = qrows - (size_t)row0 * INNER;
#pragma unroll
        for (int i = 0; i < 4; ++i) token_linear_columns<TOK>(qp, h * 16 + 4 * i, xrows, nullptr, nullptr, wf[i], row0);
        __syncthreads();
        if (tid < TOK * 16) sq[tid] = qrows[(tid / 16) * INNER + h * 16 + (tid & 15)] * 0.25f;               // 16^-0.5
    } else if (tid < TOK * 16) {
        sq[tid] = q[((size_t)p * TOK + tid / 16) * INNER + h * 16 + (tid & 15)] * 0.25f;
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
DLIMG_DEVICE void token_self_attn_out_body(const float* __restrict__ q, const float* __restrict__ kx, const float* __restrict__ v,
                                           const k::TokenLinear& op, int P, const int block_x, const int block_y, float* lds) {
    // prompts are dealt to block_y in pairs (rows row0 .. row1 of the token matrix; LDS rows are local)
    const int p0 = block_y * TL_PROMPT_SLICE, p1 = min(P, p0 + TL_PROMPT_SLICE);
    const int row0 = p0 * TOK, row1 = p1 * TOK;
    float* att = lds;                                   // [rows of this slice][256]
    float2_t* stat_res = reinterpret_cast<float2_t*>(att + (size_t)TL_PROMPT_SLICE * TOK * DIM) - row0;     // indexed by global row
    const TokenColumn w_first = token_column_prefetch(op, block_x * 4, row0, row1);
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
    DLIMG_FOR_SLICE_ROWS(row1 - row0, token_linear_columns<NR>(op, block_x * 4, att, nullptr, stat_res, w_first, row0);)
}
__global__ __launch_bounds__(256) void token_self_attn_out_kernel(const float* __restrict__ q, const float* __restrict__ kx,
                                                                  const float* __restrict__ v, k::TokenLinear op, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    token_self_attn_out_body(q, kx, v, op, P, blockIdx.x, blockIdx.y, lds);
}

// The same launch with the layer's image-side projection riding along: [K | Q | V] = keys . W^T (decoder_image side, an MFMA
// GEMM of 64 x 64 tiles, gemm_f16_tile) does not depend on the token self-attention and the self-attention not on it, and
// on their own they are 14 and 19 us one after the other, twice per decode (the stream is in order; a second stream costs
// more in cross-queue waits than it hides, and hipExtAnyOrderLaunch is not honoured on gfx950: LABNOTES r06).  The last
// `gemm_tiles` workgroups take one GEMM tile each -- the same code, tile shape and K order as the launch of its own, so the
// same bits -- in front of them the self-attention's workgroups.  Both parts use 256 threads.
constexpr int SAG_BM = 64, SAG_BN = 64;
constexpr size_t SAG_GEMM_LDS = (size_t)2 * (SAG_BM + SAG_BN) * 64 * 2 + aux_bytes(SAG_BM, SAG_BN);
constexpr size_t SAG_TOKEN_LDS = (size_t)TL_PROMPT_SLICE * TOK * (DIM * 4 + 8);
constexpr size_t SAG_LDS = SAG_GEMM_LDS > SAG_TOKEN_LDS ? SAG_GEMM_LDS : SAG_TOKEN_LDS;
static_assert(SAG_LDS <= 48 * 1024, "below the default dynamic-LDS limit: no opt-in needed");
__global__ __launch_bounds__(256, 4) void self_attn_out_and_gemm_kernel(k::GemmArgs g, int gemm_tiles, const float* __restrict__ q,
                                                                        const float* __restrict__ kx, const float* __restrict__ v,
                                                                        k::TokenLinear op, int P, int token_blocks_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the self-attention's workgroups come first: with many prompts the GEMM has thousands of tiles, and workgroups are
    // dispatched in index order -- behind them the long chain of the token part would start when the GEMM is all but done
    const int token_blocks = (int)gridDim.x - gemm_tiles;        // a multiple of 8: the tiles keep their XCDs
    if ((int)blockIdx.x < token_blocks) {
        token_self_attn_out_body(q, kx, v, op, P, blockIdx.x % token_blocks_x, blockIdx.x / token_blocks_x, reinterpret_cast<float*>(smem));
        return;
    }
    gemm_f16_tile<SAG_BM, SAG_BN, 2, 2, 64, 2, 4, k::ACT_NONE, EPI_PLAIN>(g, xcd_remap((int)blockIdx.x - token_blocks, gemm_tiles), smem);
}

// The rest of the token-to-image attention, done by its CONSUMERS (a launch of its own cost 9-16 us for 0.2 MFLOP):
//   merge_partials         folds the 8 key-group partials of (prompt, head, token) in a fixed order -> att [rows][128] in LDS
//   output projection      thread c = output column c: y[r][c] = att[r] . Wo[c] + b[c] + resid[r][c] with Wo TRANSPOSED
//                          ([128][256], prepared at load) so that a wave reads 256 contiguous bytes per k
// Every workgroup of the consumer repeats both for the rows it needs (7 x 256 x 128 FMAs at most per prompt, the 128 KB of
// Wo come out of L2), so the result is the same bits wherever it is computed and nothing has to be exchanged.
// NR rows starting at token t0 of prompt p0 (wrapping into the next prompts); blockDim.x >= 128.  Thread = (head, element)
// for the rows r = (tid / 128) + i * (blockDim.x / 128); the 24 values of up to 4 rows are requested before the first is
// used (one round trip to L2 for 4 rows, not four).
template <int NR>
DLIMG_DEVICE void merge_partials(const float* __restrict__ part, int p0, int t0, float* att /*LDS [NR][128]*/) {
    const int e = threadIdx.x & 15, h = (threadIdx.x >> 4) & (HEADS - 1);
    const int rstep = blockDim.x >> 7, rfirst = threadIdx.x >> 7;
    constexpr int GROUP = 4;
    for (int base = 0; base < NR; base += GROUP * rstep) {
        float M[GROUP][T2I_PARTS], L[GROUP][T2I_PARTS], O[GROUP][T2I_PARTS];
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            const int r = min(base + rfirst + i * rstep, NR - 1);
            const int p = p0 + (t0 + r) / TOK, t = (t0 + r) % TOK;
            const float* src = part + ((((size_t)p * HEADS + h) * TOK + t) * T2I_PARTS) * 18;
#pragma unroll
            for (int w = 0; w < T2I_PARTS; ++w) {
                M[i][w] = src[w * 18];
                L[i][w] = src[w * 18 + 1];
                O[i][w] = src[w * 18 + 2 + e];
            }
        }
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            const int r = base + rfirst + i * rstep;
            float mx = M[i][0];
#pragma unroll
            for (int w = 1; w < T2I_PARTS; ++w) mx = fmaxf(mx, M[i][w]);
            float ls = 0.f, os = 0.f;
#pragma unroll
            for (int w = 0; w < T2I_PARTS; ++w) {
                const float f = __expf(M[i][w] - mx);
                ls += L[i][w] * f;
                os += O[i][w] * f;
            }
            if (r < NR) att[r * INNER + h * 16 + e] = os / ls;
        }
    }
}
// one row (the heads kernel): threads 0..127
DLIMG_DEVICE void merge_partials_one(const float* __restrict__ part, int p, int t, float* att /*LDS [128]*/) {
    const int e = threadIdx.x & 15, h = (threadIdx.x >> 4) & (HEADS - 1);
    const float* src = part + ((((size_t)p * HEADS + h) * TOK + t) * T2I_PARTS) * 18;
    float M[T2I_PARTS], L[T2I_PARTS], O[T2I_PARTS];
#pragma unroll
    for (int w = 0; w < T2I_PARTS; ++w) {
        M[w] = src[w * 18];
        L[w] = src[w * 18 + 1];
        O[w] = src[w * 18 + 2 + e];
    }
    float mx = M[0];
#pragma unroll
    for (int w = 1; w < T2I_PARTS; ++w) mx = fmaxf(mx, M[w]);
    float ls = 0.f, os = 0.f;
#pragma unroll
    for (int w = 0; w < T2I_PARTS; ++w) {
        const float f = __expf(M[w] - mx);
        ls += L[w] * f;
        os += O[w] * f;
    }
    att[h * 16 + e] = os / ls;
}
// (mean, rstd) of rows held in LDS ([rows][256]); the arithmetic of token_row_stats
DLIMG_DEVICE void lds_row_stats(const float* y, int rows, float eps, float2_t* stat) {
    const int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < rows; r += nw) {
        const float4_t v = reinterpret_cast<const float4_t*>(y + (size_t)r * DIM)[lane];
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / DIM);
        const float4_t d = v - mean;
        const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / DIM);
        if (lane == 0) stat[r] = float2_t{mean, 1.0f / sqrtf(var + eps)};
    }
}

// Token-to-image attention finished (merge + output projection + residual -> out.Y, written by the first column block) and
// the NEXT linear (`next`, K = 256, whose input is the LayerNorm of those rows) applied to it: a workgroup = COLS columns
// of `next` for the rows of one prompt.  COLS = 16, 32 or 64: the smallest workgroups that still fit the chip in one round (16 for one or two prompts, 64
// from five) (every workgroup pays the fold, the 128 KB of Wo and the projection before its columns: at five prompts 640
// workgroups of 16 columns took 53 us, two and a half rounds on 256 CUs).  The arithmetic of a column does not depend on COLS.  [Kept small on purpose: code that runs once is fetched cold, and a 43 KB body --
// 32 columns, two prompts, everything unrolled -- took 24 us where this one takes half.]
constexpr size_t TML_LDS = (size_t)(INNER * DIM + TL_ROW_SLICE * INNER + TL_ROW_SLICE * DIM) * 4 + 2 * TL_ROW_SLICE * 8;
template <int NR, int COLS>
DLIMG_DEVICE void token_merge_linear_body(const float* __restrict__ part, const k::TokenLinear& out,
                                          const float* __restrict__ out_wt, const k::TokenLinear& next, int p0, float* lds) {
    float* wt = lds;                                     // [128][256]: the whole transposed output projection (128 KB)
    float* att = wt + INNER * DIM;                       // [rows][128]
    float* y = att + TL_ROW_SLICE * INNER;               // [rows][256]
    float2_t* stat_res = reinterpret_cast<float2_t*>(y + TL_ROW_SLICE * DIM);
    float2_t* stat_in = stat_res + TL_ROW_SLICE;
    const int row0 = p0 * TOK, row1 = row0 + NR;
    const int first = blockIdx.x * COLS;
    const int c = threadIdx.x, lane = lane_id(), wave = c >> 6;
    // everything this workgroup reads is requested here: Wo as DMA into LDS (no registers), the wave's columns of `next`,
    // the residual rows; the partials follow in merge_partials
    for (int k = wave; k < INNER; k += 4) glds16(out_wt + (size_t)k * DIM + lane * 4, wt + k * DIM);
    TokenColumn col = token_column_prefetch(next, first, row0, row1);      // the later columns are requested one step ahead
    float res[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) res[r] = out.resid.x ? out.resid.x[(size_t)(row0 + r) * DIM + c] : 0.f;
    const bool res_ln = out.resid.x && out.resid.ln_w;
    const float rw = res_ln ? out.resid.ln_w[c] : 1.f, rb = res_ln ? out.resid.ln_b[c] : 0.f;
    const float ob = out.b ? out.b[c] : 0.f;
    const float nw = next.in.ln_w ? next.in.ln_w[c] : 1.f, nb = next.in.ln_w ? next.in.ln_b[c] : 0.f;
    float nadd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) nadd[r] = next.in.add ? next.in.add[(size_t)(row0 + r) * DIM + c] : 0.f;
    merge_partials<NR>(part, p0, 0, att);
    if (res_ln) token_row_stats(out.resid, row1, stat_res - row0, row0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // output projection: thread = column, k in order
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
#pragma unroll 2
    for (int k4 = 0; k4 < INNER / 4; ++k4) {
        float w[4];
        float4_t a[NR];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = wt[(4 * k4 + i) * DIM + c];
#pragma unroll
        for (int r = 0; r < NR; ++r) a[r] = reinterpret_cast<const float4_t*>(att)[r * (INNER / 4) + k4];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[r] = fmaf(a[r][i], w[i], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float v = acc[r] + ob;
        if (out.resid.x) v += res_ln ? (res[r] - stat_res[r][0]) * stat_res[r][1] * rw + rb : res[r];
        y[r * DIM + c] = v;
        if (blockIdx.x == 0) out.Y[(size_t)(row0 + r) * DIM + c] = v;
    }
    __syncthreads();
    if (next.in.ln_w) {
        lds_row_stats(y, NR, next.in.eps, stat_in);
        __syncthreads();
    }
    if (next.in.ln_w || next.in.add) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float v = y[r * DIM + c];
            if (next.in.ln_w) v = (v - stat_in[r][0]) * stat_in[r][1] * nw + nb;
            y[r * DIM + c] = v + nadd[r];
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int i = 0; i < COLS / 4; ++i) {
        const TokenColumn following = token_column_prefetch(next, first + 4 * min(i + 1, COLS / 4 - 1), row0, row1);
        token_linear_columns<NR>(next, first + 4 * i, y, nullptr, nullptr, col, row0);
        col = following;
    }
}
template <int COLS>
__global__ __launch_bounds__(256) void token_merge_linear_kernel(const float* __restrict__ part, k::TokenLinear out,
                                                                 const float* __restrict__ out_wt, k::TokenLinear next, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    token_merge_linear_body<TOK, COLS>(part, out, out_wt, next, blockIdx.y, lds);
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

// The workgroup first finishes the final token-to-image attention for ITS token (merge of the partials, output projection,
// residual: see merge_partials) and applies norm_final_attn to it.
__global__ __launch_bounds__(HEAD_THREADS) void output_heads_kernel(const float* __restrict__ part, k::TokenLinear out,
                                                                    const float* __restrict__ out_wt, k::TokenRows norm,
                                                                    k::HeadWeights hw, float* __restrict__ hyper,
                                                                    float* __restrict__ iou) {
    __shared__ __attribute__((aligned(16))) float x0[DIM], x1[DIM], x2[DIM], att[INNER];
    __shared__ float opart[4][DIM];
    __shared__ float2_t stat[1], stat_res[1];
    const int p = blockIdx.x, mi = blockIdx.y;
    const int tok = mi < 4 ? 1 + mi : 0;
    const int row = p * TOK + tok;
    // one round trip: this thread's 32 weights of the output projection (column c, k quarter kq), the partials, the residual
    const int c = threadIdx.x & (DIM - 1), kq = threadIdx.x >> 8;
    float w[INNER / 4];
#pragma unroll
    for (int k = 0; k < INNER / 4; ++k) w[k] = out_wt[(size_t)(kq * (INNER / 4) + k) * DIM + c];
    float res = 0.f, rw = 1.f, rb = 0.f;
    if (out.resid.x && kq == 0) {
        res = out.resid.x[(size_t)row * DIM + c];
        if (out.resid.ln_w) { rw = out.resid.ln_w[c]; rb = out.resid.ln_b[c]; }
    }
    const float ob = (out.b && kq == 0) ? out.b[c] : 0.f;
    if (threadIdx.x < INNER) merge_partials_one(part, p, tok, att);
    if (out.resid.x && out.resid.ln_w && (threadIdx.x >> 6) == 15) {       // statistics of the residual row: the last wave
        const int lane = lane_id();
        const float4_t v = reinterpret_cast<const float4_t*>(out.resid.x + (size_t)row * DIM)[lane];
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / DIM);
        const float4_t d = v - mean;
        const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / DIM);
        if (lane == 0) stat_res[0] = float2_t{mean, 1.0f / sqrtf(var + out.resid.eps)};
    }
    __syncthreads();
    {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < INNER / 4; ++k) acc = fmaf(att[kq * (INNER / 4) + k], w[k], acc);
        opart[kq][c] = acc;
    }
    __syncthreads();
    if (kq == 0) {
        float v = ((opart[0][c] + opart[1][c]) + (opart[2][c] + opart[3][c])) + ob;
        if (out.resid.x) v += out.resid.ln_w ? (res - stat_res[0][0]) * stat_res[0][1] * rw + rb : res;
        x1[c] = v;
    }
    __syncthreads();
    // the token after norm_final_attn: statistics of its row by wave 0, normalised while it is staged
    if (threadIdx.x < 64) lds_row_stats(x1, 1, norm.eps, stat);
    __syncthreads();
    if (threadIdx.x < DIM)
        x0[threadIdx.x] = (x1[threadIdx.x] - stat[0][0]) * stat[0][1] * norm.ln_w[threadIdx.x] + norm.ln_b[threadIdx.x];
    __syncthreads();
    head_layer<DIM>(x0, hw.w[mi][0], hw.b[mi][0], DIM, true, x1);
    __syncthreads();
    head_layer<DIM>(x1, hw.w[mi][1], hw.b[mi][1], DIM, true, x2);
    __syncthreads();
    float* dst = mi < 4 ? hyper + ((size_t)p * 4 + mi) * 32 : iou + (size_t)p * 4;
    head_layer<32>(x2, hw.w[mi][2], hw.b[mi][2], mi < 4 ? 32 : 4, false, dst);
}

// ---------------------------------------------------------------------------------------------
// Image side, start of a decode: keys = embedding + no_mask_embed (has_mask_input == 0, segmentation.cpp:43-45) as fp32
// and f16 (the A operand of the image-side projections), for all prompts.
// One launch starts a decode: workgroups 0 .. P-1 write the prompts' tokens (the positional part every later step adds),
// the next `lin_blocks` apply the first linears of the token side (q / k / v of the first self-attention) to those same
// rows, which they rebuild in LDS instead of waiting for them, and the rest initialise the keys.  The three have nothing
// to do with each other except that all are the first step of their chain -- and every launch of the decoder costs its
// 5-9 us of dependent latency.  The prompts travel as kernel arguments: no host-to-device copy in front of a decode.
struct DecoderStart {
    k::DecoderPrompts prompts;
    const float* gauss; const float* point_embed; const float* not_a_point; const float* iou_token; const float* mask_tokens;
    float* tokens;
    LinJob first;
    int lin_blocks, lin_cols;
    const float* no_mask; float* keys; half_t* keys_h; size_t n4_per_prompt; int P;
};
__global__ __launch_bounds__(256) void decoder_start_kernel(DecoderStart a) {
    __shared__ __attribute__((aligned(16))) float rows[TL_ROW_SLICE * DIM];
    const int c = threadIdx.x;
    if ((int)blockIdx.x < a.P) {
        float v[TOK];
        prompt_token_column(a.prompts, blockIdx.x, c, a.gauss, a.point_embed, a.not_a_point, a.iou_token, a.mask_tokens, v);
#pragma unroll
        for (int t = 0; t < TOK; ++t) a.tokens[((size_t)blockIdx.x * TOK + t) * DIM + c] = v[t];
        return;
    }
    if ((int)blockIdx.x < a.P + a.lin_blocks) {
        const int bid = blockIdx.x - a.P;
        int o = 0, first = (bid % a.lin_cols) * 4;
        while (o + 1 < a.first.count && first >= a.first.op[o].N) { first -= a.first.op[o].N; ++o; }
        const k::TokenLinear& op = a.first.op[o];
        const int row0 = (bid / a.lin_cols) * TL_ROW_SLICE, row1 = min(a.first.rows, row0 + TL_ROW_SLICE);
        const TokenColumn w_first = token_column_prefetch(op, first, row0, row1);
        for (int p = row0 / TOK; p * TOK < row1; ++p) {
            float v[TOK];
            prompt_token_column(a.prompts, p, c, a.gauss, a.point_embed, a.not_a_point, a.iou_token, a.mask_tokens, v);
#pragma unroll
            for (int t = 0; t < TOK; ++t) rows[(p * TOK + t - row0) * DIM + c] = v[t];
        }
        __syncthreads();
        DLIMG_FOR_SLICE_ROWS(row1 - row0, token_linear_columns<NR>(op, first, rows, nullptr, nullptr, w_first, row0);)
        return;
    }
    const float* __restrict__ no_mask = a.no_mask;
    float* __restrict__ keys = a.keys;
    half_t* __restrict__ keys_h = a.keys_h;
    const size_t n4_per_prompt = a.n4_per_prompt;
    const int P = a.P, skip = a.P + a.lin_blocks;
    const size_t total = n4_per_prompt * P;
    const size_t nblk = gridDim.x - skip;
    for (size_t i = (blockIdx.x - skip) * (size_t)blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
        const size_t p = i / n4_per_prompt, j = i % n4_per_prompt;
        float4_t v = reinterpret_cast<const float4_t*>(a.prompts.emb[p])[j];
        v += reinterpret_cast<const float4_t*>(no_mask)[j % (DIM / 4)];
        reinterpret_cast<float4_t*>(keys)[i] = v;
        reinterpret_cast<half4_t*>(keys_h)[i] = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    }
}

}  // namespace

namespace k {

void decoder_start(const DecoderPrompts& prompts, const float* gauss, const float* point_embed, const float* not_a_point,
                   const float* iou_token, const float* mask_tokens, float* tokens, const TokenLinear* first, int n_first,
                   const float* no_mask, float* keys, half_t* keys_h, int P, hipStream_t s) {
    if (P <= 0) return;
    if (P > kDecoderMaxPrompts || n_first < 0 || n_first > TL_MAX_OPS) throw_error("decoder_start: too many prompts or layers");
    const size_t n4 = (size_t)NTOK_IMG * DIM / 4;
    const size_t total = n4 * P;
    const int key_blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    DecoderStart a{};
    a.prompts = prompts;
    a.gauss = gauss; a.point_embed = point_embed; a.not_a_point = not_a_point; a.iou_token = iou_token; a.mask_tokens = mask_tokens;
    a.tokens = tokens;
    a.first.count = n_first;
    a.first.rows = P * TOK;
    int cols = 0;
    for (int i = 0; i < n_first; ++i) {
        if (first[i].K != DIM || first[i].N <= 0 || first[i].N % 4 || first[i].in.ln_w || first[i].in.add || first[i].resid.x)
            throw_error("decoder_start: the first linears take the plain 256-wide token rows");
        a.first.op[i] = first[i];
        cols += first[i].N / 4;
    }
    a.lin_cols = cols > 0 ? cols : 1;
    a.lin_blocks = cols * ((P * TOK + TL_ROW_SLICE - 1) / TL_ROW_SLICE);
    a.no_mask = no_mask; a.keys = keys; a.keys_h = keys_h; a.n4_per_prompt = n4; a.P = P;
    hipLaunchKernelGGL(decoder_start_kernel, dim3(P + a.lin_blocks + key_blocks), dim3(256), 0, s, a);
}

size_t token_to_image_scratch_floats(int P) { return (size_t)P * HEADS * TOK * T2I_PARTS * 18; }

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
    static_assert((size_t)TL_PROMPT_SLICE * TOK * (DIM * 4 + 8) <= 64 * 1024, "below the default dynamic-LDS limit: no opt-in needed");
    hipLaunchKernelGGL(token_self_attn_out_kernel, dim3(out.N / 4, (P + TL_PROMPT_SLICE - 1) / TL_PROMPT_SLICE), dim3(256), lds, s, q, kx,
                       v, out, P);
}

bool token_self_attention_out_with_gemm(const float* q, const float* kx, const float* v, const TokenLinear& out, int P,
                                        const GemmArgs& g, hipStream_t s) {
    if (P <= 0) return true;
    if (P * TOK > TL_MAX_ROWS || out.K != DIM || out.N % 4) throw_error("token_self_attention_out: unsupported shape");
    // what the 64 x 64 plain tile computes, and nothing else: otherwise the caller launches the two on their own
    if (const char* err = gemm_check(g)) throw_error(err);
    const bool plain = !g.out_l && !g.resid_h && !g.ln_stats && !g.stats_out && g.act == ACT_NONE && g.M % SAG_BM == 0 &&
                       g.N % SAG_BN == 0 && g.K % 64 == 0 && !(g.resid && g.resid_mod % SAG_BM != 0);
    if (!plain) return false;
    const int tiles = (g.M / SAG_BM) * (g.N / SAG_BN);
    const int bx = out.N / 4, by = (P + TL_PROMPT_SLICE - 1) / TL_PROMPT_SLICE;
    hipLaunchKernelGGL(self_attn_out_and_gemm_kernel, dim3(tiles + bx * by), dim3(256), SAG_LDS, s, g, tiles, q, kx, v, out, P, bx);
    return true;
}

void token_merge_linear(const float* scratch, const TokenLinear& out, const float* out_wt, const TokenLinear& next, int P,
                        hipStream_t s) {
    if (P <= 0) return;
    if (P * TOK > TL_MAX_ROWS || out.K != INNER || out.N != DIM || out.resid.add || next.K != DIM || next.N % 64 || next.resid.x)
        throw_error("token_merge_linear: unsupported shape");
    auto launch = [&](auto cols_tag, k::LdsOptIn& opt_in) {
        constexpr int COLS = decltype(cols_tag)::value;
        opt_in.ensure((const void*)token_merge_linear_kernel<COLS>, TML_LDS, "token_merge_linear: the device refuses the kernel's LDS size");
        hipLaunchKernelGGL(token_merge_linear_kernel<COLS>, dim3(next.N / COLS, P), dim3(256), TML_LDS, s, scratch, out, out_wt, next, P);
    };
    static k::LdsOptIn opt16, opt32, opt64;
    // the smallest workgroups that still fit the chip in one round (256 CUs, one workgroup each: 150 KB of LDS)
    if ((next.N / 16) * P <= 256) launch(std::integral_constant<int, 16>{}, opt16);
    else if ((next.N / 32) * P <= 256) launch(std::integral_constant<int, 32>{}, opt32);
    else launch(std::integral_constant<int, 64>{}, opt64);
}

void token_to_image_partials(const float* q, const TokenLinear* q_proj, const half_t* K, int ldk, const half_t* V, int ldv,
                             float* scratch, int P, hipStream_t s) {
    if (P <= 0) return;
    if (ldk % 8 || ldv % 8 || (((uintptr_t)K | (uintptr_t)V) & 15))
        throw_error("token_to_image_attention: K/V rows must be 16-byte aligned");
    TokenLinear qp{};
    if (q_proj) {
        if (q_proj->K != DIM || q_proj->N != INNER || q_proj->resid.x || q_proj->relu)
            throw_error("token_to_image_attention: the query projection is 256 -> 128 without residual");
        qp = *q_proj;
    } else if (!q) {
        throw_error("token_to_image_attention: neither queries nor their projection given");
    }
    hipLaunchKernelGGL(token_to_image_partial_kernel, dim3(P * HEADS * T2I_GROUPS), dim3(T2I_THREADS), 0, s, q, qp, K, ldk, V,
                       ldv, scratch);
}

void output_heads(const float* scratch, const TokenLinear& out, const float* out_wt, const TokenRows& norm,
                  const HeadWeights& hw, float* hyper, float* iou, int P, hipStream_t s) {
    if (P <= 0) return;
    if (out.K != INNER || out.N != DIM || !norm.ln_w) throw_error("output_heads: unsupported shape");
    hipLaunchKernelGGL(output_heads_kernel, dim3(P, 5), dim3(HEAD_THREADS), 0, s, scratch, out, out_wt, norm, hw, hyper, iou);
}

}  // namespace k
}  // namespace dlimg

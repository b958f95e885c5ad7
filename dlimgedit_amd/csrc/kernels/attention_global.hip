// Global self-attention of the SAM ViT encoder (4 of the blocks attend over all 64x64 tokens),
// flash-style: the 4096x4096 score matrix is never materialised.  In the reference this is part of
// the encoder ONNX graph run by Session::run (/root/reference/src/segmentation.cpp:126-128).
//
//   S[i,j] = scale * q_i.k_j + q_i.Rh[qy_i - ky_j + 63] + q_i.Rw[qx_i - kx_j + 63]   (raw q in the bias)
//
// One workgroup = 128 consecutive queries (two rows of the token grid) of one (image, head);
// 4 waves x 32 queries.  Keys are streamed in tiles of 64 = one grid row, so inside a tile ky is
// constant and kx = 0..63: the bias is a per-query scalar (from LDS) plus a per-query 64-vector that
// lives in registers in accumulator layout and is used as the MFMA's initial accumulator.
// Operands are swapped (S^T = K . Q^T) so a lane owns one query: online-softmax statistics, the
// rescale of O^T and the bias are all lane-local; P tiles feed the second MFMA straight from the
// accumulator registers (O^T = V^T . P^T).
// K/V tiles are register-staged one tile ahead (coalesced 16-byte loads issued before the MFMAs of the
// current tile, ds_write_b128 after them), two LDS buffers, one barrier per tile.  V stays row-major
// in LDS; the transposed MFMA operand comes from ds_read_b64_tr_b16 (hardware transpose read) on
// 192-byte rows, which tiles the 64 banks exactly (4 key rows x 64 B per half-wave: conflict-free).
#include "device_common.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <mutex>

namespace dlimg {
namespace {

constexpr int GRID = 64;
constexpr int TOKENS = 4096;
constexpr int KT = 64;              // keys per tile
constexpr int V_STRIDE = 96;        // elements per key row of the V tile (192 B: conflict-free ds_read_b64_tr_b16)
constexpr int RELH_STRIDE = 32;     // floats: relh_lds[wave][ky][query]
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_t lds_short4_t;
constexpr int GW_STRIDE = 33;       // floats per query row of the prologue scratch (32 rel rows + 1)

// ABL (tuning builds only, wrong results): 1 = no K/V staging after the first tiles, 2 = no softmax arithmetic,
// 3 = no P.V product
template <int HD, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attention_global_kernel(const half_t* __restrict__ qkv,
                                                                  const half_t* __restrict__ rel_h,
                                                                  const half_t* __restrict__ rel_w,
                                                                  half_t* __restrict__ out, int heads) {
    constexpr int KS = HD / 16;
    constexpr int DT = (HD + 31) / 32;
    constexpr int K_STRIDE = HD + 8;
    constexpr int CHUNKS = HD / 8;
    constexpr int STAGE_ITERS = (KT * CHUNKS + 255) / 256;
    constexpr int K_TILE = KT * K_STRIDE;               // elements
    constexpr int V_TILE = KT * V_STRIDE;               // elements
    static_assert(DT * 32 <= V_STRIDE, "head dimension tiles must fit the padded V row");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* relh_lds = reinterpret_cast<float*>(smem);                          // [4][64][32]
    half_t* lds_k = reinterpret_cast<half_t*>(smem + 4 * 64 * RELH_STRIDE * 4);   // [2][K_TILE]
    half_t* lds_v = lds_k + 2 * K_TILE;                                         // [2][V_TILE] row-major [key][V_STRIDE]
    float* scratch = reinterpret_cast<float*>(lds_k);   // prologue only: [4][32][GW_STRIDE], aliases the tile buffers

    const int D = heads * HD;
    const int ld = 3 * D;
    const int qblk = blockIdx.x % (TOKENS / 128);
    const int head = (blockIdx.x / (TOKENS / 128)) % heads;
    const int img = blockIdx.x / ((TOKENS / 128) * heads);
    const half_t* base = qkv + (size_t)img * TOKENS * ld + head * HD;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = wave_id();
    const int hi = lane >> 5, l31 = lane & 31;

    const int qtok = qblk * 128 + wave * 32 + l31;
    const int qy = qblk * 2 + (wave >> 1);              // wave-uniform
    const int qx0 = (wave & 1) * 32;

    half8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const half8_t*>(base + (size_t)qtok * ld + ks * 16 + hi * 8);

    const float inv_scale = sqrtf((float)HD);
    const float c = rsqrtf((float)HD) * 1.44269504088896341f;

    // ---- prologue: rel-pos tables via MFMA -------------------------------------------------------
    // relh[i][ky] = q_i . rel_h[qy - ky + 63]: rows rr = 0..63 <-> rel_h[qy + rr], ky = 63 - rr
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float16_t acc = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // the tables arrive as f16 (converted once at load time): one 16-byte request per fragment instead of
            // eight 4-byte ones that each touch 64 different cache lines per wave-instruction
            const half8_t rf = *reinterpret_cast<const half8_t*>(rel_h + (size_t)(qy + t * 32 + l31) * HD + ks * 16 + hi * 8);
            acc = mfma32(rf, qf[ks], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ky = 63 - (t * 32 + acc_row(r, hi));
            relh_lds[(wave * 64 + ky) * RELH_STRIDE + l31] = acc[r] * inv_scale;
        }
    }
    // relw[i][kx] = q_i . rel_w[qx_i - kx + 63], rows rr = 0..95 <-> rel_w[qx0 + rr] (rows past 126 are zero),
    // produced 32 rows at a time through a small per-wave scratch so the kernel keeps 2 workgroups per CU
    float* gw = scratch + wave * 32 * GW_STRIDE;
    float16_t relw[2];          // accumulator layout: tile jt, register r <-> kx = jt*32 + acc_row(r, hi)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float16_t acc = zero16();
        const int row = qx0 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8_t rf = zero_h8();
            if (row < 2 * GRID - 1) rf = *reinterpret_cast<const half8_t*>(rel_w + (size_t)row * HD + ks * 16 + hi * 8);
            acc = mfma32(rf, qf[ks], acc);
        }
        __syncthreads();        // previous chunk fully consumed
#pragma unroll
        for (int r = 0; r < 16; ++r) gw[l31 * GW_STRIDE + acc_row(r, hi)] = acc[r];
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = l31 - (jt * 32 + acc_row(r, hi)) + 63 - t * 32;      // row inside this chunk?
                if (rr >= 0 && rr < 32) relw[jt][r] = gw[l31 * GW_STRIDE + rr] * inv_scale;
            }
    }
    __syncthreads();            // scratch is dead; tile buffers may be written

    // ---- K/V tile staging ------------------------------------------------------------------------
    // consecutive lanes take consecutive 16-byte chunks of a key row: every global line is used whole
    // two register sets: a tile is requested two iterations before it is written to LDS, so a full
    // iteration of MFMA work covers the (loaded) L2 latency
    half8_t kregA[STAGE_ITERS], vregA[STAGE_ITERS], kregB[STAGE_ITERS], vregB[STAGE_ITERS];
    auto load_tile = [&](int t, half8_t* kreg, half8_t* vreg) {
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int idx = it * 256 + tid;
            if (idx < KT * CHUNKS) {
                const int key = idx / CHUNKS, ch = idx % CHUNKS;
                const half_t* row = base + (size_t)(t * KT + key) * ld + ch * 8;
                kreg[it] = *reinterpret_cast<const half8_t*>(row + D);
                vreg[it] = *reinterpret_cast<const half8_t*>(row + 2 * D);
            }
        }
    };
    auto write_tile = [&](int buf, const half8_t* kreg, const half8_t* vreg) {
        half_t* kd = lds_k + buf * K_TILE;
        half_t* vd = lds_v + buf * V_TILE;
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int idx = it * 256 + tid;
            if (idx < KT * CHUNKS) {
                const int key = idx / CHUNKS, ch = idx % CHUNKS;
                *reinterpret_cast<half8_t*>(kd + key * K_STRIDE + ch * 8) = kreg[it];
                *reinterpret_cast<half8_t*>(vd + key * V_STRIDE + ch * 8) = vreg[it];
            }
        }
    };

    if (DT * 32 > HD) {         // columns of V beyond the head dimension stay zero in both buffers
        for (int idx = tid; idx < 2 * V_TILE / 2; idx += 256) reinterpret_cast<uint32_t*>(lds_v)[idx] = 0u;
        __syncthreads();
    }
    // head dimension 80 has no registers to spare for the second staging set (it spilled): one set, requested at the
    // top of the iteration before the one that needs it
    constexpr bool TWO_SETS = HD <= 64;
    load_tile(0, kregA, vregA);
    write_tile(0, kregA, vregA);
    if (TWO_SETS) load_tile(1, kregB, vregB);
    __syncthreads();

    // transposed-read addressing (cdna_hip_programming.md T10): in each 16-lane group, lane 4q+p points
    // at row q, columns 4p..4p+3 of a 4-key x 16-column block and receives column (lane&15) of the 4 keys.
    // Group g = lane>>4: columns 16*(g&1).., keys of half g>>1  ==  this lane's (l31, hi) operand slot.
    const int tr_off = ((hi * 4 + ((lane & 15) >> 2)) * V_STRIDE) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    float16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
    float m = -INFINITY, l = 0.f;

    constexpr int NT = TOKENS / KT;
    static_assert(NT % 2 == 0, "the key loop is unrolled by two");
    auto process_tile = [&](int t, int buf) {

        const half_t* kb = lds_k + buf * K_TILE;
        const half_t* vb = lds_v + buf * V_TILE + tr_off;
        const float rh = relh_lds[(wave * 64 + t) * RELH_STRIDE + l31];

        // S^T tile = relw (initial accumulator, no copy: C and D of the first MFMA are different registers)
        //            + K . Q^T; the per-tile scalar rh joins in the exponent offset below
        float16_t s[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                half8_t kf = *reinterpret_cast<const half8_t*>(kb + (jt * 32 + l31) * K_STRIDE + ks * 16 + hi * 8);
                s[jt] = mfma32(kf, qf[ks], ks == 0 ? relw[jt] : s[jt]);
            }
        }

        // online softmax (per lane = per query; the two halves hold disjoint key subsets)
        float tm = s[0][0];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[jt][r]);
        tm = fmaxf(tm, swap_halves(tm)) + rh;
        const float m_new = fmaxf(m, tm);
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
        const float off = (rh - m_new) * c;
        m = m_new;
        // exponent arguments and the row sum two at a time (v_pk_fma_f32 / v_pk_add_f32); the exponentials themselves
        // are quarter-rate scalar instructions
        float2_t ps2 = {0.f, 0.f};
        const float2_t c2 = {c, c}, off2 = {off, off};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float2_t a = float2_t{s[jt][r], s[jt][r + 1]} * c2 + off2;
                float2_t p = {ABL == 2 ? s[jt][r] : __builtin_amdgcn_exp2f(a[0]),
                              ABL == 2 ? s[jt][r + 1] : __builtin_amdgcn_exp2f(a[1])};
                s[jt][r] = p[0];
                s[jt][r + 1] = p[1];
                ps2 += p;
            }
        l = l * alpha + (ps2[0] + ps2[1]);
        if (!__all(alpha == 1.0f)) {            // the running max moved for some query of this wave
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }

        // O^T += V^T . P^T   (P tile as B operand straight from the accumulator registers)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                half8_t pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (half_t)s[jt][st * 8 + e];
                const int key0 = jt * 32 + st * 16;         // element e <-> key0 + 4*hi + 8*(e>>2) + (e&3)
#pragma unroll
                for (int dt = 0; dt < (ABL == 3 ? 0 : DT); ++dt) {
                    const half_t* vp = vb + key0 * V_STRIDE + dt * 32;
                    const short4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)vp);
                    const short4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(vp + 8 * V_STRIDE));
                    const half4_t h0 = __builtin_bit_cast(half4_t, v0), h1 = __builtin_bit_cast(half4_t, v1);
                    half8_t vf = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    o[dt] = mfma32(vf, pf, o[dt]);
                }
            }
        }

    };
    for (int t = 0; !TWO_SETS && t < NT; ++t) {
        if (t + 1 < NT && ABL != 1) load_tile(t + 1, kregA, vregA);
        process_tile(t, t & 1);
        if (t + 1 < NT && ABL != 1) write_tile((t + 1) & 1, kregA, vregA);
        __syncthreads();
    }
    for (int t = 0; TWO_SETS && t < NT; t += 2) {
        if (t + 2 < NT && ABL != 1) load_tile(t + 2, kregA, vregA);
        process_tile(t, 0);
        if (ABL != 1) write_tile(1, kregB, vregB);      // tile t+1, requested one iteration ago
        __syncthreads();
        if (t + 3 < NT && ABL != 1) load_tile(t + 3, kregB, vregB);
        process_tile(t + 1, 1);
        if (t + 2 < NT && ABL != 1) write_tile(0, kregA, vregA);    // tile t+2
        __syncthreads();
    }

    // ---- epilogue: O^T[d][i] / l_i ; lane = query, registers = d ------------------------------------
    l += swap_halves(l);
    const float inv_l = 1.0f / l;
    half_t* orow = out + ((size_t)img * TOKENS + qtok) * D + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = dt * 32 + 8 * g4 + 4 * hi;
            if (d0 < HD) {
                half4_t v = {(half_t)(o[dt][g4 * 4 + 0] * inv_l), (half_t)(o[dt][g4 * 4 + 1] * inv_l),
                             (half_t)(o[dt][g4 * 4 + 2] * inv_l), (half_t)(o[dt][g4 * 4 + 3] * inv_l)};
                *reinterpret_cast<half4_t*>(orow + d0) = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Ping-pong form of the kernel above: 8 waves = 256 queries per workgroup, two groups of four waves (one wave of each
// per SIMD) that run the same program ONE BARRIER APART.  A wave alternates between an M slot -- the two MFMA batches
// S(t) = K(t).Q^T and O += V(t-1).P(t-1), their LDS reads and its share of the K/V staging -- and an X slot, the softmax
// arithmetic of tile t (pure VALU); while one group is in M the other is in X, so the matrix pipe and the vector ALU
// of every SIMD are both fed all the time instead of taking turns (counters of the 4-wave kernel: MFMA busy 22 %,
// VALU 33 %, both together 6 % of the time; two co-resident workgroups start together and stay in lockstep).
//   group A: slot 2t = M(t), slot 2t+1 = X(t);   group B: one slot later.
//   K(t) is read in slots 2t (A), 2t+1 (B); V(t) in slots 2t+2 (A), 2t+3 (B).  In its M(t) slot a wave writes its
//   piece of K(t+1) and of V(t) to LDS (two buffers each: the slots that read the overwritten tiles, K(t-1) and
//   V(t-2), ended at 2t-1) and then requests K(t+2) and V(t+1) into the same registers.
// ABL (tuning builds only, wrong results): 1 = no softmax arithmetic, 2 = no MFMAs in the M slot, 3 = neither,
// 4 = per-slot cycle counters written over the start of `out`
// VAR (bit mask; the product build uses kPPVariant, the tuning build lets DLIMGEDIT_ATTN_VAR choose):
//   1 = no s_setprio around the MFMAs of the M slot, 2 = softmax arithmetic on single values (v_fma_f32 / v_add_f32
//   through asm helpers, so the compiler does not pack them) instead of v_pk_*_f32, 4 = row sum and f16 conversion of P
//   moved from the X slot into the next M slot (implies 2), 8 = waves 4-7 run at priority 1 throughout
template <int HD, int ABL = 0, int VAR = 0>
__global__ __launch_bounds__(512, 2) void attention_global_pp_kernel(const half_t* __restrict__ qkv,
                                                                     const half_t* __restrict__ rel_h,
                                                                     const half_t* __restrict__ rel_w,
                                                                     half_t* __restrict__ out, int heads) {
    constexpr int KS = HD / 16;
    constexpr int DT = (HD + 31) / 32;
    constexpr int K_STRIDE = HD + 8;
    constexpr int CHUNKS = HD / 8;
    constexpr int K_TILE = KT * K_STRIDE;
    constexpr int V_TILE = KT * V_STRIDE;
    constexpr int PIECES = KT * CHUNKS;                  // 16-byte pieces of a K (or V) tile
    static_assert(DT * 32 <= V_STRIDE, "head dimension tiles must fit the padded V row");
    static_assert(PIECES <= 2 * 512, "at most two pieces of K and of V per thread");
    constexpr int NP = (PIECES + 511) / 512;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* relh_lds = reinterpret_cast<float*>(smem);                              // [8][64][32]
    half_t* lds_k = reinterpret_cast<half_t*>(smem + 8 * 64 * RELH_STRIDE * 4);    // [2][K_TILE]
    half_t* lds_v = lds_k + 2 * K_TILE;                                            // [2][V_TILE]
    float* scratch = reinterpret_cast<float*>(lds_k);   // prologue only: [8][32][GW_STRIDE], aliases the tile buffers
    static_assert(8 * 32 * GW_STRIDE * 4 <= (2 * K_TILE + 2 * V_TILE) * 2, "prologue scratch must fit in the tile buffers");

    const int D = heads * HD;
    const int ld = 3 * D;
    // consecutive workgroup ids go round-robin over the 8 XCDs: remapped so that the 16 query blocks of a head run on ONE
    // XCD and its L2 serves their K / V tiles (un-mapped, every XCD fetched every head: 107 MB per launch against
    // 25 MB algorithmic, profiles/r02_hbm_traffic_pmc.json)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = bid % (TOKENS / 256);
    const int head = (bid / (TOKENS / 256)) % heads;
    const int img = bid / ((TOKENS / 256) * heads);
    const half_t* base = qkv + (size_t)img * TOKENS * ld + head * HD;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = wave_id();
    const int group = wave >> 2;
    const int hi = lane >> 5, l31 = lane & 31;

    const unsigned long long c_kernel = ABL == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    const int qtok = qblk * 256 + wave * 32 + l31;
    const int qy = qblk * 4 + (wave >> 1);              // wave-uniform
    const int qx0 = (wave & 1) * 32;

    half8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const half8_t*>(base + (size_t)qtok * ld + ks * 16 + hi * 8);

    const float inv_scale = sqrtf((float)HD);
    const float c = rsqrtf((float)HD) * 1.44269504088896341f;

    // ---- prologue: rel-pos tables via MFMA (as in the 4-wave kernel) --------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float16_t acc = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // the tables arrive as f16 (converted once at load time): one 16-byte request per fragment instead of
            // eight 4-byte ones that each touch 64 different cache lines per wave-instruction
            const half8_t rf = *reinterpret_cast<const half8_t*>(rel_h + (size_t)(qy + t * 32 + l31) * HD + ks * 16 + hi * 8);
            acc = mfma32(rf, qf[ks], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ky = 63 - (t * 32 + acc_row(r, hi));
            relh_lds[(wave * 64 + ky) * RELH_STRIDE + l31] = acc[r] * inv_scale;
        }
    }
    float* gw = scratch + wave * 32 * GW_STRIDE;
    float16_t relw[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float16_t acc = zero16();
        const int row = qx0 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8_t rf = zero_h8();
            if (row < 2 * GRID - 1) rf = *reinterpret_cast<const half8_t*>(rel_w + (size_t)row * HD + ks * 16 + hi * 8);
            acc = mfma32(rf, qf[ks], acc);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) gw[l31 * GW_STRIDE + acc_row(r, hi)] = acc[r];
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = l31 - (jt * 32 + acc_row(r, hi)) + 63 - t * 32;
                if (rr >= 0 && rr < 32) relw[jt][r] = gw[l31 * GW_STRIDE + rr] * inv_scale;
            }
    }
    __syncthreads();            // scratch is dead; tile buffers may be written

    // ---- staging: thread -> pieces tid (+512) of a tile; one register set for K and one for V -----------
    half8_t kreg[NP], vreg[NP];
    auto piece_ok = [&](int it) { return NP == 1 || it * 512 + tid < PIECES; };
    auto load_k = [&](int t) {
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) {
                const int idx = it * 512 + tid, key = idx / CHUNKS, ch = idx % CHUNKS;
                kreg[it] = *reinterpret_cast<const half8_t*>(base + (size_t)(t * KT + key) * ld + ch * 8 + D);
            }
    };
    auto load_v = [&](int t) {
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) {
                const int idx = it * 512 + tid, key = idx / CHUNKS, ch = idx % CHUNKS;
                vreg[it] = *reinterpret_cast<const half8_t*>(base + (size_t)(t * KT + key) * ld + ch * 8 + 2 * D);
            }
    };
    auto write_k = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) {
                const int idx = it * 512 + tid, key = idx / CHUNKS, ch = idx % CHUNKS;
                *reinterpret_cast<half8_t*>(lds_k + buf * K_TILE + key * K_STRIDE + ch * 8) = kreg[it];
            }
    };
    auto write_v = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) {
                const int idx = it * 512 + tid, key = idx / CHUNKS, ch = idx % CHUNKS;
                *reinterpret_cast<half8_t*>(lds_v + buf * V_TILE + key * V_STRIDE + ch * 8) = vreg[it];
            }
    };
    if (DT * 32 > HD) {         // columns of V beyond the head dimension stay zero in both buffers
        for (int idx = tid; idx < 2 * V_TILE / 2; idx += 512) reinterpret_cast<uint32_t*>(lds_v)[idx] = 0u;
        __syncthreads();
    }
    constexpr int NT = TOKENS / KT;
    load_k(0);
    write_k(0);
    load_k(1);
    load_v(0);
    __syncthreads();            // K(0) is in LDS for both groups

    const int tr_off = ((hi * 4 + ((lane & 15) >> 2)) * V_STRIDE) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    float16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
    float m = -INFINITY, l = 0.f;
    float alpha_prev = 1.0f;    // VAR & 4: rescale factor of the tile whose row sum is still to be added (see finish_softmax)
    float16_t s[2];
    half8_t pf[2][2];           // P(t) as B-operand fragments: [jt][st]

    auto slot_end = [&]() {     // this wave's LDS traffic of the slot is done, then the workgroup barrier
        __builtin_amdgcn_sched_barrier(0);       // nothing (MFMAs included) moves across the slot boundary
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto scores = [&](int t) {  // S^T(t) = relw + K(t) . Q^T
        const half_t* kb = lds_k + (t & 1) * K_TILE;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                half8_t kf = *reinterpret_cast<const half8_t*>(kb + (jt * 32 + l31) * K_STRIDE + ks * 16 + hi * 8);
                s[jt] = mfma32(kf, qf[ks], ks == 0 ? relw[jt] : s[jt]);
            }
    };
    auto values = [&](int t) {  // O^T += V(t)^T . P(t)^T
        const half_t* vb = lds_v + (t & 1) * V_TILE + tr_off;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int key0 = jt * 32 + st * 16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const half_t* vp = vb + key0 * V_STRIDE + dt * 32;
                    const short4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)vp);
                    const short4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(vp + 8 * V_STRIDE));
                    const half4_t h0 = __builtin_bit_cast(half4_t, v0), h1 = __builtin_bit_cast(half4_t, v1);
                    half8_t vf = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    o[dt] = mfma32(vf, pf[jt][st], o[dt]);
                }
            }
    };
    auto softmax = [&](int t) { // X slot: online softmax of tile t, P(t) -> pf
        if (ABL & 1) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[jt][st][e] = (half_t)s[jt][st * 8 + e];
            return;
        }
        // The slot is latency-bound, not throughput-bound (1300 cycles measured for ~100 instructions): in-order issue
        // behind dependent results.  So: the maximum as a tree of 3-input maxima (depth 4 instead of a chain of 32),
        // the row sum in four independent accumulators.
        const float rh = relh_lds[(wave * 64 + t) * RELH_STRIDE + l31];
        // v_max3_f32 directly: fmaxf() makes the compiler canonicalise every MFMA result first (32 extra v_max v, x, x)
        auto max3 = [](float a, float b, float cc) {
            float r;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(cc));
            return r;
        };
        float mx[10];
#pragma unroll
        for (int g = 0; g < 10; ++g) {          // 30 of the 32 scores in threes, the last two below
            const int e0 = g * 3, e1 = e0 + 1, e2 = e0 + 2;
            mx[g] = max3(s[e0 >> 4][e0 & 15], s[e1 >> 4][e1 & 15], s[e2 >> 4][e2 & 15]);
        }
        float tm = max3(max3(mx[0], mx[1], mx[2]), max3(mx[3], mx[4], mx[5]), max3(mx[6], mx[7], mx[8]));
        if (VAR & 6) {
            // the other half's maximum without an LDS round trip (ds_bpermute sits in the slot's serial chain: every
            // exponent argument waits for it): v_permlane32_swap on two copies of tm leaves tm of lanes 0-31 in one and tm
            // of lanes 32-63 in the other, in every lane.  As ONE asm statement with its own wait states: two before (a
            // vector write of an operand, whoever made it) and one behind; the builtin form is not usable here -- hipcc
            // 7.2 folds fmax(result0, result1) of a swap of two equal operands to result0 (checked on the device).
            tm = max3(tm, mx[9], max3(s[1][14], s[1][15], s[1][15]));
            float ta = tm, tb = tm;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ta), "+v"(tb));
            tm = max3(ta, tb, tb) + rh;
        } else {
            tm = max3(tm, mx[9], max3(s[1][14], s[1][15], s[1][15]));
            tm = max3(tm, swap_halves(tm), tm) + rh;
        }
        const float m_new = fmaxf(m, tm);
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
        const float off = (rh - m_new) * c;
        m = m_new;
        if (VAR & 4) {
            // the exponentials only; row sum and f16 conversion wait for the next M slot (finish_softmax)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[jt][r], c, off));
            alpha_prev = alpha;
        } else if (VAR & 2) {
            // single-value arithmetic written as plain C (the file is built with -fno-slp-vectorize so it stays that way;
            // asm helpers are not an option here: an asm v_add that reads a fresh v_exp result misses the wait state the
            // compiler inserts between its own instructions -- wrong sums, measured)
            float ps1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[jt][r], c, off));
                    s[jt][r] = p;
                    ps1[r & 3] += p;
                }
            l = l * alpha + ((ps1[0] + ps1[1]) + (ps1[2] + ps1[3]));
        } else {
        float2_t ps[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        const float2_t c2 = {c, c}, off2 = {off, off};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float2_t a = float2_t{s[jt][r], s[jt][r + 1]} * c2 + off2;
                float2_t p = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                s[jt][r] = p[0];
                s[jt][r + 1] = p[1];
                ps[(r >> 1) & 3] += p;
            }
        const float2_t pss = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        l = l * alpha + (pss[0] + pss[1]);
        }
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        if (!(VAR & 4)) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[jt][st][e] = (half_t)s[jt][st * 8 + e];
        }
    };
    // VAR & 4: the part of the softmax that the MFMAs of the NEXT M slot do not have to wait for is done inside that M
    // slot, behind its LDS requests and before its first MFMA (which overwrites s): row sum of P(t-1), l, P(t-1) -> f16.
    // The X slot is the longer one (1280 against 900-1200 cycles measured); the M slot spends ~300 cycles waiting for
    // its first fragments, which is where this work now sits.
    auto finish_softmax = [&] {
        float ps1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ps1[r & 3] += s[jt][r];
        l = l * alpha_prev + ((ps1[0] + ps1[1]) + (ps1[2] + ps1[3]));
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[jt][st][e] = (half_t)s[jt][st * 8 + e];
    };
    // M slot of tile t (t = NT: only the last P.V product).  With registers to spare (head dimension 64) every LDS read
    // of the slot is issued before the first MFMA: read-then-wait pairs in front of each MFMA exposed the LDS latency
    // eight times per slot (1500 cycles per slot measured against 512 cycles of matrix work).
    constexpr bool PRELOAD = HD <= 64;
    auto m_slot = [&](int t) {
        if (PRELOAD) {
            half8_t kf[2][KS];
            short4_t vv[2][2][DT][2];
            if (t < NT) {
                const half_t* kb = lds_k + (t & 1) * K_TILE;
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        kf[jt][ks] = *reinterpret_cast<const half8_t*>(kb + (jt * 32 + l31) * K_STRIDE + ks * 16 + hi * 8);
            }
            if (t > 0) {
                const half_t* vb = lds_v + ((t - 1) & 1) * V_TILE + tr_off;
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const half_t* vp = vb + (jt * 32 + st * 16) * V_STRIDE + dt * 32;
                            vv[jt][st][dt][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)vp);
                            vv[jt][st][dt][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(vp + 8 * V_STRIDE));
                        }
            }
            __builtin_amdgcn_sched_barrier(0);   // all requests first
            if ((VAR & 4) && t > 0) {
                finish_softmax();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(VAR & 1)) __builtin_amdgcn_s_setprio(2);       // the MFMA stream wins the issue arbitration; the partner's VALU fills its gaps
            if (t < NT) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ABL & 2) { asm volatile("" :: "v"(kf[jt][ks])); s[jt] = relw[jt]; }
                        else s[jt] = mfma32(kf[jt][ks], qf[ks], ks == 0 ? relw[jt] : s[jt]);
                    }
            }
            if (t > 0) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const half4_t h0 = __builtin_bit_cast(half4_t, vv[jt][st][dt][0]);
                            const half4_t h1 = __builtin_bit_cast(half4_t, vv[jt][st][dt][1]);
                            half8_t vf = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                            if (ABL & 2) asm volatile("" :: "v"(vf), "v"(pf[jt][st]));
                            else o[dt] = mfma32(vf, pf[jt][st], o[dt]);
                        }
            }
            if (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
        } else {
            if ((VAR & 4) && t > 0) finish_softmax();
            if (t < NT) scores(t);
            if (t > 0) values(t - 1);
        }
        if (t < NT) {
            if (t + 1 < NT) write_k((t + 1) & 1);
            write_v(t & 1);
            if (t + 2 < NT) load_k(t + 2);
            if (t + 1 < NT) load_v(t + 1);
        }
    };

    if ((VAR & 8) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);   // T5, static form
    if (group == 1) slot_end();                  // group B runs one slot behind group A
    unsigned long long tm = 0, tmb = 0, tx = 0, txb = 0, c_start = 0, r_start = 0, c_loop = 0, r_loop = 0;
    if (ABL == 4) {
        // diagnostic build: shader cycles of this wave's M work, X work and the barrier waits behind them; the totals
        // overwrite the beginning of `out` AFTER the regular epilogue (so nothing is optimised away)
        c_start = __builtin_amdgcn_s_memtime();
        r_start = __builtin_amdgcn_s_memrealtime();
        for (int t = 0; t < NT; ++t) {
            const unsigned long long a0 = __builtin_amdgcn_s_memtime();
            m_slot(t);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long a1 = __builtin_amdgcn_s_memtime();
            slot_end();
            const unsigned long long a2 = __builtin_amdgcn_s_memtime();
            softmax(t);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long a3 = __builtin_amdgcn_s_memtime();
            slot_end();
            const unsigned long long a4 = __builtin_amdgcn_s_memtime();
            tm += a1 - a0; tmb += a2 - a1; tx += a3 - a2; txb += a4 - a3;
        }
        m_slot(NT);
        slot_end();
        if (group == 0) slot_end();
        c_loop = __builtin_amdgcn_s_memtime() - c_start;
        r_loop = __builtin_amdgcn_s_memrealtime() - r_start;
    } else {
    for (int t = 0; t < NT; ++t) {
        m_slot(t);
        slot_end();
        softmax(t);
        slot_end();
    }
    m_slot(NT);
    slot_end();
    if (group == 0) slot_end();                  // barrier counts of the two groups match
    }

    l += swap_halves(l);
    const float inv_l = 1.0f / l;
    half_t* orow = out + ((size_t)img * TOKENS + qtok) * D + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = dt * 32 + 8 * g4 + 4 * hi;
            if (d0 < HD) {
                half4_t v = {(half_t)(o[dt][g4 * 4 + 0] * inv_l), (half_t)(o[dt][g4 * 4 + 1] * inv_l),
                             (half_t)(o[dt][g4 * 4 + 2] * inv_l), (half_t)(o[dt][g4 * 4 + 3] * inv_l)};
                *reinterpret_cast<half4_t*>(orow + d0) = v;
            }
        }
    }
    if (ABL == 4) {
        __syncthreads();
        if (lane == 0 && (wave & 3) == 0 && blockIdx.x % 3 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // inside this workgroup's own output: rows 0 and 1 of its 256, the head's 64 columns (128 bytes each)
            half_t* own = out + ((size_t)img * TOKENS + qblk * 256) * D + head * HD;
            unsigned long long* d = reinterpret_cast<unsigned long long*>(own) + group * 4;
            d[0] = tm; d[1] = tmb; d[2] = tx; d[3] = txb;
            if (group == 0) {
                unsigned long long* e = reinterpret_cast<unsigned long long*>(own + D);
                e[0] = c_loop; e[1] = r_loop; e[2] = c_start - c_kernel; e[3] = r_start;
            }
        }
    }
}


constexpr int kPPVariant = 3;        // VAR of the product build (see attention_global_pp_kernel): 105 -> 92 us per ViT-B launch

template <int HD>
void launch_global(const half_t* qkv, const half_t* rel_h, const half_t* rel_w, half_t* out, int B, int heads,
                   hipStream_t s) {
    const size_t tiles = 2 * ((size_t)KT * (HD + 8) + (size_t)KT * V_STRIDE) * 2;
    const size_t scratch = 4 * 32 * GW_STRIDE * 4;
    const size_t lds = 4 * 64 * RELH_STRIDE * 4 + (tiles > scratch ? tiles : scratch);
    const size_t pp_lds = 8 * 64 * RELH_STRIDE * 4 + tiles;
#ifndef DLIMG_TUNING
    constexpr bool pingpong = true;
#else
    static const bool pingpong = [] { const char* e = std::getenv("DLIMGEDIT_ATTN_PP"); return !e || std::atoi(e) != 0; }();
#endif
#ifdef DLIMG_TUNING     // tuning build only (python -m dlimgedit_amd.build --tuning): ablated variants with WRONG results
    static const int ablate = [] { const char* e = std::getenv("DLIMGEDIT_ATTN_ABLATE"); return e ? std::atoi(e) : 0; }();
    static const int pp_abl = [] { const char* e = std::getenv("DLIMGEDIT_ATTN_PP_ABLATE"); return e ? std::atoi(e) : 0; }();
    static const int pp_var = [] { const char* e = std::getenv("DLIMGEDIT_ATTN_VAR"); return e ? std::atoi(e) : -1; }();
    if (pingpong && !ablate && !pp_abl && pp_var >= 0 && pp_var <= 15) {
        typedef void (*PPK)(const half_t*, const half_t*, const half_t*, half_t*, int);
        static const PPK variants[16] = {
            attention_global_pp_kernel<HD, 0, 0>,  attention_global_pp_kernel<HD, 0, 1>,  attention_global_pp_kernel<HD, 0, 2>,
            attention_global_pp_kernel<HD, 0, 3>,  attention_global_pp_kernel<HD, 0, 4>,  attention_global_pp_kernel<HD, 0, 5>,
            attention_global_pp_kernel<HD, 0, 6>,  attention_global_pp_kernel<HD, 0, 7>,  attention_global_pp_kernel<HD, 0, 8>,
            attention_global_pp_kernel<HD, 0, 9>,  attention_global_pp_kernel<HD, 0, 10>, attention_global_pp_kernel<HD, 0, 11>,
            attention_global_pp_kernel<HD, 0, 12>, attention_global_pp_kernel<HD, 0, 13>, attention_global_pp_kernel<HD, 0, 14>,
            attention_global_pp_kernel<HD, 0, 15>};
        static k::LdsOptIn once[16];
        once[pp_var].ensure((const void*)variants[pp_var], pp_lds, "attention_global: the device refuses the kernel's LDS size");
        hipLaunchKernelGGL(variants[pp_var], dim3(B * heads * (TOKENS / 256)), dim3(512), pp_lds, s, qkv, rel_h, rel_w, out, heads);
        return;
    }
    if (HD == 64 && pingpong && !ablate && pp_abl == 4 && pp_var == 3) {       // slot stamps of variant 3
        auto ppk = attention_global_pp_kernel<HD, 4, 3>;
        static k::LdsOptIn once;
        once.ensure((const void*)ppk, pp_lds, "attention_global: the device refuses the kernel's LDS size");
        hipLaunchKernelGGL(ppk, dim3(B * heads * (TOKENS / 256)), dim3(512), pp_lds, s, qkv, rel_h, rel_w, out, heads);
        return;
    }
    if (HD == 64 && pingpong && !ablate && pp_abl >= 1 && pp_abl <= 4) {
        auto ppk = pp_abl == 1 ? attention_global_pp_kernel<HD, 1> : pp_abl == 2 ? attention_global_pp_kernel<HD, 2>
                   : pp_abl == 3 ? attention_global_pp_kernel<HD, 3> : attention_global_pp_kernel<HD, 4>;
        static k::LdsOptIn once[5];
        once[pp_abl].ensure((const void*)ppk, pp_lds, "attention_global: the device refuses the kernel's LDS size");
        hipLaunchKernelGGL(ppk, dim3(B * heads * (TOKENS / 256)), dim3(512), pp_lds, s, qkv, rel_h, rel_w, out, heads);
        return;
    }
    if (HD == 64 && ablate >= 1 && ablate <= 3) {
        auto kern = ablate == 1 ? attention_global_kernel<HD, 1> : ablate == 2 ? attention_global_kernel<HD, 2>
                                                                                : attention_global_kernel<HD, 3>;
        static k::LdsOptIn once[4];
        once[ablate].ensure((const void*)kern, lds, "attention_global: the device refuses the kernel's LDS size");
        hipLaunchKernelGGL(kern, dim3(B * heads * (TOKENS / 128)), dim3(256), lds, s, qkv, rel_h, rel_w, out, heads);
        return;
    }
#endif
    if (pingpong) {
        static k::LdsOptIn once;       // one per template instance; state per device (lanes and replicas launch concurrently)
        once.ensure((const void*)attention_global_pp_kernel<HD, 0, kPPVariant>, pp_lds, "attention_global: the device refuses the kernel's LDS size");
        hipLaunchKernelGGL((attention_global_pp_kernel<HD, 0, kPPVariant>), dim3(B * heads * (TOKENS / 256)), dim3(512), pp_lds, s, qkv,
                           rel_h, rel_w, out, heads);
        return;
    }
    static k::LdsOptIn once4;
    once4.ensure((const void*)attention_global_kernel<HD, 0>, lds, "attention_global: the device refuses the kernel's LDS size");
    hipLaunchKernelGGL((attention_global_kernel<HD, 0>), dim3(B * heads * (TOKENS / 128)), dim3(256), lds, s, qkv, rel_h, rel_w,
                       out, heads);
}

}  // namespace

namespace k {

void attention_global(const half_t* qkv, const half_t* rel_h, const half_t* rel_w, half_t* out, int B, int heads, int hd,
                      hipStream_t s) {
    if (B <= 0 || heads <= 0) throw_error("attention_global: empty problem");
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)rel_h | (uintptr_t)rel_w) & 15)
        throw_error("attention_global: buffers must be 16-byte aligned");
    switch (hd) {
    case 64: return launch_global<64>(qkv, rel_h, rel_w, out, B, heads, s);
    case 80: return launch_global<80>(qkv, rel_h, rel_w, out, B, heads, s);
    default: throw_error("attention_global: head dimension must be 64 or 80");
    }
}

}  // namespace k
}  // namespace dlimg

// Global self-attention of the SAM ViT encoder (4 of the blocks attend over all 64x64 tokens),
// flash-style: the 4096x4096 score matrix is never materialised.  In the reference this is part of
// the encoder ONNX graph run by Session::run (/root/reference/src/segmentation.cpp:126-128).
//
//   S[i,j] = scale * q_i.k_j + q_i.Rh[qy_i - ky_j + 63] + q_i.Rw[qx_i - kx_j + 63]   (raw q in the bias)
//
// Contract (k::attention_global, kernels.hpp): the q columns of `qkv` arrive multiplied by log2(e) / sqrt(hd) and the
// rel-pos tables by sqrt(hd) -- both folded into the weights when they are loaded (sam_model.cpp) -- so q'.k_j is the
// scaled score and q'.R' the bias, both in units of log2: a score IS the argument of v_exp_f32.
//
// One workgroup = 256 consecutive queries (four rows of the token grid) of one (image, head); 8 waves x 32 queries.
// Keys are streamed in tiles of 64 = one grid row, so inside a tile ky is constant and kx = 0..63: the bias is a
// per-query scalar rh(t) plus a per-query 64-vector relw that lives in registers in accumulator layout.
// Operands are swapped (S^T = K . Q^T) so a lane owns one query: the softmax statistics are lane-local and P tiles
// feed the second MFMA straight from the accumulator registers (O^T = V^T . P^T).  V stays row-major in LDS; the
// transposed MFMA operand comes from ds_read_b64_tr_b16 on 192-byte rows (conflict-free).
//
// Structure (r05; round 1-4 forms and what was measured on them: LABNOTES.md):
//   * two groups of four waves (one wave of each per SIMD) run the same program ONE BARRIER APART; a wave alternates
//     between an M slot (matrix pipe) and an X slot (vector ALU), so both units of a SIMD are fed all the time:
//       M(t): S(t) = bias + K(t).Q'^T and O += V(t-1)^T.P(t-1)^T, 2 + 2*KS + 4*DT MFMAs, with the LDS requests of a
//             rolling window of operands between them and this wave's share of the K / V staging;
//       X(t): the head of M(t+1)'s operand window, then P(t) = exp2(S(t)): per score one v_exp_f32, one v_add_f32 (row
//             sum) and half a v_cvt_pk_f16_f32 -- nothing else.  The vector ALU's issue port is what bounds the kernel
//             (counters: VALU-class issue of the two waves of a SIMD adds up to more than the MFMA time), so
//             everything a score lacks is added by the MATRIX pipe:
//   * the rel-pos row term rh(t) and the reference maximum enter through one extra MFMA per 32 keys whose A operand
//     is ones and whose B operand holds, per query, (hi, lo) f16 pairs of rh(t) -- precomputed in LDS, one
//     ds_read_b32 per tile -- and of -m_ref; the column term relw is that MFMA's C operand;
//   * the reference maximum m_ref is LAZY (cdna_hip_programming.md T13): the exact maximum of tile 0, raised only when
//     a tile's partial row sum exceeds 2^8 relative to it (then every p <= 2^8, far inside f16; the normalisation by l
//     at the end is exact whatever the reference).  The exponentials of a tile wait for no maximum; the rare tile
//     that crosses the threshold is redone in its X slot: P(t-1).V(t-1) is complete and S(t+1) not begun, so O and l
//     are rescaled there, everything at the old reference exactly once;
//   * K / V tiles go global -> registers -> LDS (three buffers each: a fragment is read one slot before it is used);
//     the key loop is unrolled by three so ring buffers are immediate offsets.
#include "device_common.hpp"
#include "kernels.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace dlimg {
namespace {

constexpr int GRID = 64;
constexpr int TOKENS = 4096;
constexpr int KT = 64;              // keys per tile
constexpr int V_STRIDE = 96;        // elements per key row of the V tile (192 B: conflict-free ds_read_b64_tr_b16)
constexpr int RELH_STRIDE = 32;     // words: relh_pk[wave][ky][query]
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_t lds_short4_t;

//   group A: slot 2t = M(t), slot 2t+1 = X(t);  group B one slot later.
//   K(t+2) is written in slots 2t (A), 2t+1 (B) over K(t-1) (last read in slots 2t-3, 2t-2) and read in slots 2t+3, 2t+4.
// ABL (tuning builds only, wrong results; bit mask): 1 = no exponentials, 4 = per-slot cycle stamps written over the
// start of `out`, 8 = no LDS operand requests in the M slot, 16 = no K / V staging, 32 = s_setprio 2 around the M slot
template <int HD, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attention_global_kernel(const half_t* __restrict__ qkv,
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
    constexpr int NT = TOKENS / KT;
    constexpr float kRaise = 256.0f;                     // partial row sum (32 keys) that raises the reference maximum

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* relh_pk = reinterpret_cast<uint32_t*>(smem);                         // [8][64][32] (hi, lo) f16 pairs
    half_t* lds_k = reinterpret_cast<half_t*>(smem + 8 * 64 * RELH_STRIDE * 4);    // [3][K_TILE]
    half_t* lds_v = lds_k + 3 * K_TILE;                                            // [3][V_TILE]

    const int D = heads * HD;
    const int ld = 3 * D;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);   // the 16 query blocks of a head on ONE XCD (its L2 serves K / V)
    const int qblk = bid % (TOKENS / 256);
    const int head = (bid / (TOKENS / 256)) % heads;
    const int img = bid / ((TOKENS / 256) * heads);
    const half_t* base = qkv + (size_t)img * TOKENS * ld + head * HD;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = wave_id();
    const int group = wave >> 2;
    const int hi = lane >> 5, l31 = lane & 31;

    const unsigned long long c_kernel = (ABL & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
    const int qtok = qblk * 256 + wave * 32 + l31;
    const int qy = qblk * 4 + (wave >> 1);              // wave-uniform
    const int qx0 = (wave & 1) * 32;

    // ---- staging: thread -> pieces tid (+512) of a tile; one register set for K and one for V -----------
    // A piece's address is a wave-uniform tile base (scalar arithmetic) plus a 32-bit offset that the thread computes
    // once: no 64-bit vector multiply per request.
    half8_t kreg[NP], vreg[NP];
    uint32_t piece_off[NP], piece_k[NP], piece_v[NP];       // bytes: inside qkv's tile / the K image / the V image
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int idx = it * 512 + tid, key = idx / CHUNKS, ch = idx % CHUNKS;
        piece_off[it] = (uint32_t)(key * ld + ch * 8) * 2u;
        piece_k[it] = (uint32_t)(key * K_STRIDE + ch * 8) * 2u;
        piece_v[it] = (uint32_t)(key * V_STRIDE + ch * 8) * 2u;
    }
    auto piece_ok = [&](int it) { return NP == 1 || it * 512 + tid < PIECES; };
    const char* const kbase = reinterpret_cast<const char*>(base + D);
    const char* const vbase = reinterpret_cast<const char*>(base + 2 * D);
    const size_t tile_bytes = (size_t)KT * ld * 2;
    auto load_piece = [&](const char* tiles, int t, half8_t (&reg)[NP]) {
        const char* tb = tiles + (size_t)t * tile_bytes;
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) reg[it] = *reinterpret_cast<const half8_t*>(tb + piece_off[it]);
    };
    auto write_k = [&](int buf, const half8_t (&reg)[NP]) {
        char* kb = reinterpret_cast<char*>(lds_k + buf * K_TILE);
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) *reinterpret_cast<half8_t*>(kb + piece_k[it]) = reg[it];
    };
    auto write_v = [&](int buf, const half8_t (&reg)[NP]) {
        char* vb = reinterpret_cast<char*>(lds_v + buf * V_TILE);
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (piece_ok(it)) *reinterpret_cast<half8_t*>(vb + piece_v[it]) = reg[it];
    };

    // ---- every global request of the prologue up front: one memory latency instead of eight ------------------------
    half8_t qf[KS], rhf[2][KS], rwf[3][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const half8_t*>(base + (size_t)qtok * ld + ks * 16 + hi * 8);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            rhf[t][ks] = *reinterpret_cast<const half8_t*>(rel_h + (size_t)(qy + t * 32 + l31) * HD + ks * 16 + hi * 8);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = qx0 + t * 32 + l31;                  // rows past 126 do not exist: clamped here, zeroed below
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            rwf[t][ks] = *reinterpret_cast<const half8_t*>(rel_w + (size_t)(row < 2 * GRID - 1 ? row : 2 * GRID - 2) * HD + ks * 16 + hi * 8);
    }
    half8_t k0[NP], k1[NP], v0[NP];
    load_piece(kbase, 0, k0);
    load_piece(kbase, 1, k1);
    load_piece(vbase, 0, v0);
    load_piece(kbase, 2, kreg);
    load_piece(vbase, 1, vreg);

    // scores in units of log2: the caller hands over q' = q * log2(e) / sqrt(hd) and rel-pos tables R' = R * sqrt(hd)
    // (both folded into the weights at load time), so q'.k is the scaled score and q'.R' the bias term q.R, times log2(e)
    auto pack2 = [](half_t a, half_t b) { return __builtin_bit_cast(uint32_t, half2_t{a, b}); };
    auto pstamp = [&]() -> unsigned long long {
        if (!(ABL & 4)) return 0ull;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    const unsigned long long p1 = pstamp();

    // ---- rel-pos tables via MFMA ----------------------------------------------------------------------------------
    // relw[i][kx] = q_i . rel_w[qx_i - kx + 63] in accumulator layout (the C operand of the bias MFMA).  The MFMA gives,
    // for query i (a lane), the products with rows d = 0..95 of rel_w[qx0 + d] in accumulator-ROW order; what the lane
    // needs is row d = i - kx + 63 for each of its kx: a per-lane selection of registers, done through LDS.  Each wave
    // uses its own 8 KB of the relh table (filled afterwards) as [kx][query]: product (d, i) goes to kx = i + 63 - d
    // -- every d = 32..63 lands inside 0..63, and of d and d + 64 exactly one does (kx mod 64), so 32 unconditional
    // stores per lane -- and comes back in accumulator order.  Lanes of a wave-instruction hit consecutive words both
    // ways (conflict-free), and every address is one register plus an immediate (the prologue is straight-line code
    // that runs once per workgroup, from a cold instruction cache: its length is its cost).
    char* const gwb = reinterpret_cast<char*>(relh_pk) + wave * (64 * RELH_STRIDE * 4);
    float16_t relw[2];
    const unsigned long long p2 = pstamp();
    {
        float16_t racc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            racc[t] = zero16();
            const bool row_ok = qx0 + t * 32 + l31 < 2 * GRID - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) racc[t] = mfma32(row_ok ? rwf[t][ks] : zero_h8(), qf[ks], racc[t]);
        }
        // byte address of (kx, query i) = 128 * kx + 4 * i.  acc_row(r, hi) = e_r + 4 * hi with e_r a compile-time constant.
        const int d31 = l31 - 4 * hi;                       // i - 4 * hi
        char* const mid = gwb + 132 * l31 - 512 * hi + 128 * (31 - 27);      // d = 32 + rr: kx = i + 31 - rr, + 128 * (27 - e_r)
        const uint32_t wrap = (uint32_t)(128 * (d31 + 63) + 4 * l31);      // d = rr or rr + 64: kx = (i + 63 - rr) mod 64
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e_r = (r & 3) + 8 * (r >> 2);
            *reinterpret_cast<float*>(mid + 128 * (27 - e_r)) = racc[1][r];
            *reinterpret_cast<float*>(gwb + ((wrap - 128u * (uint32_t)e_r) & 8191u)) = d31 <= e_r ? racc[0][r] : racc[2][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own stores have landed (LDS is in order per wave)
        const char* const back = gwb + 4 * l31 + 512 * hi;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                relw[jt][r] = *reinterpret_cast<const float*>(back + 128 * (jt * 32 + (r & 3) + 8 * (r >> 2)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // read before the relh rows below overwrite the scratch
        __builtin_amdgcn_sched_barrier(0);
    }
    // relh[i][ky] = q_i . rel_h[qy - ky + 63]: rows rr = 0..63 <-> rel_h[qy + rr], ky = 63 - rr; kept in LDS as an
    // (hi, lo) f16 pair per (key row, query): that IS the B operand of the bias MFMA
    {
        char* const top = gwb + 4 * l31 - 512 * hi;          // (ky, i) at 128 * ky + 4 * i, ky = 63 - 32 t - e_r - 4 hi
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float16_t acc = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = mfma32(rhf[t][ks], qf[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r];
                const half_t vh = (half_t)v;
                *reinterpret_cast<uint32_t*>(top + 128 * (63 - 32 * t - ((r & 3) + 8 * (r >> 2)))) = pack2(vh, (half_t)(v - (float)vh));
            }
        }
    }
    const unsigned long long p3 = pstamp();
    write_k(0, k0);
    write_k(1, k1);
    if (DT * 32 > HD) {         // columns of V beyond the head dimension stay zero in all three buffers (no V piece covers them)
        constexpr int PADW = (DT * 32 - HD) / 2;             // 32-bit words of padding per key row
        for (int idx = tid; idx < 3 * KT * PADW; idx += 512)
            reinterpret_cast<uint32_t*>(lds_v + (idx / PADW) * V_STRIDE + HD)[idx % PADW] = 0u;
    }
    write_v(0, v0);

    const int tr_off = ((hi * 4 + ((lane & 15) >> 2)) * V_STRIDE) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    // bias MFMA: A = ones in the k slots 0..3 (lanes 0-31; lanes 32-63 hold the slots 8..15: zero), B = per query
    // (rh_hi, rh_lo, -m_hi, -m_lo, 0, 0, 0, 0)
    const half_t one_or_zero = hi == 0 ? (half_t)1.0f : (half_t)0.0f;
    const half8_t ones = {one_or_zero, one_or_zero, one_or_zero, one_or_zero, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    float m_ref = 0.f;          // reference maximum (log2 units); == -(hi + lo) of mref_pk exactly
    uint32_t mref_pk = 0u;
    float l = 0.f;
    float16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
    float16_t s[2];             // S(t), then (in place) nothing: P(t) leaves as f16 fragments
    half8_t pf[2][2];           // P(t) as B-operand fragments [jt][st]
    // LDS operands of an M slot in the order its MFMAs take them: op k < NKOP = K fragment (ks = k / 2, jt = k % 2),
    // then V fragment pairs (jt, st, dt).  MFMA step j (0, 1: the bias MFMAs) takes op j - 2 and, before it is issued,
    // the wave requests op j + AHEAD: every read is AHEAD + 2 MFMAs (~200 cycles) old when it is needed, at most
    // AHEAD + 2 fragments are in flight, and the reads are spread over the slot instead of queueing in front of it
    // (25 requests back to back overflow the 4-bit lgkmcnt and stall the wave ~300 cycles).  Ops 0..AHEAD-1 are
    // requested by the X slot before.
    constexpr int NKOP = 2 * KS, NVOP = 4 * DT, NOPS = NKOP + NVOP;
#if defined(DLIMG_TUNING) && defined(DLIMG_PP2_AHEAD)      // operand window depth, A/B in the tuning build only
    constexpr int AHEAD = DLIMG_PP2_AHEAD;
#else
    constexpr int AHEAD = 4;
#endif
    static_assert(AHEAD <= NKOP, "the X slot requests K fragments only");
    half8_t kop[NKOP];
    short4_t vop[NVOP][2];
    uint32_t rh_pk = 0u;
    if (ABL & 8) {
#pragma unroll
        for (int i = 0; i < NKOP; ++i) kop[i] = zero_h8();
#pragma unroll
        for (int i = 0; i < NVOP; ++i) vop[i][0] = vop[i][1] = short4_t{0, 0, 0, 0};
    }

    auto slot_end = [&]() {     // this wave's LDS traffic of the slot is done, then the workgroup barrier
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // ph = t % 3 is a compile-time constant everywhere below (the key loop is unrolled by three): ring buffers are
    // immediate offsets of two fixed address registers
    const half_t* const k_rd = lds_k + l31 * K_STRIDE + hi * 8;
    const half_t* const v_rd = lds_v + tr_off;
    auto issue_op = [&](int ph, int k) {     // operands of M(t), ph = t % 3: K(t), V(t-1)
        if (k < NKOP) {
            const int ks = k / 2, jt = k % 2;
            kop[k] = *reinterpret_cast<const half8_t*>(k_rd + ph * K_TILE + jt * 32 * K_STRIDE + ks * 16);
        } else {
            const int v = k - NKOP, js = v / DT, dt = v % DT;
            const half_t* vp = v_rd + ((ph + 2) % 3) * V_TILE + js * 16 * V_STRIDE + dt * 32;
            vop[v][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)vp);
            vop[v][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(vp + 8 * V_STRIDE));
        }
    };
    auto fetch_head = [&](int t, int ph) {   // what M(t) needs before its first requests return
        rh_pk = relh_pk[(wave * 64 + t) * RELH_STRIDE + l31];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) issue_op(ph, k);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto m_slot = [&](int t, auto phase, auto with_qk, auto with_pv) {
        constexpr int ph = decltype(phase)::value;
        const uint4_t bw = {rh_pk, mref_pk, 0u, 0u};
        const half8_t bf = __builtin_bit_cast(half8_t, bw);
        if (ABL & 32) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int j = 0; j < 2 + NOPS; ++j) {
            const int k = j + AHEAD;
            if (!(ABL & 8) && k < NOPS && (k < NKOP ? with_qk.value : with_pv.value)) issue_op(ph, k);
            if (!(ABL & 16) && with_qk.value && j == 6) {
                // staging, unconditional: past the last tile the clamped loads fetch tile NT-1 again and the writes
                // land in ring buffers that nobody reads any more
                write_k((ph + 2) % 3, kreg);
                write_v((ph + 1) % 3, vreg);
            }
            if (!(ABL & 16) && with_qk.value && j == 10) {
                load_piece(kbase, t + 3 < NT ? t + 3 : NT - 1, kreg);
                load_piece(vbase, t + 2 < NT ? t + 2 : NT - 1, vreg);
            }
            if (j < 2) {
                if (with_qk.value) s[j] = mfma32(ones, bf, relw[j]);
            } else if (j < 2 + NKOP) {
                const int i = j - 2;
                if (with_qk.value) s[i % 2] = mfma32(kop[i], qf[i / 2], s[i % 2]);
            } else if (with_pv.value) {
                const int v = j - 2 - NKOP, js = v / DT, dt = v % DT;
                const half4_t h0 = __builtin_bit_cast(half4_t, vop[v][0]);
                const half4_t h1 = __builtin_bit_cast(half4_t, vop[v][1]);
                const half8_t vf = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                o[dt] = mfma32(vf, pf[js / 2][js % 2], o[dt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL & 32) __builtin_amdgcn_s_setprio(0);
    };
    auto expo = [&]() {         // P = exp2(S), partial row sum of this lane's 32 keys
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = (ABL & 1) ? s[jt][st * 8 + e] : __builtin_amdgcn_exp2f(s[jt][st * 8 + e]);
                    ps[e & 3] += p;
                    pf[jt][st][e] = (half_t)p;
                }
        return (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
    unsigned long long st_raises = 0;
    auto x_slot = [&](int t, auto phase) {      // phase = t % 3
        fetch_head(t + 1, (decltype(phase)::value + 1) % 3);
        float ts = expo();
        // the rare tile whose scores outgrow the reference (and tile 0, which sets it): raise m_ref to this tile's row
        // maximum, bring O and l to the new reference (P(t-1).V(t-1) is complete, S(t+1) not begun), redo P(t)
        if (__any(!(ts <= kRaise)) || t == 0) {
            float tm = s[0][0];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[jt][r]);
            tm = fmaxf(tm, swap_halves(tm));
            const float m_new = m_ref + (t == 0 ? tm : fmaxf(tm, 0.f));
            const half_t nh = (half_t)(-m_new), nl = (half_t)(-m_new - (float)nh);
            const float m_rep = -((float)nh + (float)nl);       // what the bias MFMA will subtract from now on
            const float delta = m_rep - m_ref;
            m_ref = m_rep;
            mref_pk = pack2(nh, nl);
            if (t != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jt][r] -= delta;
            ts = expo();
            if ((ABL & 4)) ++st_raises;
        }
        l += ts;
    };

    __syncthreads();            // K(0), K(1), V(0) are in LDS
    const unsigned long long p4 = pstamp();
    fetch_head(0, 0);
    if (group == 1) slot_end();                  // group B runs one slot behind group A
    std::integral_constant<bool, true> yes;
    std::integral_constant<bool, false> no;
    unsigned long long tm = 0, tmb = 0, tx = 0, txb = 0, c_start = 0, r_start = 0, c_loop = 0, r_loop = 0;
    auto stamp = [&]() -> unsigned long long {
        if (!(ABL & 4)) return 0ull;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    if ((ABL & 4)) {
        c_start = __builtin_amdgcn_s_memtime();
        r_start = __builtin_amdgcn_s_memrealtime();
    }
    static_assert((NT - 1) % 3 == 0, "the key loop is unrolled by three (the ring buffers are compile-time constants)");
    std::integral_constant<int, 0> ph0;
    std::integral_constant<int, 1> ph1;
    std::integral_constant<int, 2> ph2;
    m_slot(0, ph0, yes, no);
    slot_end();
    for (int t = 0; t < NT - 1; t += 3) {
        const unsigned long long a0 = stamp();
        x_slot(t, ph0);
        const unsigned long long a1 = stamp();
        slot_end();
        const unsigned long long a2 = stamp();
        m_slot(t + 1, ph1, yes, yes);
        const unsigned long long a3 = stamp();
        slot_end();
        const unsigned long long a4 = stamp();
        tx += a1 - a0; txb += a2 - a1; tm += a3 - a2; tmb += a4 - a3;
        x_slot(t + 1, ph1);
        slot_end();
        m_slot(t + 2, ph2, yes, yes);
        slot_end();
        x_slot(t + 2, ph2);
        slot_end();
        m_slot(t + 3, ph0, yes, yes);
        slot_end();
    }
    x_slot(NT - 1, ph0);
    slot_end();
    m_slot(NT, ph1, no, yes);
    slot_end();
    if (group == 0) slot_end();                  // barrier counts of the two groups match
    if ((ABL & 4)) {
        c_loop = __builtin_amdgcn_s_memtime() - c_start;
        r_loop = __builtin_amdgcn_s_memrealtime() - r_start;
    }

    l += swap_halves(l);
    const float inv_l = 1.0f / l;
    // Result: lane = query; registers o[dt][4 g4 ..] = columns dt * 32 + 8 g4 + 4 hi .. + 3, i.e. the two lanes of a query (hi =
    // 0 / 1) hold alternate runs of four columns: 8-byte stores, 16 per lane.  r06: the two lanes trade runs first
    // (v_permlane32_swap: the lower lane's odd run against the upper lane's even run, cdna_hip_programming.md T21), so each
    // stores whole 16-byte pieces -- half the store instructions, and 16 bytes is the width at which a write-through result
    // store (store16_result) costs what a plain one does.
    half_t* orow = out + ((size_t)img * TOKENS + qtok) * D + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (dt * 32 + 16 * j < HD) {         // (compile-time; head dimension 80: the last tile has one such pair)
                const int e = 8 * j, od = 8 * j + 4;          // first registers of the even (g4 = 2 j) and odd (g4 = 2 j + 1) run
                const half2_t e01 = {(half_t)(o[dt][e + 0] * inv_l), (half_t)(o[dt][e + 1] * inv_l)};
                const half2_t e23 = {(half_t)(o[dt][e + 2] * inv_l), (half_t)(o[dt][e + 3] * inv_l)};
                const half2_t o01 = {(half_t)(o[dt][od + 0] * inv_l), (half_t)(o[dt][od + 1] * inv_l)};
                const half2_t o23 = {(half_t)(o[dt][od + 2] * inv_l), (half_t)(o[dt][od + 3] * inv_l)};
                // lanes 0-31 keep their even run and receive the upper lane's even run in place of their odd one;
                // lanes 32-63 receive the lower lane's odd run in place of their even one and keep their odd run
                const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, e01), __builtin_bit_cast(unsigned, o01), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, e23), __builtin_bit_cast(unsigned, o23), false, false);
                const uint4_t piece = {s0[0], s1[0], s0[1], s1[1]};
                store16_result(orow + dt * 32 + 16 * j + 8 * hi, piece);
            }
        }
    }
    if ((ABL & 4)) {
        __syncthreads();
        if (lane == 0 && blockIdx.x % 3 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            half_t* own = out + ((size_t)img * TOKENS + qblk * 256) * D + head * HD;
            unsigned long long* w = reinterpret_cast<unsigned long long*>(own + 3 * D) + (wave & 3) * 2;
            if (wave >= 4) w = reinterpret_cast<unsigned long long*>(own + 4 * D) + (wave & 3) * 2;
            w[0] = c_kernel; w[1] = p3;
        }
        if (lane == 0 && (wave & 3) == 0 && blockIdx.x % 3 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            half_t* own = out + ((size_t)img * TOKENS + qblk * 256) * D + head * HD;
            unsigned long long* d = reinterpret_cast<unsigned long long*>(own) + group * 4;
            d[0] = tm; d[1] = tmb; d[2] = tx; d[3] = txb;
            if (group == 0) {
                unsigned long long* e = reinterpret_cast<unsigned long long*>(own + D);
                e[0] = c_loop; e[1] = r_loop; e[2] = c_start - c_kernel; e[3] = st_raises;
                unsigned long long* f = reinterpret_cast<unsigned long long*>(own + 2 * D);
                f[0] = p1 - c_kernel; f[1] = p2 - p1; f[2] = p3 - p2; f[3] = p4 - p3;
            }
        }
    }
}

template <int HD>
void launch_global(const half_t* qkv, const half_t* rel_h, const half_t* rel_w, half_t* out, int B, int heads,
                   hipStream_t s) {
    const size_t lds = 8 * 64 * RELH_STRIDE * 4 + 3 * ((size_t)KT * (HD + 8) + (size_t)KT * V_STRIDE) * 2;
    typedef void (*Kern)(const half_t*, const half_t*, const half_t*, half_t*, int);
    Kern kern = attention_global_kernel<HD, 0>;
#ifdef DLIMG_TUNING     // tuning build only (python -m dlimgedit_amd.build --tuning): ablated variants with WRONG results
    static const int abl = [] { const char* e = std::getenv("DLIMGEDIT_ATTN_ABLATE"); return e ? std::atoi(e) : 0; }();
    switch (abl) {
    case 1: kern = attention_global_kernel<HD, 1>; break;
    case 4: kern = attention_global_kernel<HD, 4>; break;
    case 12: kern = attention_global_kernel<HD, 12>; break;
    case 20: kern = attention_global_kernel<HD, 20>; break;
    case 28: kern = attention_global_kernel<HD, 28>; break;
    case 29: kern = attention_global_kernel<HD, 29>; break;
    case 32: kern = attention_global_kernel<HD, 32>; break;
    case 36: kern = attention_global_kernel<HD, 36>; break;
    default: break;
    }
    static k::LdsOptIn once_abl[64];
    k::LdsOptIn& once = once_abl[abl >= 0 && abl < 64 ? abl : 0];
#else
    static k::LdsOptIn once;       // one per template instance; state per device (lanes and replicas launch concurrently)
#endif
    once.ensure((const void*)kern, lds, "attention_global: the device refuses the kernel's LDS size");
    hipLaunchKernelGGL(kern, dim3(B * heads * (TOKENS / 256)), dim3(512), lds, s, qkv, rel_h, rel_w, out, heads);
}

}  // namespace

namespace k {

float attention_global_q_scale(int hd) { return 1.44269504088896341f / std::sqrt((float)hd); }
float attention_global_rel_scale(int hd) { return std::sqrt((float)hd); }

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

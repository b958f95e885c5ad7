// f16 MFMA GEMM with fused epilogues:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces every dense contraction the reference hands to onnxruntime inside
// Session::run (/root/reference/src/session.cpp:119-136): patch embedding, qkv, proj, fc1, fc2,
// neck convolutions (1x1 and im2col'ed 3x3), and the image-side projections of the mask decoder.
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear [out,in] layout), so
// an MFMA fragment is one 16-byte LDS read.  Structure per workgroup (4 waves, WGM x WGN):
//   * BM x BN output tile, BK = 64; each wave owns a (BM/WGM) x (BN/WGN) sub-tile of 32x32 MFMA tiles
//   * operand tiles go HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip) through a ring
//     of 2-4 LDS stages: up to 3 K-tiles in flight behind a counted s_waitcnt vmcnt(N) and ONE raw
//     s_barrier per K-tile (never vmcnt(0) in the steady state)
//   * LDS image is lane-linear (a DMA wave-instruction writes 8 rows x 128 B); the bank-conflict
//     swizzle chunk ^= (row>>1)&7 is applied on the per-lane SOURCE address and again on the read
//     (cdna_hip_programming.md rule 21), making every ds_read_b128 group conflict-free
//   * MFMA operands are swapped (weights as A, activations as B) so a lane's accumulator registers
//     hold 4 CONSECUTIVE output columns of one row: the tile goes to LDS with ds_write_b128
//     (XOR-swizzled, conflict-free) and comes back row-wise, so bias / residual / outputs are
//     all 16-byte, fully coalesced global accesses
//   * XCD-aware workgroup remap so the tiles of one XCD share A panels in its L2
// Epilogue (all optional, fp32; gemm_epilogue.inc): + bias[n], GELU(erf), + residual[m % resid_mod][n], store f32
// and/or f16; the encoder's LayerNorms are folded in (EPI_STATS on the producing GEMM, EPI_NORM on the consuming one).
// Two kernels share all of this: gemm_f16_kernel on v_mfma_f32_32x32x16_f16 (BK 64 or 32, 4 or 8 waves) and
// gemm16_f16_kernel on v_mfma_f32_16x16x32_f16 (BK 32), whose operand fragments travel one K tile ahead in registers.
#include "device_common.hpp"
#include "kernels.hpp"

#include <hip/hip_ext.h>

#include <mutex>
#include <cstdlib>
#include <type_traits>

namespace dlimg {
namespace {

// In-kernel cycle stamps (GemmArgs::stamps) exist in the tuning build only (python -m dlimgedit_amd.build --tuning);
// the product kernels carry neither the branches nor the s_memtime reads.
#ifdef DLIMG_TUNING
#define DLIMG_STAMPS(a) ((a).stamps)
#else
#define DLIMG_STAMPS(a) (static_cast<unsigned long long*>(nullptr))
#endif

constexpr int BK_CHECK = 64;    // K must be a multiple of this for every configuration

// tuning build only: the stream writers skip their residual read (WRONG results) -- what 4 of the epilogue's 10 bytes per
// element cost
#if defined(DLIMG_TUNING) && defined(DLIMG_NO_RESID_READ)
constexpr bool kNoResidRead = true;
#else
constexpr bool kNoResidRead = false;
#endif
// ... and skip the write of the f16 copy: what 2 of the 6 bytes written per element cost
#if defined(DLIMG_TUNING) && defined(DLIMG_NO_COPY_WRITE)
constexpr bool kNoCopyWrite = true;
#else
constexpr bool kNoCopyWrite = false;
#endif

// ... the statistics arithmetic of the stream writers, and the stores of the stream itself
#if defined(DLIMG_TUNING) && defined(DLIMG_NO_STATS_MATH)
constexpr bool kNoStatsMath = true;
#else
constexpr bool kNoStatsMath = false;
#endif
#if defined(DLIMG_TUNING) && defined(DLIMG_NO_STREAM_STORE)
constexpr bool kNoStreamStore = true;
#else
constexpr bool kNoStreamStore = false;
#endif

#include "gemm_f16_tile.inc"

typedef float float4v __attribute__((ext_vector_type(4)));

// Same GEMM on v_mfma_f32_16x16x32_f16 (K tile 32, 64-byte LDS rows).  Same FLOPs per cycle as the 32x32x16
// form, but the chip holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7;
// these GEMMs are clock-limited: the same kernel runs ~1.5x faster on all-zero operands).
//   A operand: lane l holds rows (l & 15), k = 8*(l >> 4) .. +7  -> one 16-byte read of chunk (l >> 4)
//   C/D      : col = l & 15, row = 4*(l >> 4) + reg              -> with swapped operands a lane owns 4
//                                                                    consecutive output columns of one row
template <int BM, int BN, int WGM, int WGN, int NSTAGE, int MINW, int ACT, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void gemm16_f16_kernel(k::GemmArgs a) {
    constexpr int BKT = 32;
    constexpr int NW = WGM * WGN;
    constexpr int ROW_BYTES = BKT * 2;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 16, TN = WN / 16;       // 16x16 MFMA tiles per wave
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int LOADS = (BM + BN) * ROW_BYTES / 1024 / NW;
    constexpr int JG = ((WN / 32) % 3 == 0) ? 3 : (((WN / 32) % 2 == 0) ? 2 : 1);     // 32-column groups per slab
    constexpr int CHUNKS = JG * 8;
    constexpr int OUT_BYTES = 32 * CHUNKS * 16;
    static_assert(NW * OUT_BYTES <= NSTAGE * STAGE_BYTES, "output staging must fit in the operand buffers");
    static_assert(TM % 2 == 0 && TN % 2 == 0, "wave tile must be a multiple of 32x32");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = lane_id();
    const int wave = wave_id();
    const int wr = wave / WGN, wc = wave % WGN;
    const int l15 = lane & 15, quad = lane >> 4;

    const int ntn = a.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;

    float4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    float* rowstat = reinterpret_cast<float*>(smem + NSTAGE * STAGE_BYTES);      // auxiliary area behind the ring
    float* colvec = rowstat + 2 * BM;
    const int nk = a.K / BKT;
    auto stage = [&](int kt) {
        char* dst = smem + (kt % NSTAGE) * STAGE_BYTES;
        stage_tile<BM, BKT, NW, 1>(a.A, a.lda, m0, kt * BKT, dst, wave, lane);
        stage_tile<BN, BKT, NW, 1>(a.W, a.ldw, n0, kt * BKT, dst + A_BYTES, wave, lane);
    };
    auto frag = [&](const char* lds, int row) {
        return *reinterpret_cast<const half8_t*>(lds + row * ROW_BYTES + ((quad ^ swz16(row)) << 4));
    };
    ColumnVectors<BM, BN, 64 * NW, EPI> column_vectors;
    RowStats<BM, 64 * NW, EPI> row_stats;
    column_vectors.issue(a, n0);                 // ahead of the operand tiles, consumed behind them
    row_stats.issue(a, m0);
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t)
        if (t < nk) stage(t);
    column_vectors.store(colvec);
    row_stats.finish(a, rowstat);

    // Fragments travel one K tile ahead of the MFMAs that use them: while the matrix pipe works on tile kt (all of
    // its operands already in registers), the wave passes the barrier for tile kt+1 and requests that tile's B
    // fragments and the first half of its A fragments; the second half of the current A fragments is requested at
    // the top of the step and is needed half a step later.  Neither the barrier nor the LDS latency is exposed.
    static_assert(NSTAGE >= 3, "a tile is read one step before it is multiplied");
    constexpr int TH = TM / 2;
    half8_t fb[2][TN], fal[2][TH], fah[TH];
    auto wait_tile = [&](int newer) {            // wave's copies of a tile have landed; `newer` tiles may be in flight
        static_assert((NSTAGE - 2) * LOADS < 64 && NSTAGE <= 8, "vmcnt is a 6-bit counter");
        if (newer >= 6) wait_dma<6 * LOADS>();
        else if (newer == 5) wait_dma<5 * LOADS>();
        else if (newer == 4) wait_dma<4 * LOADS>();
        else if (newer == 3) wait_dma<3 * LOADS>();
        else if (newer == 2) wait_dma<2 * LOADS>();
        else if (newer == 1) wait_dma<LOADS>();
        else wait_dma<0>();
    };
    auto read_next = [&](int buf, int kt) {
        const char* la = smem + (kt % NSTAGE) * STAGE_BYTES;
        const char* lb = la + A_BYTES;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[buf][j] = frag(lb, wc * WN + j * 16 + l15);
#pragma unroll
        for (int i = 0; i < TH; ++i) fal[buf][i] = frag(la, wr * WM + i * 16 + l15);
    };
    // STEADY: far enough from the end that every step prefetches and stages -- no branches, so the whole step is
    // one basic block and the compiler's wait counts stay exact (at a join they collapse to "wait for everything")
    auto step = [&](int kt, auto cur_tag, auto steady_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* la = smem + (kt % NSTAGE) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < TH; ++i) fah[i] = frag(la, wr * WM + (TH + i) * 16 + l15);
        if (STEADY) {
            wait_dma<(NSTAGE - 3) * LOADS>();
            __builtin_amdgcn_s_barrier();        // tile kt+1 is visible; nobody still reads tile kt-1
            read_next(cur ^ 1, kt + 1);
            stage(kt + NSTAGE - 1);
        } else if (kt + 1 < nk) {
            wait_tile(min(NSTAGE - 3, nk - 2 - kt));
            __builtin_amdgcn_s_barrier();
            read_next(cur ^ 1, kt + 1);
            if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1);
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the requests above ahead of the MFMAs below
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[cur][j], fal[cur][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[TH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[cur][j], fah[i], acc[TH + i][j], 0, 0, 0);
    };
    wait_tile(min(NSTAGE - 2, nk - 1));
    __builtin_amdgcn_s_barrier();
    read_next(0, 0);
    // lgkmcnt(0): the loop is entered with nothing pending, like its back edge -- otherwise the compiler's merged
    // view at the loop header makes the first step wait for the requests it has just issued
    __builtin_amdgcn_s_waitcnt(0xC07F);
    using Even = std::integral_constant<int, 0>;
    using Odd = std::integral_constant<int, 1>;
    int kt = 0;
    for (; kt + 2 + (NSTAGE - 1) <= nk; kt += 2) {
        step(kt, Even{}, std::true_type{});
        step(kt + 1, Odd{}, std::true_type{});
    }
    for (; kt < nk; kt += 2) {
        step(kt, Even{}, std::false_type{});
        if (kt + 1 < nk) step(kt + 1, Odd{}, std::false_type{});
    }

    // ---- epilogue: 32-row bands through an LDS slab, as in the 32x32 kernel --------------------------
    __syncthreads();
    char* slab = smem + wave * OUT_BYTES;
    float2_t* rowpart = reinterpret_cast<float2_t*>(smem + NW * OUT_BYTES);   // EPI_STATS: [BM][BN/32] behind the slabs
    static_assert(NW * OUT_BYTES + BM * (BN / 32) * 8 <= NSTAGE * STAGE_BYTES, "tile statistics must fit behind the slabs");
    const int resid_row0 = a.resid ? m0 % a.resid_mod : 0;
    constexpr int NIT = 32 * CHUNKS / 64;
    constexpr int NJ = (WN / 32) / JG, NSLAB = (TM / 2) * NJ;
    // the residual of the next slab is requested while the current one is worked on -- unless the register budget
    // of the tile (MINW waves per SIMD) has no room for a second buffer
    constexpr int RV_BUFS = (MINW >= 4 && NIT > 4) ? 1 : 2;
    float4_t rv[RV_BUFS][NIT];
    if (RV_BUFS == 2) {
        const int pf_row_base = wr * WM, pf_col_base = wc * WN;
        float4_t(&pf_dst)[NIT] = rv[0];
#include "gemm_epilogue_prefetch.inc"
    }
#pragma unroll
    for (int band = 0; band < TM / 2; ++band) {
#pragma unroll
        for (int jg = 0; jg < NJ; ++jg) {
            const int sl = band * NJ + jg;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int row = ii * 16 + l15;
#pragma unroll
                for (int jj = 0; jj < JG * 2; ++jj) {
                    const int chunk = jj * 4 + quad;            // columns 4*chunk .. 4*chunk+3 of the slab
                    *reinterpret_cast<float4v*>(slab + (row * CHUNKS + (chunk ^ (row & 7))) * 16) =
                        acc[band * 2 + ii][jg * JG * 2 + jj];
                }
            }
            if (sl + RV_BUFS - 1 < NSLAB) {
                constexpr int AHEAD = RV_BUFS - 1;
                const int pf_row_base = wr * WM + ((sl + AHEAD) / NJ) * 32;
                const int pf_col_base = wc * WN + ((sl + AHEAD) % NJ) * JG * 32;
                float4_t(&pf_dst)[NIT] = rv[(sl + AHEAD) % RV_BUFS];
#include "gemm_epilogue_prefetch.inc"
            }
            const int row_local_base = wr * WM + band * 32, col_local_base = wc * WN + jg * JG * 32;
            float4_t(&rv_cur)[NIT] = rv[sl % RV_BUFS];
#include "gemm_epilogue.inc"
        }
    }
    if (EPI == EPI_STATS) {
        __syncthreads();
        write_tile_stats<BM, BN>(a, rowpart, m0, n0);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 "ping-pong" kernel on v_mfma_f32_16x16x32_f16 (cdna_hip_programming.md, "The 256^2 8-phase template").
//
// Eight waves, 2 (M) x 4 (N), each owning 128 x 64 of the tile (8 x 4 accumulator tiles of 16 x 16).  The four waves
// with wr = 0 and the four with wr = 1 are two GROUPS, one wave of each per SIMD, that run the same program ONE
// BARRIER APART: per K tile of 64 a wave goes through two phases, each { L: fragment reads from LDS + four DMA
// requests; barrier; M: 32 MFMAs (one 64-row half of its accumulators over the whole K tile); barrier }, and while
// one group is in an M segment the other is in an L segment -- the matrix pipe of every SIMD always has a wave feeding
// it, and LDS reads / DMA issue never sit in front of MFMAs of the same wave.  [The template's four phases of 16 MFMAs
// measured 2700 cycles per K tile against 2048 of MFMA issue: every M segment pays ~80 cycles for its barrier and
// wait; with two phases of 32 MFMAs 2405 -- 4096^3: 2528 -> 2152, 1297 TFLOP/s at the 1.38 GHz the chip holds there.
// The fragment registers are the same 64: both W halves were live across the four phases anyway.]
//
// LDS: 2 buffers (K tile parity) x 4 half-tiles of 16 KB: A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255
// (128 rows x 64 halves, lane-linear DMA image, chunk ^= (row >> 1) & 7 on the source address and on the read).
// All eight waves stage every half-tile (2 DMA wave-instructions each).  Half-tile life cycle, in slots (= intervals
// between barriers; group 0 has L(t,p) in slot 4t+2p and M(t,p) in 4t+2p+1, group 1 one slot later):
//   phase        reads (group's own A half; W half by wave column)          stages
//   A (rows mh0) W cols nh0 and nh1 (8 x b128), A rows mh0 (8)              A0, A1 of tile t+1
//   B (rows mh1) A rows mh1 (8); the W fragments stay in registers          W0, W1 of tile t+2, then wait: tile t+1 landed
// WAR: every L segment ends with lgkmcnt(0) BEFORE its barrier, so a half-tile is overwritten only in a later slot
// than its last read (A0/A1 of the other buffer were last read in phase B of tile t-1, by the later group in slot
// 4t-1, and are restaged from slot 4t on; W of THIS buffer is last read in phase A of this tile by the later group,
// slot 4t+1, and restaged from slot 4t+2 on).  RAW: the vmcnt wait of phase B (the 4 newer requests, W of tile t+2,
// may stay in flight) is passed by every wave before the barrier that ends slot 4t+3; tile t+1 is first read in
// slot 4t+4.
constexpr int kPPHalfBytes = 128 * 128;
constexpr int kPPBufBytes = 4 * kPPHalfBytes;
// Auxiliary area behind the operand buffers: rowstat [BM] (mean, rstd), colvec [2][256] (bias, LayerNorm column sums), then
// EITHER rowpart [BM][4] (EPI_STATS: the waves' 64-column partials) OR the raw row-statistic partials of an EPI_NORM tile as
// the producer left them, [group][BM] (sum, M2), kPPStatGroups groups at most (ping-pong producers leave 3 / 4 / 5 for
// ViT-B / L / H; the 128- and 96-column tiles of a pass without other lanes up to 10)
constexpr int kPPStatGroups = 12;
constexpr int pp_aux_bytes(int bm) { return bm * 8 + 2 * 256 * 4 + (bm * 4 * 8 > kPPStatGroups * bm * 8 ? bm * 4 * 8 : kPPStatGroups * bm * 8); }
constexpr int kPPAuxBytes = pp_aux_bytes(256);
constexpr int kPPLds = 2 * kPPBufBytes + kPPAuxBytes;
static_assert(kPPLds <= 160 * 1024, "gemm_pp_kernel: LDS");

// L segment's end: fragment reads of this wave have returned, then the workgroup barrier.  One statement with a memory
// clobber: the compiler moves no LDS access across it.
DLIMG_DEVICE void pp_barrier_after_reads() {
    __builtin_amdgcn_sched_barrier(0);           // nothing (MFMAs included) moves across the segment boundary
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
DLIMG_DEVICE void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// An f16 pair (hi, lo) of four values in one 16-byte register quadruple: what the residual prefetch buffers hold when the
// residual stream travels as pairs (GemmArgs::resid_h / resid_l)
struct HiLo4 { half4_t h, l; };
DLIMG_DEVICE float4_t hilo_pack(half4_t h, half4_t l) { return __builtin_bit_cast(float4_t, HiLo4{h, l}); }
// v_fma_mix_f32 reads an f16 half of a register as an fp32 operand (op_sel_hi: which sources are f16; op_sel: their high
// half), so the conversions of the pair arithmetic cost nothing: hi + lo in one instruction, v - hi in one.  (Selects on
// src0 / src2 only; DESIGN.md section 6 on why src1 selects of VOP3P instructions are kept out of this build.)
template <bool HIGH> DLIMG_DEVICE float mix_sum(uint32_t hi, uint32_t lo) {          // f16 + f16 -> fp32 (exact)
    float r;
    if (HIGH) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}
template <bool HIGH> DLIMG_DEVICE float mix_rest(float v, uint32_t hi) {             // v - f16 -> fp32
    float r;
    if (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(v));
    return r;
}
DLIMG_DEVICE float4_t hilo_value(float4_t packed) {
    const uint4_t x = __builtin_bit_cast(uint4_t, packed);             // h01, h23, l01, l23
    return float4_t{mix_sum<false>(x[0], x[2]), mix_sum<true>(x[0], x[2]), mix_sum<false>(x[1], x[3]), mix_sum<true>(x[1], x[3])};
}
// v -> (hi, lo): hi = f16(v), lo = f16(v - hi), both round-to-nearest-even
DLIMG_DEVICE HiLo4 hilo_split(float4_t v) {
    HiLo4 out;
    out.h = __builtin_convertvector(v, half4_t);
    const uint2_t hb = __builtin_bit_cast(uint2_t, out.h);
    const float4_t rest = {mix_rest<false>(v[0], hb[0]), mix_rest<true>(v[1], hb[0]), mix_rest<false>(v[2], hb[1]), mix_rest<true>(v[3], hb[1])};
    out.l = __builtin_convertvector(rest, half4_t);
    return out;
}
DLIMG_DEVICE float4_t hilo_load(const half_t* hi, const half_t* lo, size_t offset) {
    return hilo_pack(*reinterpret_cast<const half4_t*>(hi + offset), *reinterpret_cast<const half4_t*>(lo + offset));
}

// Bias, LayerNorm column sums and the row-statistic partials of a ping-pong tile (r05).  They used to arrive through
// registers (ColumnVectors / RowStats: loads ahead of the operand DMA, consumed behind it), and hipcc then waits with
// vmcnt(0) in front of their first use -- register loads and LDS-DMA share the counter but may complete out of order with
// each other, so it cannot count -- i.e. for EVERY operand request of the prologue, and the merge of the partials ran after
// that: 5.4 k cycles of prologue against 2.9 k for the same GEMM without the folded LayerNorm, of a ~40 k cycle
// workgroup.  Now they are DMA requests like the operands (the oldest ones of their wave, so every counted wait of the
// prologue covers them), land in the auxiliary area as they lie in memory, and the partials are merged -- same
// arithmetic, same order -- after the K loop, in front of the epilogue, which is the first to need (mean, rstd).
template <int BM, int EPI>
struct PPAux {
    static constexpr int BN = 256;
    static constexpr int TPR = 512 / BM;                 // threads that share a row in the merge (as RowStats)
    // piece p of the tile's auxiliary data is requested by wave p % 8: 0 = bias, 1 = column sums, 2 + g * GP + h = rows
    // 128 h .. of group g's partials (GP pieces of 1 KB per group; a 64-row tile has half a piece per group)
    static constexpr int GP = BM * 8 > 1024 ? BM * 8 / 1024 : 1;
    static DLIMG_DEVICE void issue(const k::GemmArgs& a, int m0, int n0, float* colvec, float2_t* raw, int wave, int lane) {
        if (wave == 0) {
            if (a.bias) glds16(a.bias + n0 + lane * 4, colvec);
            else *reinterpret_cast<float4_t*>(colvec + lane * 4) = float4_t{0.f, 0.f, 0.f, 0.f};
        }
        if (EPI != EPI_NORM) return;
        if (wave == 1) glds16(a.ln_colsum + n0 + lane * 4, colvec + BN);
        const int pieces = a.ln_groups * GP;
        for (int p = wave; p < pieces + 2; p += 8) {         // (wave-uniform trip count)
            if (p < 2) continue;
            const int g = (p - 2) / GP, h = (p - 2) % GP;
            const float2_t* src = reinterpret_cast<const float2_t*>(a.ln_stats) + (size_t)g * a.M + m0 + h * 128 + lane * 2;
            if (BM * 8 >= 1024 || lane < BM / 2) glds16(src, raw + g * BM + h * 128);
        }
    }
    // after the K loop: (mean, rstd) of every row of the tile from its partials (Chan et al.; the arithmetic and the
    // order of RowStats::finish: sums over the groups sub, sub + TPR, ... per thread, then over the TPR threads of a row)
    static DLIMG_DEVICE void merge(const k::GemmArgs& a, const float2_t* raw, float* rowstat) {
        if (EPI != EPI_NORM) return;
        const int r = threadIdx.x / TPR, sub = threadIdx.x % TPR;
        float s1 = 0.f;
        for (int g = sub; g < a.ln_groups; g += TPR) s1 += raw[g * BM + r][0];
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) s1 += __shfl_xor(s1, o, 64);
        const float n_g = (float)(a.K / a.ln_groups), inv_n_g = 1.0f / n_g;
        const float mean = s1 / (float)a.K;
        float m2 = 0.f;
        for (int g = sub; g < a.ln_groups; g += TPR) {
            const float2_t q = raw[g * BM + r];
            const float dm = q[0] * inv_n_g - mean;
            m2 += q[1] + n_g * dm * dm;
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) m2 += __shfl_xor(m2, o, 64);
        if (sub == 0) reinterpret_cast<float2_t*>(rowstat)[r] = float2_t{mean, rsqrtf(m2 / (float)a.K + a.ln_eps)};
        __syncthreads();
    }
};

// Epilogue of the ping-pong kernels: the wave's (NI * 16) x 64 part of the tile, rows row_base .. of the workgroup's tile.
// PRE: the residual of the whole wave tile was requested before the main loop (pre[band][kk]; only where the
// registers allow it) -- the epilogue is bound by the CU's memory pipe (~30 B/clk: 320 KB per 128 x 256 tile of a
// stream writer), so every byte moved earlier comes off it.
template <int NI, int ACT, int EPI, bool PRE = false>
DLIMG_DEVICE void pp_epilogue(const k::GemmArgs& a, float4v (&acc)[NI][4], char* smem, const float* rowstat, const float* colvec,
                              float2_t* rowpart, int m0, int n0, int row_base, int wc, int wave, int lane,
                              float4_t (*pre)[4] = nullptr) {
    // ---- epilogue.  A lane owns 4 consecutive columns of one row per accumulator tile (row = l15, columns 4 * quad ..):
    // storing that directly touches 16 cache lines per wave-instruction with 32 or 64 bytes each, and the store path
    // pays per line touched (measured: 16-22 k cycles for the tile).  So the wave's 16 x 64 band goes through a private
    // LDS slab (the operand buffers are dead: every wave is past the last barrier) and leaves in whole 128-byte lines.
    //   f16 output only (qkv, fc1): bias / LayerNorm / GELU in registers, f16 slab rows of 128 B, 8 lanes per row
    //   fp32 output (residual stream writers): fp32 slab rows of 256 B, 16 lanes per row; residual, both outputs and
    //   the row statistics after the read-back, where a row's 64 columns sit in 16 adjacent lanes (DPP row)
    constexpr int BN = 256;
    const int l15 = lane & 15, quad = lane >> 4;
    const int col_base = wc * 64 + quad * 4;
    float4_t bias4[4], csum4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias4[j] = *reinterpret_cast<const float4_t*>(colvec + col_base + j * 16);
        if (EPI == EPI_NORM) csum4[j] = *reinterpret_cast<const float4_t*>(colvec + BN + col_base + j * 16);
    }
    char* slab = smem + wave * 8192;             // two slabs of 4 KB per wave, used alternately
    const int resid_row0 = (a.resid || a.resid_h) ? m0 % a.resid_mod : 0;
    if (a.out_f32 == nullptr && a.out_l == nullptr) {
        // ---- f16 rows: slot = 8-byte piece (j*4 + quad) of a 128-byte row, XORed with (row & 7) << 1 (pairs stay adjacent)
        const int rd_row = lane >> 3, rd_chunk = lane & 7;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            char* sl = slab + (i & 1) * 4096;
            float2_t st = float2_t{0.f, 1.f};
            if (EPI == EPI_NORM) st = reinterpret_cast<const float2_t*>(rowstat)[row_base + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4_t v = float4_t{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (EPI == EPI_NORM) v = (v - csum4[j] * st[0]) * st[1];
                v += bias4[j];
                if (ACT == k::ACT_GELU) v = gelu4(v);
                const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                // piece index (j*4 + quad) ^ ((l15&7)<<1): j*4 occupies bits 2-3, the XOR mask bits 1-3
                const int piece = (j * 4 + quad) ^ ((l15 & 7) << 1);
                *reinterpret_cast<half4_t*>(sl + l15 * 128 + piece * 8) = h;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int r = kk * 8 + rd_row;
                const float4_t piece16 = *reinterpret_cast<const float4_t*>(sl + r * 128 + ((rd_chunk ^ (r & 7)) << 4));
                const size_t m = (size_t)(m0 + row_base + i * 16 + r);
                half_t* const dst = a.out_h + m * a.ldc16 + n0 + wc * 64 + rd_chunk * 8;
                store16_result(dst, piece16);
            }
        }
    } else {
        // The residual / f16-copy options are compile-time inside the band loop: a run-time test per load makes the
        // compiler branch around each one and drain the memory counter at every join (cdna_hip_programming.md, "Three
        // .s-level traps" (c)): 11-12 k cycles for four bands instead of 3 k.
        auto fp32_bands = [&](auto resid_tag, auto h_tag) {
            constexpr int RESID = decltype(resid_tag)::value;        // 0 none, 1 fp32, 2 f16 pair (hi + lo)
            constexpr int OUT = decltype(h_tag)::value;              // 0 fp32, 1 fp32 + f16 copy, 2 f16 pair
            constexpr bool HAS_H = OUT == 1;
            // ---- fp32 rows of 64 floats: 16-byte slot (j*4 + quad) ^ (row & 7).  Read back EIGHT columns per lane (two
            // slots), 8 lanes per row, 8 rows per pass, two passes per band: every global access of the band is then a
            // 16-byte one -- the f16 pair's hi and lo halves of 8 columns are 16 bytes each (r05; four columns per lane
            // made them 8-byte accesses, twice as many instructions through the CU's memory pipe, which is what bounds
            // this epilogue).  A row's 64 columns sit in 8 adjacent lanes for the statistics.
            const int rd_row = lane >> 3, rd_col = (lane & 7) * 8;
            // the band's residual, requested one band ahead of its use (its latency would otherwise be paid once per
            // band: 3.5 k cycles per band measured).  Per pass two 16-byte registers: fp32 columns 0-3 / 4-7, or hi / lo.
            float4_t rv[2][4];
            auto request_residual = [&](int i, float4_t (&dst)[4]) {
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    dst[kk * 2] = dst[kk * 2 + 1] = float4_t{0.f, 0.f, 0.f, 0.f};
                    const int r = kk * 8 + rd_row;
                    if (RESID == 1 && !kNoResidRead) {
                        const float* src = a.resid + (size_t)(resid_row0 + row_base + i * 16 + r) * a.ldr + n0 + wc * 64 + rd_col;
                        dst[kk * 2] = *reinterpret_cast<const float4_t*>(src);
                        dst[kk * 2 + 1] = *reinterpret_cast<const float4_t*>(src + 4);
                    }
                    if (RESID == 2 && !kNoResidRead) {
                        const size_t off = (size_t)(resid_row0 + row_base + i * 16 + r) * a.ldrs + n0 + wc * 64 + rd_col;
                        dst[kk * 2] = *reinterpret_cast<const float4_t*>(a.resid_h + off);
                        dst[kk * 2 + 1] = *reinterpret_cast<const float4_t*>(a.resid_l + off);
                    }
                }
            };
            if (!PRE) request_residual(0, rv[0]);
    #pragma unroll
            for (int i = 0; i < NI; ++i) {
                char* sl = slab + (i & 1) * 4096;
                float2_t st = float2_t{0.f, 1.f};
                if (EPI == EPI_NORM) st = reinterpret_cast<const float2_t*>(rowstat)[row_base + i * 16 + l15];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4_t v = float4_t{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (EPI == EPI_NORM) v = (v - csum4[j] * st[0]) * st[1];
                    v += bias4[j];
                    if (ACT == k::ACT_GELU) v = gelu4(v);
                    *reinterpret_cast<float4_t*>(sl + l15 * 256 + (((j * 4 + quad) ^ (l15 & 7)) << 4)) = v;
                }
                if (!PRE && i + 1 < NI) request_residual(i + 1, rv[(i + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int r = kk * 8 + rd_row;
                    const int slot = (lane & 7) * 2;
                    float4_t v0 = *reinterpret_cast<const float4_t*>(sl + r * 256 + ((slot ^ (r & 7)) << 4));
                    float4_t v1 = *reinterpret_cast<const float4_t*>(sl + r * 256 + (((slot + 1) ^ (r & 7)) << 4));
                    const float4_t ra = PRE ? pre[i][kk * 2] : rv[i & 1][kk * 2], rb = PRE ? pre[i][kk * 2 + 1] : rv[i & 1][kk * 2 + 1];
                    if (RESID == 1) {
                        v0 += ra;
                        v1 += rb;
                    } else if (RESID == 2) {
                        // ra = hi of the 8 columns, rb = lo: (h01, h23, l01, l23) per group of four
                        const uint4_t xa = __builtin_bit_cast(uint4_t, ra), xb = __builtin_bit_cast(uint4_t, rb);
                        v0 += hilo_value(__builtin_bit_cast(float4_t, uint4_t{xa[0], xa[1], xb[0], xb[1]}));
                        v1 += hilo_value(__builtin_bit_cast(float4_t, uint4_t{xa[2], xa[3], xb[2], xb[3]}));
                    }
                    const size_t m = (size_t)(m0 + row_base + i * 16 + r);
                    const int col = n0 + wc * 64 + rd_col;
                    if (kNoStreamStore) {
                        if (v0[0] + v1[3] == 123.456f) a.out_h[0] = (half_t)1.f;
                    } else if (OUT == 2) {
                        const HiLo4 p0 = hilo_split(v0), p1 = hilo_split(v1);
                        const uint2_t h0 = __builtin_bit_cast(uint2_t, p0.h), h1 = __builtin_bit_cast(uint2_t, p1.h);
                        const uint2_t l0 = __builtin_bit_cast(uint2_t, p0.l), l1 = __builtin_bit_cast(uint2_t, p1.l);
                        store16_result(a.out_h + m * a.ldc16 + col, uint4_t{h0[0], h0[1], h1[0], h1[1]});
                        store16_result(a.out_l + m * a.ldc16 + col, uint4_t{l0[0], l0[1], l1[0], l1[1]});
                    } else {
                        store16_result(a.out_f32 + m * a.ldc32 + col, v0);
                        store16_result(a.out_f32 + m * a.ldc32 + col + 4, v1);
                    }
                    if (HAS_H && !kNoCopyWrite) {
                        const half8_t h = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3],
                                           (half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
                        store16_result(a.out_h + m * a.ldc16 + col, h);
                    }
                    if (EPI == EPI_STATS) {
                        // the row's 64 columns of this wave are in 8 adjacent lanes: (sum, squared deviations)
                        float s1 = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
                        if (kNoStatsMath) { if ((lane & 7) == 0) rowpart[(row_base + i * 16 + r) * 4 + wc] = float2_t{s1, s1}; continue; }
                        s1 = sum_over_8_lanes(s1);
                        const float4_t d0 = v0 - s1 * (1.0f / 64.0f), d1 = v1 - s1 * (1.0f / 64.0f);
                        float m2 = ((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
                                   ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]));
                        m2 = sum_over_8_lanes(m2);
                        if ((lane & 7) == 0) rowpart[(row_base + i * 16 + r) * 4 + wc] = float2_t{s1, m2};
                    }
                }
            }

        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (a.out_l) {                           // the stream as an f16 pair (stream writers of the encoder)
            if (a.resid_h) fp32_bands(I2{}, I2{});
            else if (a.resid) fp32_bands(I1{}, I2{});
            else fp32_bands(I0{}, I2{});
        } else if (a.resid) {
            if (a.out_h) fp32_bands(I1{}, I1{});
            else fp32_bands(I1{}, I0{});
        } else {
            if (a.out_h) fp32_bands(I0{}, I1{});
            else fp32_bands(I0{}, I0{});
        }
    }
    if (EPI == EPI_STATS) {
        __syncthreads();
        for (int r = threadIdx.x; r < NI * 32; r += 512) {
            float s1 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) s1 += rowpart[r * 4 + g][0];
            const float mean = s1 * (1.0f / (float)BN);
            float m2 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float dm = rowpart[r * 4 + g][0] * (1.0f / 64.0f) - mean;
                m2 += rowpart[r * 4 + g][1] + 64.0f * dm * dm;
            }
            reinterpret_cast<float2_t*>(a.stats_out)[(size_t)(n0 / BN) * a.M + m0 + r] = float2_t{s1, m2};
        }
    }
}

// One tile per workgroup.  [r04, measured and NOT kept (commit e1871cc holds the code and its parity tests): a PERSISTENT
// form -- grid = ceil(tiles / rounds), each workgroup walking its tiles, the next tile's first operands (96 KB of DMA)
// requested before the epilogue, slabs in the one LDS region those leave free, the next tile's vectors / statistics into
// the other parity of the auxiliary area, kernel arguments re-read per phase so that the tile loop does not pin 45 SGPRs.
// Bit-equal to this kernel; but on the two-image shapes qkv 46.2 us against 42.2 here (4 concurrent streams: 31.9 / 29.2),
// fc1 59.0 / 56.0 (54.2 / 51.2), eight images qkv 145 / 134, and the bench 784-787 against 808 images/s on the same box.
// Why the turn-around it removes (3.0-5.6 us of launch + first fetch by the stamps) does not come back: vmcnt counts
// stores and DMA copies in order, so the chained tile's first counted wait stands behind the WHOLE drain of the previous
// epilogue's stores -- which a fresh workgroup overlaps with its own launch for free; the tile loop costs the epilogue
// 15-70 spilled registers whatever is done about it; and with the grid cut to tiles / rounds the short last round of the
// one-tile launch (32 of 288 tiles on an empty chip, at a higher clock) is gone.]
template <int ACT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(k::GemmArgs a) {
    constexpr int BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const int wave = wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, quad = lane >> 4;

    const int ntn = a.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;
    const int nk = a.K / 64;

    float* rowstat = reinterpret_cast<float*>(smem + 2 * kPPBufBytes);
    float* colvec = rowstat + 2 * BM;
    float2_t* rowpart = reinterpret_cast<float2_t*>(colvec + 2 * BN);

    float4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    // ---- DMA sources: piece p = 2*wave + q of a half-tile = rows 8p .. 8p+7, lane -> (row, swizzled chunk)
    const half_t* src_a[2];
    const half_t* src_w[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        src_a[q] = a.A + (size_t)(m0 + row) * a.lda + chunk * 8;
        src_w[q] = a.W + (size_t)(n0 + row) * a.ldw + chunk * 8;
    }
    const size_t a_half = (size_t)128 * a.lda, w_half = (size_t)128 * a.ldw;
    // H: 0 = A rows 0-127, 1 = A rows 128-255, 2 = W rows 0-127, 3 = W rows 128-255
    auto stage = [&](int t, int H) {
        char* dst = smem + (t & 1) * kPPBufBytes + H * kPPHalfBytes + wave * 2048;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const half_t* g = (H < 2 ? src_a[q] + (H & 1) * a_half : src_w[q] + (H & 1) * w_half) + (size_t)t * 64;
            glds16(g, dst + q * 1024);
        }
    };

    // ---- fragment addresses: row l15 of a 16-row tile, chunk 4*ks + quad, swizzle (l15 >> 1) & 7 (tile bases are
    // multiples of 16 rows and do not touch the swizzle bits)
    const int sw = (l15 >> 1) & 7;
    const int off0 = l15 * 128 + ((quad ^ sw) << 4);
    const int off1 = l15 * 128 + (((quad ^ sw) ^ 4) << 4);
    const char* a_base = smem + wr * kPPHalfBytes;                                     // + buffer + (mh*64 + i*16) * 128
    const char* w_base = smem + (2 + (wc >> 1)) * kPPHalfBytes + (wc & 1) * 64 * 128;  // + buffer + (nh*32 + j*16) * 128
    auto frag = [&](const char* p) { return *reinterpret_cast<const half8_t*>(p); };

    const unsigned long long t_start = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_start = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    PPAux<BM, EPI>::issue(a, m0, n0, colvec, rowpart, wave, lane);      // the oldest requests of their wave
    stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
    if (nk > 1) { stage(1, 2); stage(1, 3); }
    if (nk > 1) wait_dma<4>(); else wait_dma<0>();
    pp_barrier();                                // tile 0 is visible to every wave
    if (wr == 1) pp_barrier();                   // group 1 runs one slot behind group 0

    half8_t fa[4][2], fw[2][2][2];               // A: [i][ks]; W: [nh][j][ks]
    auto mfma_quadrant = [&](int mh, int nh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][nh * 2 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[nh][j][ks], fa[i][ks], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto read_a = [&](int buf, int mh) {
        const char* p = a_base + buf * kPPBufBytes + mh * 64 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = frag(p + i * 2048 + off0);
            fa[i][1] = frag(p + i * 2048 + off1);
        }
    };
    auto read_w = [&](int buf, int nh) {
        const char* p = w_base + buf * kPPBufBytes + nh * 32 * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fw[nh][j][0] = frag(p + j * 2048 + off0);
            fw[nh][j][1] = frag(p + j * 2048 + off1);
        }
    };
    // One K tile.  STEADY: tiles t+1 and t+2 exist, nothing is conditional (one basic block: the compiler's own wait
    // counts stay exact).
    auto step = [&](int t, auto buf_tag, auto steady_tag) {
        constexpr int buf = decltype(buf_tag)::value;
        constexpr bool STEADY = decltype(steady_tag)::value;
        // phase A (rows mh0): all of W(t) and the first A half; 32 MFMAs
        read_w(buf, 0);
        read_w(buf, 1);
        read_a(buf, 0);
        if (STEADY || t + 1 < nk) { stage(t + 1, 0); stage(t + 1, 1); }
        pp_barrier_after_reads();
        mfma_quadrant(0, 0);
        mfma_quadrant(0, 1);
        pp_barrier();
        // phase B (rows mh1): the second A half, W fragments still in registers; 32 MFMAs
        read_a(buf, 1);
        if (STEADY || t + 2 < nk) {
            stage(t + 2, 2);
            stage(t + 2, 3);
            wait_dma<4>();                       // everything older than W(t+2) has landed: tile t+1 is complete
        } else {
            wait_dma<0>();
        }
        pp_barrier_after_reads();
        mfma_quadrant(1, 1);
        mfma_quadrant(1, 0);
        pp_barrier();
    };
    using Even = std::integral_constant<int, 0>;
    using Odd = std::integral_constant<int, 1>;
    const unsigned long long t_loop = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_loop = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int t = 0;
    for (; t + 3 < nk; t += 2) {
        step(t, Even{}, std::true_type{});
        step(t + 1, Odd{}, std::true_type{});
    }
    for (; t < nk; t += 2) {
        step(t, Even{}, std::false_type{});
        if (t + 1 < nk) step(t + 1, Odd{}, std::false_type{});
    }
    if (wr == 0) pp_barrier();                   // group 0 waits for group 1's last M segment: barrier counts match
    const unsigned long long t_loop_end = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_loop_end = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memrealtime() : 0ull;

    PPAux<BM, EPI>::merge(a, rowpart, rowstat);
    pp_epilogue<8, ACT, EPI>(a, acc, smem, rowstat, colvec, rowpart, m0, n0, wr * 128, wc, wave, lane);
    if (DLIMG_STAMPS(a) && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the wave's own stores have left
        unsigned long long* d = DLIMG_STAMPS(a) + (size_t)blockIdx.x * 4;
        d[0] = t_loop_end - t_loop;
        d[1] = r_loop_end - r_loop;
        d[2] = ((t_loop - t_start) << 32) | ((__builtin_amdgcn_s_memtime() - t_loop_end) & 0xffffffffull);   // prologue | epilogue cycles
        d[3] = __builtin_amdgcn_s_memrealtime() - r_start;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 128 x 256 x 64 variant of the ping-pong kernel, for the GEMMs whose N gives too few 256 x 256 tiles (proj / fc2 of
// ViT-B: N = 768 -> 96 workgroups instead of 48).  Same two wave groups one barrier apart; a wave owns 64 x 64 (4 x 4
// accumulator tiles), group g the A rows 64g .. 64g+63, and a K tile is ONE read slot (all fragments of the tile: 8 A
// and 8 W reads of 16 bytes, and the six DMA requests of tile t+2) followed by ONE slot of 32 MFMAs -- with two phases
// of 16 MFMAs per tile the barrier and the wait in front of every MFMA slot cost ~100 of 358 cycles per slot
// (1432 cycles per K tile against 1024 of MFMA issue).
// LDS: A in 3 buffers of 16 KB (128 rows), W in 3 buffers of 2 x 16 KB, tile t in buffers t % 3.  Group 0 reads tile t
// in slot 2t, group 1 in slot 2t+1; tile t+2 is requested in those same slots into the buffers of tile t-1, whose last
// reader was group 1 in slot 2t-1.  [Two A buffers would do only with the A request of tile t+2 issued after BOTH
// groups have read tile t, i.e. in a second read slot.]
// vmcnt(6): the six requests of tile t+2 issued in this slot may stay in flight, everything older has landed.
// r04: the same kernel with BM = 64 (a wave owns 32 x 64: two A row tiles, 16 MFMAs per K tile): the N = 768 GEMMs of ONE
// image then make 192 workgroups instead of 96.  Its main loop is bound by the LDS-DMA intake like the 128-row one (40 KB per
// K tile for half the FLOPs), i.e. it costs ~1.7 x the CU time per FLOP -- the wrong trade while other lanes share the chip,
// the right one for a pass that has the GPU to itself (a synchronous caller of slots 3 / 4: fc2 41 -> ~28 us, proj 18 -> ~10).
// Same BN, MFMA, K order, epilogue arithmetic and 64-column statistics groups: the bits do not depend on BM.
constexpr int pp128_a_bytes(int bm) { return bm * 128; }                  // one A buffer: bm rows x 64 halves
constexpr int pp128_w_base(int bm) { return 3 * pp128_a_bytes(bm); }      // A buffers first
constexpr int pp128_operands(int bm) { return 3 * pp128_a_bytes(bm) + 3 * 2 * kPPHalfBytes; }   // 144 KB / 120 KB
constexpr int pp128_lds(int bm) { return pp128_operands(bm) + pp_aux_bytes(bm); }
static_assert(pp128_lds(128) <= 160 * 1024 && pp128_lds(64) <= 160 * 1024, "gemm_pp128_kernel: LDS");

template <int BM, int ACT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pp128_kernel(k::GemmArgs a) {
    static_assert(BM == 128 || BM == 64, "128 or 64 rows per tile");
    constexpr int BN = 256;
    constexpr int NI = BM / 32;                  // 16-row accumulator tiles per wave (the group's BM / 2 rows)
    constexpr int APIECES = BM / 64;             // DMA wave-instructions per wave for one A tile (8 rows each)
    constexpr int LOADS = 4 + APIECES;           // ... for one K tile: W half 0, W half 1, A
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const int wave = wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, quad = lane >> 4;

    const int ntn = a.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;
    const int nk = a.K / 64;

    float* rowstat = reinterpret_cast<float*>(smem + pp128_operands(BM));
    float* colvec = rowstat + 2 * BM;
    float2_t* rowpart = reinterpret_cast<float2_t*>(colvec + 2 * BN);

    float4v acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const half_t* src_a[APIECES];
    const half_t* src_w[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        src_w[q] = a.W + (size_t)(n0 + row) * a.ldw + chunk * 8;
    }
#pragma unroll
    for (int q = 0; q < APIECES; ++q) {          // piece p = APIECES * wave + q of the A tile = rows 8p .. 8p+7
        const int row = (wave * APIECES + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        src_a[q] = a.A + (size_t)(m0 + row) * a.lda + chunk * 8;
    }
    const size_t w_half = (size_t)128 * a.ldw;
    auto stage_a = [&](int t, int abuf) {
        char* dst = smem + abuf * pp128_a_bytes(BM) + wave * (APIECES * 1024);
#pragma unroll
        for (int q = 0; q < APIECES; ++q) glds16(src_a[q] + (size_t)t * 64, dst + q * 1024);
    };
    auto stage_w = [&](int t, int wbuf, int h) {
        char* dst = smem + pp128_w_base(BM) + (wbuf * 2 + h) * kPPHalfBytes + wave * 2048;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(src_w[q] + h * w_half + (size_t)t * 64, dst + q * 1024);
    };

    const int sw = (l15 >> 1) & 7;
    const int off0 = l15 * 128 + ((quad ^ sw) << 4);
    const int off1 = l15 * 128 + (((quad ^ sw) ^ 4) << 4);
    const char* a_base = smem + wr * (BM / 2) * 128;                                         // + A buffer + i * 2048
    const char* w_base = smem + pp128_w_base(BM) + (wc >> 1) * kPPHalfBytes + (wc & 1) * 64 * 128;  // + W buffer + (nh*32 + j*16)*128
    auto frag = [&](const char* p) { return *reinterpret_cast<const half8_t*>(p); };

    PPAux<BM, EPI>::issue(a, m0, n0, colvec, rowpart, wave, lane);      // the oldest requests of their wave
    // tiles 0 and 1 (tile t lives in A buffer and W buffer t % 3)
    stage_a(0, 0); stage_w(0, 0, 0); stage_w(0, 0, 1);
    if (nk > 1) { stage_a(1, 1); stage_w(1, 1, 0); stage_w(1, 1, 1); }
    if (nk > 1) wait_dma<LOADS>(); else wait_dma<0>();
    // Stream writers: the residual of the wave's 64 x 64 part (64 registers, which this tile can afford) is requested
    // now and used in the epilogue.  These requests are younger than everything the main loop's counted waits must
    // see landed, and in-order completion means they can only make those waits stricter while they are in flight.
    constexpr bool PRE = EPI == EPI_STATS;
    float4_t rpre[NI][4];
    if (PRE) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) rpre[i][kk] = float4_t{0.f, 0.f, 0.f, 0.f};
        // (lane mapping of pp_epilogue's read-back: 8 lanes per row, 8 columns each, 8 rows per pass)
        if (a.resid && !kNoResidRead) {          // one uniform branch around all the requests
            const int resid_row0 = m0 % a.resid_mod;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const float* src = a.resid + (size_t)(resid_row0 + wr * (BM / 2) + i * 16 + kk * 8 + (lane >> 3)) * a.ldr + n0 + wc * 64 + (lane & 7) * 8;
                    rpre[i][kk * 2] = *reinterpret_cast<const float4_t*>(src);
                    rpre[i][kk * 2 + 1] = *reinterpret_cast<const float4_t*>(src + 4);
                }
        } else if (a.resid_h) {                  // the residual as an f16 pair: hi and lo of 8 columns, 16 bytes each
            const int resid_row0 = m0 % a.resid_mod;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const size_t off = (size_t)(resid_row0 + wr * (BM / 2) + i * 16 + kk * 8 + (lane >> 3)) * a.ldrs + n0 + wc * 64 + (lane & 7) * 8;
                    rpre[i][kk * 2] = *reinterpret_cast<const float4_t*>(a.resid_h + off);
                    rpre[i][kk * 2 + 1] = *reinterpret_cast<const float4_t*>(a.resid_l + off);
                }
        }
    }
    pp_barrier();
    if (wr == 1) pp_barrier();

    half8_t fa[NI][2], fw[4][2];                 // A: [i][ks]; W: [column tile j of the wave's 64][ks]
    auto mfma_tile = [&] {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    int buf3 = 0;                                // t % 3
    auto step = [&](int t, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* ab = a_base + buf3 * pp128_a_bytes(BM);
        const char* wb = w_base + buf3 * 2 * kPPHalfBytes;
        const int next2 = buf3 == 0 ? 2 : buf3 - 1;      // (t + 2) % 3
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fw[j][0] = frag(wb + j * 2048 + off0);
            fw[j][1] = frag(wb + j * 2048 + off1);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            fa[i][0] = frag(ab + i * 2048 + off0);
            fa[i][1] = frag(ab + i * 2048 + off1);
        }
        if (STEADY || t + 2 < nk) {
            stage_w(t + 2, next2, 0);
            stage_w(t + 2, next2, 1);
            stage_a(t + 2, next2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(%0)\n\ts_barrier" ::"n"(LOADS) : "memory");
        } else {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile();
        pp_barrier();
        buf3 = buf3 == 2 ? 0 : buf3 + 1;
    };
    const unsigned long long t_loop = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_loop = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int t = 0;
    for (; t + 2 < nk; ++t) step(t, std::true_type{});
    for (; t < nk; ++t) step(t, std::false_type{});
    if (wr == 0) pp_barrier();
    const unsigned long long t_loop_end = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_loop_end = DLIMG_STAMPS(a) ? __builtin_amdgcn_s_memrealtime() : 0ull;

    PPAux<BM, EPI>::merge(a, rowpart, rowstat);
    pp_epilogue<NI, ACT, EPI, PRE>(a, acc, smem, rowstat, colvec, rowpart, m0, n0, wr * (BM / 2), wc, wave, lane, PRE ? rpre : nullptr);
    if (DLIMG_STAMPS(a) && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = DLIMG_STAMPS(a) + (size_t)blockIdx.x * 4;
        d[0] = t_loop_end - t_loop;
        d[1] = r_loop_end - r_loop;
        d[2] = (__builtin_amdgcn_s_memtime() - t_loop_end) & 0xffffffffull;
        d[3] = 0;
    }
}

typedef void (*GemmKernel)(k::GemmArgs);

// Picks the epilogue flavour the arguments ask for and launches; LDS above the default limit is opted into once.
struct Timing { hipEvent_t start = nullptr, stop = nullptr; };

void launch_flavour(GemmKernel const (&kernels)[5], k::LdsOptIn (&opt_in)[5], const k::GemmArgs& a, int grid,
                    int threads, size_t lds, hipStream_t s, Timing t) {
    const int index = a.stats_out ? 4 : (a.ln_stats ? 2 : 0) + (a.act == k::ACT_GELU ? 1 : 0);
    // concurrent lanes (and the replicas of several GPUs) launch the same kernels: per-device state, k::LdsOptIn
    if (lds > 48 * 1024)
        opt_in[index].ensure((const void*)kernels[index], lds, "gemm: the device refuses the LDS size of this tile configuration");
    if (t.start && t.stop)
        hipExtLaunchKernelGGL(kernels[index], dim3(grid), dim3(threads), lds, s, t.start, t.stop, 0, a);
    else
        hipLaunchKernelGGL(kernels[index], dim3(grid), dim3(threads), lds, s, a);
}

template <int BM, int BN, int WGM, int WGN, int NSTAGE, int MINW>
void launch16(const k::GemmArgs& a, hipStream_t s, Timing t) {
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 64 + aux_bytes(BM, BN);
    static const GemmKernel kernels[5] = {
        gemm16_f16_kernel<BM, BN, WGM, WGN, NSTAGE, MINW, k::ACT_NONE, EPI_PLAIN>,
        gemm16_f16_kernel<BM, BN, WGM, WGN, NSTAGE, MINW, k::ACT_GELU, EPI_PLAIN>,
        gemm16_f16_kernel<BM, BN, WGM, WGN, NSTAGE, MINW, k::ACT_NONE, EPI_NORM>,
        gemm16_f16_kernel<BM, BN, WGM, WGN, NSTAGE, MINW, k::ACT_GELU, EPI_NORM>,
        gemm16_f16_kernel<BM, BN, WGM, WGN, NSTAGE, MINW, k::ACT_NONE, EPI_STATS>,
    };
    static k::LdsOptIn attr_once[5];
    launch_flavour(kernels, attr_once, a, (a.M / BM) * (a.N / BN), 64 * WGM * WGN, lds, s, t);
}

template <int BM, int BN, int WGM, int WGN, int BKT, int NSTAGE, int MINW>
void launch(const k::GemmArgs& a, hipStream_t s, Timing t) {
    const size_t lds = (size_t)NSTAGE * (BM + BN) * BKT * 2 + aux_bytes(BM, BN);
#ifdef DLIMG_TUNING     // tuning build only (python -m dlimgedit_amd.build --tuning): ablated variants with WRONG results
    static const int ablate = [] { const char* e = std::getenv("DLIMGEDIT_GEMM_ABLATE"); return e ? std::atoi(e) : 0; }();
#endif
    static const GemmKernel kernels[5] = {
#ifdef DLIMG_TUNING
        ablate == 1   ? gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_NONE, EPI_PLAIN, 1>
        : ablate == 2 ? gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_NONE, EPI_PLAIN, 2> :
#endif
        gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_NONE, EPI_PLAIN>,
        gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_GELU, EPI_PLAIN>,
        gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_NONE, EPI_NORM>,
        gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_GELU, EPI_NORM>,
        gemm_f16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, MINW, k::ACT_NONE, EPI_STATS>,
    };
    static k::LdsOptIn attr_once[5];
    launch_flavour(kernels, attr_once, a, (a.M / BM) * (a.N / BN), 64 * WGM * WGN, lds, s, t);
}

void launch_pp(const k::GemmArgs& a, hipStream_t s, Timing t) {
    static const GemmKernel kernels[5] = {
        gemm_pp_kernel<k::ACT_NONE, EPI_PLAIN>, gemm_pp_kernel<k::ACT_GELU, EPI_PLAIN>, gemm_pp_kernel<k::ACT_NONE, EPI_NORM>,
        gemm_pp_kernel<k::ACT_GELU, EPI_NORM>,  gemm_pp_kernel<k::ACT_NONE, EPI_STATS>,
    };
    static k::LdsOptIn attr_once[5];
    launch_flavour(kernels, attr_once, a, (a.M / 256) * (a.N / 256), 512, kPPLds, s, t);
}

template <int BM>
void launch_pp128(const k::GemmArgs& a, hipStream_t s, Timing t) {
    static const GemmKernel kernels[5] = {
        gemm_pp128_kernel<BM, k::ACT_NONE, EPI_PLAIN>, gemm_pp128_kernel<BM, k::ACT_GELU, EPI_PLAIN>,
        gemm_pp128_kernel<BM, k::ACT_NONE, EPI_NORM>,  gemm_pp128_kernel<BM, k::ACT_GELU, EPI_NORM>,
        gemm_pp128_kernel<BM, k::ACT_NONE, EPI_STATS>,
    };
    static k::LdsOptIn attr_once[5];
    launch_flavour(kernels, attr_once, a, (a.M / BM) * (a.N / 256), 512, pp128_lds(BM), s, t);
}

}  // namespace

namespace k {

const char* gemm_check(const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return "gemm: empty problem";
    if (a.M % 64 || a.N % 64 || a.K % BK_CHECK) return "gemm: M, N must be multiples of 64 and K of 64";
    if (a.lda % 8 || a.ldw % 8) return "gemm: operand leading dimensions must be multiples of 8 (16-byte rows)";
    if (a.lda < a.K || a.ldw < a.K) return "gemm: leading dimension smaller than K";
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return "gemm: operands must be 16-byte aligned";
    if (a.resid && (a.resid_mod <= 0 || a.resid_mod % 64)) return "gemm: resid_mod must be a positive multiple of 64";
    if (!a.out_f32 && !a.out_h) return "gemm: no output";
    if (a.out_l && (a.out_f32 || !a.out_h || ((uintptr_t)a.out_l & 7))) return "gemm: an f16-pair result needs out_h and out_l (8-byte aligned) and no out_f32";
    if ((a.resid_h != nullptr) != (a.resid_l != nullptr) || (a.resid_h && a.resid))
        return "gemm: the residual is either fp32 or an f16 pair (resid_h and resid_l)";
    if (a.resid_h && ((((uintptr_t)a.resid_h | (uintptr_t)a.resid_l) & 7) || a.ldrs % 4 || a.resid_mod <= 0 || a.resid_mod % 64))
        return "gemm: f16-pair residual rows must be 8-byte aligned, resid_mod a positive multiple of 64";
    if ((a.bias && ((uintptr_t)a.bias & 15)) || (a.resid && (((uintptr_t)a.resid & 15) || a.ldr % 4)) ||
        (a.out_f32 && (((uintptr_t)a.out_f32 & 15) || a.ldc32 % 4)) ||
        (a.out_h && (((uintptr_t)a.out_h & 7) || a.ldc16 % 4)))
        return "gemm: bias/residual/output rows must be 16-byte (f16 output: 8-byte) aligned";
    if (a.ln_stats && (!a.ln_colsum || ((uintptr_t)a.ln_colsum & 15) || a.ln_groups <= 0 || a.K % a.ln_groups ||
                       a.ln_groups > kStatRegs * 2))
        return "gemm: folded LayerNorm needs aligned column sums and 1..24 statistic groups that divide K";
    if (a.stats_out && (a.ln_stats || a.act != ACT_NONE))
        return "gemm: row statistics cannot be combined with an activation or a folded LayerNorm";
    return nullptr;
}

// Tile configurations.  At batch 1 (M = 4096) a GEMM is only a few hundred workgroups, so what matters
// is how evenly they cover the 256 CUs: every configuration is scored by (fill of the last round of
// workgroup slots) x (relative efficiency of the tile) and the best one is launched.
struct TileCfg { int bm, bn, per_cu; float eff; };
constexpr TileCfg kTiles[] = {
    {128, 384, 1, 1.00f},   // 0: 2x2 waves (64x192 each), BK 64, 2 stages, 128 KB LDS
    {128, 288, 1, 1.00f},   // 1: 4x1 waves (32x288 each), BK 64, 3 stages, 156 KB LDS
    {128, 128, 2, 0.80f},   // 2: 2x2 waves, BK 64, 2 stages, 64 KB LDS
    {128, 96, 1, 0.70f},    // 3: 4x1 waves, BK 64, 4 stages, 112 KB LDS
    {128, 64, 3, 0.55f},    // 4: 2x2 waves, BK 64, 2 stages, 48 KB LDS
    {64, 64, 4, 0.40f},     // 5: 2x2 waves, BK 64, 2 stages, 32 KB LDS
    {256, 256, 1, 0.00f},   // 6: 8 waves 2x4 (128x64 each), BK 32, 4 stages, 128 KB LDS (shared-GPU mode or forced)
    {256, 256, 1, 0.00f},   // 7: as 6 on v_mfma_f32_16x16x32_f16, fragments one K tile ahead (4096^3: 1010 TFLOP/s); forced only
    {128, 128, 2, 0.00f},   // 8: 2x2 waves on 16x16x32, BK 32, 4 stages, 64 KB LDS (forced only until measured)
    {256, 256, 1, 1.60f},   // 9: ping-pong kernel (gemm_pp_kernel): 8 waves in two groups one barrier apart, BK 64
                            //    (4096^3: 1300 TFLOP/s at the 1.4 GHz the chip holds under that load)
    {128, 256, 1, 0.00f},   // 10: 128 x 256 ping-pong kernel (gemm_pp128_kernel), one read slot + one MFMA slot per K tile
    {64, 256, 1, 0.00f},    // 11: the same kernel with 64-row tiles: twice the workgroups for a pass that has the GPU to itself
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

// GemmArgs::shared_gpu -- with several execution lanes the GPU is shared between kernels of different images: tiles
// that leave room for a second workgroup on the CU (<= 64 KB LDS) let those kernels overlap, which is worth more
// than the better isolated efficiency of the one-workgroup-per-CU tiles (measured: +8 % images/s).  It is a property
// of the caller (SamModel knows how many lanes share its device), not process state.
bool gemm_tile_fits(const GemmArgs& a, int tile) {
    if (tile < 0 || tile >= kNumTiles) return false;
    const TileCfg& t = kTiles[tile];
    if ((a.out_l || a.resid_h) && !(tile >= 9 && tile <= 11)) return false;      // f16-pair stream: ping-pong epilogue only
    if (tile >= 9 && tile <= 11 && a.ln_stats && a.ln_groups > kPPStatGroups) return false;   // room for the raw partials
    return a.M % t.bm == 0 && a.N % t.bn == 0 && !((a.resid || a.resid_h) && a.resid_mod % t.bm != 0);
}

int gemm_pick_tile(const GemmArgs& a) {
    if (a.tile >= 0) return gemm_tile_fits(a, a.tile) ? a.tile : -1;     // chosen earlier (pick once, use twice) or forced
    const int forced = -1;
    int best = -1;
    float best_score = -1.f;
    const int unit = (a.unit_rows > 0 && a.M % a.unit_rows == 0) ? a.unit_rows : a.M;   // rows the choice is made for
    // a residual that wraps (row m % resid_mod) must wrap on tile boundaries: the epilogue adds row offsets to the
    // tile's first residual row without a modulo per element
    auto wraps_inside = [&](int bm) { return (a.resid || a.resid_h) && a.resid_mod % bm != 0; };
    const bool shared = a.shared_gpu;
    // 256x256 workgroups use a CU about 2.5x better than 128x128 ones (LDS fill rate per FLOP); with other lanes on
    // the remaining CUs that is worth having even when they cover a quarter of the chip (ViT-H proj / fc2: 80
    // workgroups, +2 % images/s; at 48, ViT-B proj / fc2, the longer kernel costs more than it frees)
    if (shared && forced < 0 && unit % 256 == 0 && a.N % 256 == 0 &&
        (unit / 256) * (a.N / 256) >= 64 && !wraps_inside(256)) {
        // one image with the GPU to itself and fewer than half the CUs covered (ViT-H's proj / fc2: 80 tiles): the
        // 128-row tiles double the workgroups (160); same bits for a stream writer (not for a LayerNorm-folded consumer)
        if (a.alone && !a.ln_stats && a.M == unit && (unit / 256) * (a.N / 256) < 128 && (unit / 128) * (a.N / 256) <= 256 &&
            !wraps_inside(128))
            return 10;
        return 9;
    }
    // The same with the rows of a whole BATCHED pass (several images stacked in M): two images give ViT-B's proj / fc2
    // 96 tiles of 256 x 256.  Tiles 9 and 10 compute the same bits (BN = 256, the same MFMA, K order, epilogue
    // arithmetic and 64-column statistics groups), so the result does not depend on which one a pass uses -- the
    // batch-equals-single tests assert it.
    // (only where a single unit would run tile 10: the other tiles use a different MFMA shape, i.e. another summation order)
    // too few 256 x 256 tiles (ViT-B proj / fc2: 48): the 128 x 256 ping-pong kernel doubles them
#ifdef DLIMG_TUNING     // A/B switches of the tuning build only; the product's choice is not steerable from outside
    static const bool use_pp128 = [] { const char* e = std::getenv("DLIMGEDIT_GEMM_PP128"); return !e || std::atoi(e) != 0; }();
    static const bool batch_pp = [] { const char* e = std::getenv("DLIMGEDIT_GEMM_BATCH_PP"); return !e || std::atoi(e) != 0; }();
    // r06 A/B (the form it proposes): proj -- the stream writer with K = N -- on the 128-row tile, whose
    // 64 spare registers take the residual before the K loop, while fc2 keeps the 256-row tile
    static const bool proj128 = [] { const char* e = std::getenv("DLIMGEDIT_GEMM_PROJ128"); return e && std::atoi(e) != 0; }();
#else
    constexpr bool use_pp128 = true, batch_pp = true, proj128 = false;
#endif
    if (use_pp128 && shared && forced < 0 && unit % 128 == 0 && a.N % 256 == 0 && (unit / 128) * (a.N / 256) >= 64 &&
        !wraps_inside(128)) {
        // (not for a LayerNorm-folded consumer: its row statistics are merged in an order that depends on the tile height
        // -- RowStats, threads per row -- so a consumer that lands in this branch keeps tile 10 whatever the pass looks like;
        // ViT-B / L / H consumers never do: their N gives >= 64 tiles of 256 x 256)
        if (a.ln_stats) return 10;
        if (proj128 && a.resid_h && a.K == a.N) return 10;
        if (batch_pp && a.M % 256 == 0 && (a.M / 256) * (a.N / 256) >= 96 && !wraps_inside(256)) return 9;
        // one image with the GPU to itself: 64-row tiles while they still fit the chip in one round (ViT-B's patch / proj /
        // fc2: 96 -> 192 workgroups; ViT-H's 160 would become 320, more than one round: stays)
        if (a.alone && a.M == unit && unit % 64 == 0 && (unit / 64) * (a.N / 256) <= 256 && !wraps_inside(64)) return 11;
        return 10;
    }
    for (int i = 0; i < kNumTiles; ++i) {
        const TileCfg& t = kTiles[i];
        if (unit % t.bm || a.N % t.bn || wraps_inside(t.bm) || !gemm_tile_fits(a, i)) continue;
        if (i == forced) return i;
        if (shared && forced < 0) {
            // shared GPU: other lanes fill the CUs this launch leaves free, so the only question is operand
            // traffic per FLOP -- the 256x256 tile (128 FLOP/B) whenever it yields enough workgroups,
            // otherwise the tiles that can share a CU
            if (t.per_cu < 2) continue;
        }
        const float eff = t.eff;
        const int blocks = (unit / t.bm) * (a.N / t.bn);
        const int slots = 256 * t.per_cu;
        const int rounds = (blocks + slots - 1) / slots;
        const float score = eff * (float)blocks / (float)(rounds * slots);
        if (score > best_score) { best_score = score; best = i; }
    }
    return best;
}

int gemm_choose_tile(GemmArgs& a) {
    if (const char* err = gemm_check(a)) throw_error(err);
    if (!gemm_tile_fits(a, a.tile)) a.tile = -1;      // a tile set beforehand (test hooks) stays if it can run the problem
    a.tile = gemm_pick_tile(a);
    if (a.tile < 0) throw_error("gemm: no tile configuration fits this shape");
    return kTiles[a.tile].bn;
}

void gemm(const GemmArgs& a, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    if (const char* err = gemm_check(a)) throw_error(err);
    const Timing t{start, stop};
    const int tile = gemm_pick_tile(a);
    if ((a.out_l || a.resid_h) && !(tile >= 9 && tile <= 11)) throw_error("gemm: the f16-pair stream needs a ping-pong tile (N % 256 == 0, shared GPU)");
    switch (tile) {
    case 0: return launch<128, 384, 2, 2, 64, 2, 1>(a, s, t);
    case 1: return launch<128, 288, 4, 1, 64, 3, 1>(a, s, t);
    case 2: return launch<128, 128, 2, 2, 64, 2, 4>(a, s, t);
    case 3: return launch<128, 96, 4, 1, 64, 4, 1>(a, s, t);
    case 4: return launch<128, 64, 2, 2, 64, 2, 3>(a, s, t);
    case 5: return launch<64, 64, 2, 2, 64, 2, 4>(a, s, t);
    case 6: return launch<256, 256, 2, 4, 32, 4, 2>(a, s, t);
    case 7: return launch16<256, 256, 2, 4, 4, 2>(a, s, t);
    case 8: return launch16<128, 128, 2, 2, 4, 2>(a, s, t);
    case 9: return launch_pp(a, s, t);
    case 10: return launch_pp128<128>(a, s, t);
    case 11: return launch_pp128<64>(a, s, t);
    default: throw_error("gemm: no tile configuration fits this shape");
    }
}

}  // namespace k
}  // namespace dlimg

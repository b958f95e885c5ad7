// Windowed self-attention of the SAM ViT encoder blocks (14x14 windows over the 64x64 token grid,
// zero-padded to 70x70 -> 25 windows), with decomposed relative-position bias.
// This arithmetic lives inside the encoder ONNX graph in the reference
// (/root/reference/src/segmentation.cpp:126-128 runs it through Session::run); the published
// definition is the SAM `Attention` / `window_partition` / `add_decomposed_rel_pos` trio.
//
// Semantics reproduced exactly (SURVEY.md §7 "window padding semantics"):
//   * the pad tokens are zeros AFTER LayerNorm, so their q/k/v equal the qkv bias; they take part
//     in the softmax as keys/values, un-masked; pad queries are computed nowhere (discarded);
//   * S[i,j] = scale * q_i.k_j + q_i.Rh[ty_i - ty_j + 13] + q_i.Rw[tx_i - tx_j + 13] with the RAW q.
//
// One workgroup = one (image, window, head); 7 waves, each owning 32 query slots.
// Slots are numbered ty*16 + tx (tx = 14, 15 are dummies), which makes 224 = 7*32 slots and lets
// every MFMA accumulator register know its key's (ty, tx) at compile time.
// Per wave:   G   = rel_pos . Q^T   (MFMA, 27 rows each for h and w)  -> per-lane bias registers
//             S^T = K . Q^T + bias/scale   (swapped operands: a lane owns one query column), ONE pass over the seven
//                   key tiles with a lazily raised reference maximum (r05)
//             O^T = V^T . P^T   with the S^T accumulator tile as the B operand as it stands and V^T read
//                   through ds_read_b64_tr_b16 from V's row-major LDS image; lane = query, so 1 / rowsum is lane-local
//             O leaves through an LDS slab as whole 128-byte rows.
// LDS: K image of the window's 196 tokens + V image of the 224 slots; the rel-pos scratch of the prologue and the output
// slabs alias them: 71 KB (head dimension 64) / 77.5 KB (80) and <= 128 VGPRs -> two workgroups per CU for both.
// r05, by the in-kernel phase stamps (tuning build, DLIMGEDIT_WINDOW_STAMPS=1): unconditional requests in the order Q, tables,
// K / V; wave-local ordering in the prologue; one softmax pass; 49.5 -> 37.5 us per four-image ViT-B launch, ViT-H 96 -> 67.
// [Measured and NOT kept: a workgroup walking several items with the next item's requests issued before the current
// item's stores (the cold start of an item is 2.7-4.4 k of its 21 k cycles).  Inside a loop the compiler hoists everything
// that depends on the thread index alone (~40 registers of addresses and masks, the rel-pos table loads) and, at 128
// registers, keeps it in scratch memory; with those made opaque per item the kernel still needed 8-26 spilled registers,
// and ran 20.4 us against 15.6 for one image, 45.7 against 37.5 for four.]
#include "device_common.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace dlimg {
namespace {

constexpr int WS = 14;            // window size
constexpr int SLOTS = 224;        // 14 rows x 16 (14 real + 2 dummy) columns
constexpr int NW = 5;             // windows per axis (70 / 14)
constexpr int GRID = 64;
constexpr int V_STRIDE = 96;      // elements per key row of the V image (192 B: conflict-free ds_read_b64_tr_b16)
constexpr int G_STRIDE = 33;      // floats per row of the per-wave rel-pos scratch
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_t lds_short4_t;

struct WinSlot { int token; bool dummy; bool pad; };

// slot -> token of window (wy, wx); dummy = not a key at all; pad = zero-padded token (qkv == bias)
DLIMG_DEVICE WinSlot win_slot(int slot, int wy, int wx) {
    int ty = slot >> 4, tx = slot & 15;
    int gy = wy * WS + ty, gx = wx * WS + tx;
    WinSlot s;
    s.dummy = tx >= WS;
    s.pad = !s.dummy && (gy >= GRID || gx >= GRID);
    s.token = gy * GRID + gx;
    return s;
}

// row of a key slot in the K image: the window's tokens packed 14 per row; the two dummy slots of a row share its last token's
DLIMG_DEVICE int k_row(int slot) { const int tx = slot & 15; return (slot >> 4) * WS + (tx < WS ? tx : WS - 1); }

template <int HD>
__global__ __launch_bounds__(448, 4) void attention_window_kernel(const half_t* __restrict__ qkv,
                                                               const half_t* __restrict__ qkv_pad,
                                                               const half_t* __restrict__ rel_h,
                                                               const half_t* __restrict__ rel_w,
                                                               half_t* __restrict__ out, int heads,
                                                               unsigned long long* stamps_arg) {
#ifdef DLIMG_TUNING      // tuning build: s_memtime at the phase boundaries of wave 0, eight values per workgroup
    unsigned long long* const stamps = stamps_arg;
#else
    unsigned long long* const stamps = nullptr;
#endif
    unsigned long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) { if (stamps) tstamp[i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    constexpr int KS = HD / 16;                 // MFMA k-steps over the head dimension
    constexpr int DT = (HD + 31) / 32;          // 32-wide output tiles over the head dimension
    constexpr int K_STRIDE = HD + 8;            // elements; +16 B keeps ds_read_b128 conflict-free
    constexpr int CHUNKS = HD / 8;              // 16-byte chunks per row
    static_assert(DT * 32 <= V_STRIDE, "head dimension tiles must fit the padded V row");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lds_k = reinterpret_cast<half_t*>(smem);                              // [196][K_STRIDE]
    half_t* lds_v = lds_k + WS * WS * K_STRIDE;                                   // [224][V_STRIDE]
    float* lds_g = reinterpret_cast<float*>(smem);          // prologue only: [7 waves][32][G_STRIDE]

    const int D = heads * HD;
    const int ld = 3 * D;
    const int head = blockIdx.x % heads;
    const int win = (blockIdx.x / heads) % (NW * NW);
    const int img = blockIdx.x / (heads * NW * NW);
    const int wy = win / NW, wx = win % NW;
    const half_t* base = qkv + (size_t)img * 4096 * ld;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = wave_id();
    const int hi = lane >> 5, l31 = lane & 31;

    // Every request of the prologue is UNCONDITIONAL (r05): a load under a per-lane condition sits in a branch, and the
    // compiler drains the memory counter where the branch joins -- the old pad path (K / V of a zero-padding token = the
    // qkv bias, fetched and converted per lane) did that once per K / V piece in the 9 of 25 windows that have padding,
    // and the rel-pos tables, requested behind K / V, made the first MFMA wait for the whole window.  Now: a slot that is
    // not a token reads token 0 of its window (a line the window needs anyway) and is replaced afterwards; the order is
    // Q, tables, padding values, K / V, so the rel-pos prologue runs while K / V are still on their way.
    const int token0 = wy * WS * GRID + wx * WS;                     // the window's first token: always real
    // ---- this wave's 32 queries as B-operand fragments ------------------------------------------
    const int qslot = wave * 32 + l31;
    const WinSlot qs = win_slot(qslot, wy, wx);
    const bool q_real = !qs.dummy && !qs.pad;
    half8_t qf[KS];
    {
        const half_t* qrow = base + (size_t)(q_real ? qs.token : token0) * ld + head * HD + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const half8_t*>(qrow + ks * 16);
    }
    half8_t rfh[KS], rfw[KS];
    {
        const int rrow = l31 < 2 * WS - 1 ? l31 : 2 * WS - 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            rfh[ks] = *reinterpret_cast<const half8_t*>(rel_h + rrow * HD + ks * 16 + hi * 8);
            rfw[ks] = *reinterpret_cast<const half8_t*>(rel_w + rrow * HD + ks * 16 + hi * 8);
        }
    }
    // ---- K and V of the window: requested now, parked in registers; their latency runs behind the rel-pos prologue
    // below, whose LDS scratch aliases the K / V images.  Consecutive threads take consecutive 16-byte chunks of one
    // slot (whole global lines per request).  A zero-padding token's k and v are the qkv bias: qkv_pad is laid out like
    // a token's row, so such a slot simply reads THAT row.
    static_assert((SLOTS * CHUNKS) % 448 == 0, "K/V chunks split evenly over the workgroup");
    constexpr int KV_IT = SLOTS * CHUNKS / 448;
    // (head dimension 80: the last two of a thread's five pieces are requested after the rel-pos products, when the
    // tables' fragments are dead -- with all five in front the kernel does not fit 128 registers, and a spilled load
    // result is a wait for that load in the middle of the request phase: 12 k cycles measured)
#if defined(DLIMG_TUNING) && defined(DLIMG_WINDOW_KV_EARLY)      // A/B of the split, tuning build only
    constexpr int KV_EARLY = HD == 64 ? KV_IT : DLIMG_WINDOW_KV_EARLY;
#else
    constexpr int KV_EARLY = HD == 64 ? KV_IT : 3;
#endif
    half8_t kreg[KV_IT], vreg[KV_IT];
    auto request_kv = [&](int it) {
        const int idx = tid + it * 448;
        const WinSlot ws = win_slot(idx / CHUNKS, wy, wx);
        const half_t* row = (ws.pad ? qkv_pad : base + (size_t)(ws.dummy ? token0 : ws.token) * ld) + head * HD + (idx % CHUNKS) * 8;
        kreg[it] = *reinterpret_cast<const half8_t*>(row + D);
        vreg[it] = *reinterpret_cast<const half8_t*>(row + 2 * D);
    };
#pragma unroll
    for (int it = 0; it < KV_EARLY; ++it) request_kv(it);
    __builtin_amdgcn_sched_barrier(0);          // every request is out before the first wait (the scheduler otherwise
                                                // puts the first table's MFMAs in front of the K / V requests)
    stamp(1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (!q_real) qf[ks] = zero_h8();
        if (l31 >= 2 * WS - 1) { rfh[ks] = zero_h8(); rfw[ks] = zero_h8(); }
    }

    // ---- decomposed rel-pos: G[r][i] = rel[r] . q_i  via MFMA, gathered into per-lane registers ----
    // The scratch is private to the wave and a wave's LDS operations execute in order: no workgroup barrier between the
    // scatter of one table's products and their gather, only before K / V are written over the scratch.
    float* g = lds_g + wave * 32 * G_STRIDE;
    const int ty_q = qslot >> 4, tx_q = qslot & 15;
    float bh[WS];       // bias / scale from the key's row, index = key ty
    float bw[8];        // bias / scale from the key's column, index e <-> tx = (e&3) + 8*(e>>2) + 4*hi
    {
        float16_t acc = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma32(rfh[ks], qf[ks], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[l31 * G_STRIDE + acc_row(r, hi)] = acc[r];
        __builtin_amdgcn_sched_barrier(0);      // (the first table's fragments and products are dead before the second's live)
        float16_t acc_w = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc_w = mfma32(rfw[ks], qf[ks], acc_w);
        if (KV_EARLY < KV_IT) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = KV_EARLY; it < KV_IT; ++it) request_kv(it);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ky = 0; ky < WS; ++ky) bh[ky] = g[l31 * G_STRIDE + ty_q + (WS - 1) - ky] * sqrtf((float)HD);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) g[l31 * G_STRIDE + acc_row(r, hi)] = acc_w[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tx = (e & 3) + 8 * (e >> 2) + 4 * hi;
            // dummy key columns (tx >= 14) are removed from the softmax by a -inf bias
            bw[e] = tx < WS ? g[l31 * G_STRIDE + tx_q + (WS - 1) - tx] * sqrtf((float)HD) : -INFINITY;
        }
        __syncthreads();        // scratch is dead; K / V images may be written over it
    }
    stamp(2);

    // ---- K and V (both row-major, padded rows) of the whole window: registers -> LDS ------------------
#pragma unroll
    for (int it = 0; it < KV_IT; ++it) {
        const int idx = tid + it * 448;
        const int slot = idx / CHUNKS, ch = idx % CHUNKS;
        const bool dummy = (slot & 15) >= WS;
        // K image: 196 rows, one per token of the window (a dummy slot's scores are removed by its -inf bias, so
        // score_tile lets it read the row of its neighbour); V image: all 224 slots, dummies zero (their p is 0)
        if (!dummy) *reinterpret_cast<half8_t*>(lds_k + k_row(slot) * K_STRIDE + ch * 8) = kreg[it];
        *reinterpret_cast<half8_t*>(lds_v + slot * V_STRIDE + ch * 8) = dummy ? zero_h8() : vreg[it];
    }
    if (DT * 32 > HD) {         // V columns beyond the head dimension must not hold NaN patterns
        constexpr int PADC = (DT * 32 - HD) / 8;
        for (int idx = tid; idx < SLOTS * PADC; idx += 448)
            *reinterpret_cast<half8_t*>(lds_v + (idx / PADC) * V_STRIDE + HD + (idx % PADC) * 8) = zero_h8();
    }
    __syncthreads();
    stamp(3);

    const float scale = rsqrtf((float)HD);

    // ---- S^T tile jt (32 key slots x this wave's 32 queries) = K . Q^T on top of bias / scale ------------------
    const half_t* kb = lds_k + k_row(l31) * K_STRIDE + hi * 8;
    auto score_tile = [&](int jt) {
        float16_t t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ty = jt * 2 + (r >> 3);
            const int e = (r & 3) + 4 * ((r >> 2) & 1);
            t[r] = bh[ty] + bw[e];                 // both already divided by the softmax scale
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // row of slot jt * 32 + l31 in the packed K image = k_row(l31) + 28 jt: one base address, the rest is immediates
            half8_t kf = *reinterpret_cast<const half8_t*>(kb + (jt * 2 * WS) * K_STRIDE + ks * 16);
            t = mfma32(kf, qf[ks], t);
        }
        return t;
    };
    // ONE pass for both head dimensions (r05), with a reference maximum per query that is raised lazily: the reference
    // starts as the maximum of key tile 0 and is raised -- accumulators and row sum rescaled, a wave-uniform branch -- only
    // when a tile's maximum exceeds it by more than 2^8 in the exponent's units, so probabilities stay below 2^8 (f16: 65504)
    // and the result o / l is the same quotient (tests: test_attention_window_lazy_maximum_rescale forces the branch in
    // every window at head dimension 64 and 80).  [Until r05 a first pass computed the exact maximum and a second one
    // recomputed the scores (head dimension 64: 28 MFMAs, 28 fragment reads and 2.5 k cycles of the workgroup's 21 k) or
    // kept them in 112 registers (head dimension 80, one workgroup per CU); the tuning build still has the two-pass form
    // for head dimension 64 (-DDLIMG_WINDOW_TWO_PASS).]
#if defined(DLIMG_TUNING) && defined(DLIMG_WINDOW_TWO_PASS)
    constexpr bool ONLINE = false;
#else
    constexpr bool ONLINE = true;
#endif
    constexpr bool TWO_PASS = HD == 64 && !ONLINE;
    float16_t kept[(TWO_PASS || ONLINE) ? 1 : 7];
    float m = -INFINITY;
    const float c = scale * 1.44269504088896341f;
    if (!ONLINE) {
#pragma unroll
        for (int jt = 0; jt < 7; ++jt) {
            const float16_t t = score_tile(jt);
            if (!TWO_PASS) kept[jt] = t;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, t[r]);
            // the loops are fully unrolled (bias registers are indexed by the tile); without a fence per tile the scheduler
            // hoists the fragment reads of all seven tiles to the front and the kernel no longer fits 128 registers
            __builtin_amdgcn_sched_barrier(0);
        }
        m = fmaxf(m, swap_halves(m));
    }
    stamp(4);
    float mc = -m * c;
    // The second pass must really recompute: seen through, the compiler keeps the 112 scores of the first pass alive
    // (common subexpressions) and the kernel is back at 180 registers.  The bias registers are made opaque here, so
    // nothing computed from them before this point is known to equal anything computed after it.
    if (TWO_PASS) {
#pragma unroll
        for (int i = 0; i < WS; ++i) asm volatile("" : "+v"(bh[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(bw[i]));
    }

    // ---- O^T = V^T . P^T : the S^T accumulator tile is the B operand as it stands (lane = query); V^T fragments by
    // hardware transpose read.  In each 16-lane group, lane 4q+p points at key row q, columns 4p..4p+3 of a 4-key x
    // 16-column block.  Lane = query for the result too, so the normalisation by the row sum is lane-local.
    const half_t* vb = lds_v + ((hi * 4 + ((lane & 15) >> 2)) * V_STRIDE) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    float16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < 7; ++jt) {
        float16_t t = (TWO_PASS || ONLINE) ? score_tile(jt) : kept[(TWO_PASS || ONLINE) ? 0 : jt];
        if (ONLINE) {
            float tm = fmaxf(fmaxf(t[0], t[1]), t[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) tm = fmaxf(fmaxf(tm, t[r]), t[r + 1]);
            tm = fmaxf(tm, t[15]);
            tm = fmaxf(tm, swap_halves(tm));                    // the query's maximum over the tile's 32 keys, in both halves
            if (jt == 0) {
                m = tm;
                mc = -m * c;
            } else if (__any(__builtin_fmaf(tm, c, mc) > 8.0f)) {
                const float m_new = fmaxf(m, tm);
                const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);     // 1 where the lane's reference stays
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                l *= alpha;
                m = m_new;
                mc = -m * c;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(t[r], c, mc));
            t[r] = pv;
            l += pv;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            half8_t pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (half_t)t[st * 8 + e];
            const int key0 = jt * 32 + st * 16;         // element e <-> key0 + 4*hi + 8*(e>>2) + (e&3)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half_t* vp = vb + key0 * V_STRIDE + dt * 32;
                const short4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)vp);
                const short4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(vp + 8 * V_STRIDE));
                const half4_t h0 = __builtin_bit_cast(half4_t, v0), h1 = __builtin_bit_cast(half4_t, v1);
                half8_t vf = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                o[dt] = mfma32(vf, pf, o[dt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    l += swap_halves(l);
    const float inv_l = 1.0f / l;
    stamp(5);

    // ---- store.  O^T[d][query]: lane = query, registers = 4-runs of d.  Written straight from here a store
    // instruction would touch 32 rows with 8 bytes each; instead the wave's 32 x HD block goes through an LDS slab (the
    // K image is dead once every wave has left the loop above) and leaves as whole rows, 16 bytes per lane.
    __syncthreads();
    stamp(6);
    constexpr int ROWB = HD * 2 + 16;                    // slab row in bytes (padded: the 8-byte writes of a wave spread over the banks)
    static_assert(7 * 32 * ROWB <= WS * WS * K_STRIDE * 2 + SLOTS * V_STRIDE * 2, "output slabs must fit in the K / V images");
    char* slab = smem + wave * 32 * ROWB;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = dt * 32 + 8 * g4;
            if (d0 < HD) {                      // (compile-time: the lane's + 4 hi stays inside the same 8 columns)
                const half4_t v = {(half_t)(o[dt][g4 * 4 + 0] * inv_l), (half_t)(o[dt][g4 * 4 + 1] * inv_l),
                                   (half_t)(o[dt][g4 * 4 + 2] * inv_l), (half_t)(o[dt][g4 * 4 + 3] * inv_l)};
                *reinterpret_cast<half4_t*>(slab + l31 * ROWB + hi * 8 + d0 * 2) = v;
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab is private to the wave: wave-local ordering is enough
    constexpr int NIT = (32 * CHUNKS + 63) / 64;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CHUNKS, ch = idx % CHUNKS;
        if (row < 32) {
            const WinSlot os = win_slot(wave * 32 + row, wy, wx);
            if (!os.dummy && !os.pad) {
                const float4_t v = *reinterpret_cast<const float4_t*>(slab + row * ROWB + ch * 16);
                store16_result(out + ((size_t)img * 4096 + os.token) * D + head * HD + ch * 8, v);
            }
        }
    }
    if (stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[7] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) stamps[(size_t)blockIdx.x * 8 + i] = tstamp[i];
    }
}

template <int HD>
void launch_window(const half_t* qkv, const half_t* bias, const half_t* rel_h, const half_t* rel_w, half_t* out, int B,
                   int heads, hipStream_t s) {
    const size_t images = (size_t)WS * WS * (HD + 8) * 2 + (size_t)SLOTS * V_STRIDE * 2;
    const size_t scratch = 7 * 32 * G_STRIDE * 4;
    const size_t lds = images > scratch ? images : scratch;
    static k::LdsOptIn opt_in;       // one per template instance, state per device; lanes and replicas launch concurrently
    opt_in.ensure((const void*)attention_window_kernel<HD>, lds, "attention_window: the device refuses the kernel's LDS size");
    unsigned long long* stamps = nullptr;
#ifdef DLIMG_TUNING      // DLIMGEDIT_WINDOW_STAMPS=1: per-phase cycles of every launch (median over the workgroups) on stderr
    static const bool want = std::getenv("DLIMGEDIT_WINDOW_STAMPS") != nullptr;
    static unsigned long long* buf = nullptr;
    const int wgs = B * NW * NW * heads;
    if (want) {
        if (!buf) (void)hipMalloc(&buf, (size_t)64 * NW * NW * 64 * 8 * sizeof(unsigned long long));
        if (wgs <= 64 * NW * NW * 64) stamps = buf;
    }
#endif
    hipLaunchKernelGGL(attention_window_kernel<HD>, dim3(B * NW * NW * heads), dim3(448), lds, s, qkv, bias, rel_h,
                       rel_w, out, heads, stamps);
#ifdef DLIMG_TUNING
    if (stamps) {
        static int launches = 0;
        if (++launches % 50 == 0) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)wgs * 8);
            (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            static const char* names[7] = {"requests", "rel-pos", "K/V->LDS", "pass 1", "pass 2", "barrier", "store"};
            std::string line = "[window stamps] " + std::to_string(wgs) + " workgroups, median cycles:";
            for (int p = 0; p < 7; ++p) {
                std::vector<unsigned long long> d(wgs);
                for (int w = 0; w < wgs; ++w) d[w] = h[(size_t)w * 8 + p + 1] - h[(size_t)w * 8 + p];
                std::nth_element(d.begin(), d.begin() + wgs / 2, d.end());
                line += std::string(" ") + names[p] + " " + std::to_string(d[wgs / 2]);
            }
            std::vector<unsigned long long> d(wgs);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int w = 0; w < wgs; ++w) { d[w] = h[(size_t)w * 8 + 7] - h[(size_t)w * 8]; t0 = std::min(t0, h[(size_t)w * 8]); t1 = std::max(t1, h[(size_t)w * 8 + 7]); }
            std::nth_element(d.begin(), d.begin() + wgs / 2, d.end());
            line += " | whole " + std::to_string(d[wgs / 2]) + " | first start to last end " + std::to_string(t1 - t0);
            fprintf(stderr, "%s\n", line.c_str());
        }
    }
#endif
}

}  // namespace

namespace k {

void attention_window(const half_t* qkv, const half_t* qkv_pad, const half_t* rel_h, const half_t* rel_w, half_t* out,
                      int B, int heads, int hd, hipStream_t s) {
    if (B <= 0 || heads <= 0) throw_error("attention_window: empty problem");
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)qkv_pad | (uintptr_t)rel_h | (uintptr_t)rel_w) & 15)
        throw_error("attention_window: buffers must be 16-byte aligned");
    switch (hd) {
    case 64: return launch_window<64>(qkv, qkv_pad, rel_h, rel_w, out, B, heads, s);
    case 80: return launch_window<80>(qkv, qkv_pad, rel_h, rel_w, out, B, heads, s);
    default: throw_error("attention_window: head dimension must be 64 or 80");
    }
}

}  // namespace k
}  // namespace dlimg

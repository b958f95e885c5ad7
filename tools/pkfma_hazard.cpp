// Standalone probe of the instruction pattern behind the wrong element of DESIGN.md section 6 (round 3): the SLP
// vectoriser paired the rows of the decoder's token linears into chains of v_pk_fma_f32 with op_sel / op_sel_hi source
// selects whose HIGH halves are fed by v_mov_b32 shuffles into the odd register of an aligned pair, e.g.
//     v_mov_b32    v15, v10
//     v_pk_fma_f32 v[4:5], v[14:15], v[2:3], v[4:5] op_sel_hi:[1,0,1]
// and the element that went wrong (about once in 1e4 decodes, only with other kernels on the chip) was always the LOW
// half of a pair.  This program runs such sequences (fixed registers, inline asm) ~1e12 times over all CUs and compares
// each result bit for bit with the same sums from plain v_fma_f32 -- alone and beside neighbour kernels on the same SIMDs
// (MFMA only, v_exp_f32 only, LDS + global traffic) -- and bisects the sequence:
//   case 0  the compiler's chain for one pair of rows, as emitted (4 packed FMAs, 6 v_mov shuffles)
//   case 1  the same with `s_nop 0` behind every v_mov, case 2 with `s_nop 1`
//   case 3  ONE packed FMA with a source select (op_sel_hi:[1,0,1]) on settled registers (s_nop 7 on both sides)
//   case 4  ONE packed FMA without any source select on settled registers
//   case 5  four dependent packed FMAs WITHOUT source selects, back to back (the shape of hand-written packed code)
//   case 6  four dependent packed FMAs WITH op_sel_hi:[1,0,1] (scalar broadcast of src1), back to back, no v_mov
//   case 7  case 6 with `s_nop 0` between the FMAs (what hipcc inserts by itself between dependent ones)
//   case 8  settled FMAs; the third takes its LOW lane's src1 from the HIGH register of a pair: op_sel:[0,1,0]
//   case 9  the same four back to back;  case 10  the compiler's chain with that op_sel:[0,1,0] replaced by a {w1, w1} broadcast
//   case 11 the compiler's third FMA (two v_mov + op_sel:[0,1,0] on {w0, w1}) with everything around it settled
//   case 12 / 13  one settled packed FMA with the low-lane select on src0 (op_sel:[1,0,0]) / on src2 (op_sel:[0,0,1])
//   case 14 / 15  v_pk_mul_f32 with op_sel:[1,0] / op_sel:[0,1];  case 16  v_pk_add_f32 with op_sel:[0,1]
//   case 17 / 18  v_fma_mix_f32 the way the GEMM epilogue's f16-pair arithmetic uses it (kernels/gemm.hip, mix_sum / mix_rest):
//                 f16 half + f16 half (selects on src0 and src2: op_sel:[1,0,1] / [0,0,0], op_sel_hi:[1,0,1]) and
//                 fp32 - f16 half (op_sel:[1,0,0] / [0,0,0], op_sel_hi:[1,0,0])
//   hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o tools/_bin/pkfma_hazard tools/pkfma_hazard.cpp
//   tools/_bin/pkfma_hazard [seconds per run]
// (-fno-slp-vectorize: otherwise the REFERENCE sums are paired into the very pattern under test.)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;      // [-1, 1)
}

// registers: row 1 (a) in v8..v11, row 2 (b) in v12..v15, weights w0 w1 w2 in v0 v1 v2, w3 in v38; v3 = w2, v39 = w3
#define LOAD_REGS                                                                                   \
    "v_mov_b32 v8, %[a0]\n\tv_mov_b32 v9, %[a1]\n\tv_mov_b32 v10, %[a2]\n\tv_mov_b32 v11, %[a3]\n\t"  \
    "v_mov_b32 v12, %[b0]\n\tv_mov_b32 v13, %[b1]\n\tv_mov_b32 v14, %[b2]\n\tv_mov_b32 v15, %[b3]\n\t" \
    "v_mov_b32 v0, %[w0]\n\tv_mov_b32 v1, %[w1]\n\tv_mov_b32 v2, %[w2]\n\tv_mov_b32 v38, %[w3]\n\t"    \
    "v_mov_b32 v39, %[w3]\n\tv_mov_b32 v3, %[w2]\n\t"                                               \
    "s_nop 7\n\t"
#define STORE_REGS "s_nop 7\n\tv_mov_b32 %[lo], v46\n\tv_mov_b32 %[hi], v47\n\t"
// lo = b . w (row 2), hi = a . w (row 1), both as fma(x0,w0, fma(x1,w1, fma(x2,w2, fma(x3,w3, 0))))
#define CHAIN(NOP)                                                                                  \
    "v_mov_b32_e32 v4, v15\n\t"                                                                     \
    "v_mov_b32_e32 v5, v11\n\t" NOP                                                                 \
    "v_pk_fma_f32 v[4:5], v[4:5], v[38:39], 0 op_sel_hi:[1,0,0]\n\t"                                \
    "v_mov_b32_e32 v15, v10\n\t" NOP                                                                \
    "v_pk_fma_f32 v[4:5], v[14:15], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n\t"                           \
    "v_mov_b32_e32 v6, v13\n\t"                                                                     \
    "v_mov_b32_e32 v7, v9\n\t" NOP                                                                  \
    "v_pk_fma_f32 v[4:5], v[6:7], v[0:1], v[4:5] op_sel:[0,1,0]\n\t"                                \
    "v_mov_b32_e32 v13, v8\n\t" NOP                                                                 \
    "v_pk_fma_f32 v[46:47], v[12:13], v[0:1], v[4:5] op_sel_hi:[1,0,1]\n\t"
// the pairs {b_k, a_k} built beforehand (settled): v[16:17] = {b3,a3}, v[18:19] = {b2,a2}, v[20:21] = {b1,a1}, v[22:23] = {b0,a0};
// weight pairs for the select-free form: v[24:25] = {w3,w3}, v[26:27] = {w2,w2}, v[28:29] = {w1,w1}, v[30:31] = {w0,w0}
#define BUILD_PAIRS                                                                                  \
    "v_mov_b32 v16, v15\n\tv_mov_b32 v17, v11\n\tv_mov_b32 v18, v14\n\tv_mov_b32 v19, v10\n\t"          \
    "v_mov_b32 v20, v13\n\tv_mov_b32 v21, v9\n\tv_mov_b32 v22, v12\n\tv_mov_b32 v23, v8\n\t"            \
    "v_mov_b32 v24, v38\n\tv_mov_b32 v25, v38\n\tv_mov_b32 v26, v2\n\tv_mov_b32 v27, v2\n\t"            \
    "v_mov_b32 v28, v1\n\tv_mov_b32 v29, v1\n\tv_mov_b32 v30, v0\n\tv_mov_b32 v31, v0\n\t"              \
    "v_mov_b32 v32, 0x447a0000\n\tv_mov_b32 v33, v1\n\t" /* {1000.0, w1}: the LOW lane must select the HIGH register */ \
    "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\t"                                                         \
    "s_nop 7\n\t"
#define FMA_LOSEL(dst, x, w, c) "v_pk_fma_f32 " dst ", " x ", " w ", " c " op_sel:[0,1,0]\n\t"
#define SETTLED(FMA) FMA "s_nop 7\n\t"
#define FMA_SEL(dst, x, w, c) "v_pk_fma_f32 " dst ", " x ", " w ", " c " op_sel_hi:[1,0,1]\n\t"
#define FMA_PLAIN(dst, x, w, c) "v_pk_fma_f32 " dst ", " x ", " w ", " c "\n\t"

template <int CASE>
__device__ __forceinline__ void chain(float4v a /* row 1 */, float4v b /* row 2 */, float4v w, float& lo, float& hi) {
#define OPERANDS                                                                                                            \
    : [lo] "=&v"(lo), [hi] "=&v"(hi)                                                                                        \
    : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),          \
      [b3] "v"(b[3]), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3])                                           \
    : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", \
      "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38", "v39", "v46", "v47"
    if (CASE == 0) asm volatile(LOAD_REGS CHAIN("") STORE_REGS OPERANDS);
    else if (CASE == 1) asm volatile(LOAD_REGS CHAIN("s_nop 0\n\t") STORE_REGS OPERANDS);
    else if (CASE == 2) asm volatile(LOAD_REGS CHAIN("s_nop 1\n\t") STORE_REGS OPERANDS);
    else if (CASE == 3)     // every packed FMA alone on settled registers, WITH the source select
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED(FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]")) SETTLED(FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]"))
                     SETTLED(FMA_SEL("v[4:5]", "v[20:21]", "v[28:29]", "v[4:5]")) SETTLED(FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]")) STORE_REGS OPERANDS);
    else if (CASE == 4)     // every packed FMA alone on settled registers, no source select
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED(FMA_PLAIN("v[4:5]", "v[16:17]", "v[24:25]", "v[4:5]")) SETTLED(FMA_PLAIN("v[4:5]", "v[18:19]", "v[26:27]", "v[4:5]"))
                     SETTLED(FMA_PLAIN("v[4:5]", "v[20:21]", "v[28:29]", "v[4:5]")) SETTLED(FMA_PLAIN("v[46:47]", "v[22:23]", "v[30:31]", "v[4:5]")) STORE_REGS OPERANDS);
    else if (CASE == 5)     // dependent, back to back, no source select
        asm volatile(LOAD_REGS BUILD_PAIRS FMA_PLAIN("v[4:5]", "v[16:17]", "v[24:25]", "v[4:5]") FMA_PLAIN("v[4:5]", "v[18:19]", "v[26:27]", "v[4:5]")
                     FMA_PLAIN("v[4:5]", "v[20:21]", "v[28:29]", "v[4:5]") FMA_PLAIN("v[46:47]", "v[22:23]", "v[30:31]", "v[4:5]") STORE_REGS OPERANDS);
    else if (CASE == 6)     // dependent, back to back, with the source select
        asm volatile(LOAD_REGS BUILD_PAIRS FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]") FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]")
                     FMA_SEL("v[4:5]", "v[20:21]", "v[28:29]", "v[4:5]") FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]") STORE_REGS OPERANDS);
    else if (CASE == 8)     // settled, the third FMA takes w1 for its LOW lane from the HIGH register of {1000, w1}: op_sel:[0,1,0]
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED(FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]")) SETTLED(FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]"))
                     SETTLED(FMA_LOSEL("v[4:5]", "v[20:21]", "v[32:33]", "v[4:5]")) SETTLED(FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]")) STORE_REGS OPERANDS);
    else if (CASE == 9)     // the same four, back to back
        asm volatile(LOAD_REGS BUILD_PAIRS FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]") FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]")
                     FMA_LOSEL("v[4:5]", "v[20:21]", "v[32:33]", "v[4:5]") FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]") STORE_REGS OPERANDS);
    else if (CASE == 10)    // the compiler's chain with its third FMA's op_sel:[0,1,0] replaced by a broadcast of {w1, w1}
        asm volatile(LOAD_REGS BUILD_PAIRS
                     "v_mov_b32_e32 v4, v15\n\tv_mov_b32_e32 v5, v11\n\t"
                     "v_pk_fma_f32 v[4:5], v[4:5], v[38:39], 0 op_sel_hi:[1,0,0]\n\t"
                     "v_mov_b32_e32 v15, v10\n\t"
                     "v_pk_fma_f32 v[4:5], v[14:15], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n\t"
                     "v_mov_b32_e32 v6, v13\n\tv_mov_b32_e32 v7, v9\n\t"
                     "v_pk_fma_f32 v[4:5], v[6:7], v[28:29], v[4:5] op_sel_hi:[1,0,1]\n\t"
                     "v_mov_b32_e32 v13, v8\n\t"
                     "v_pk_fma_f32 v[46:47], v[12:13], v[0:1], v[4:5] op_sel_hi:[1,0,1]\n\t" STORE_REGS OPERANDS);
    else if (CASE == 11)    // the compiler's third FMA alone: v_mov x2, op_sel:[0,1,0] on the {w0, w1} pair the compiler used, everything else settled
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED(FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]")) SETTLED(FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]"))
                     "v_mov_b32_e32 v6, v13\n\tv_mov_b32_e32 v7, v9\n\t"
                     SETTLED(FMA_LOSEL("v[4:5]", "v[6:7]", "v[0:1]", "v[4:5]")) SETTLED(FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]")) STORE_REGS OPERANDS);
    else if (CASE == 12)    // settled; the select on SRC0: {1000, w1} first, op_sel:[1,0,0] (both lanes then use w1)
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED(FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]")) SETTLED(FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]"))
                     SETTLED("v_pk_fma_f32 v[4:5], v[32:33], v[20:21], v[4:5] op_sel:[1,0,0]\n\t") SETTLED(FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]")) STORE_REGS OPERANDS);
    else if (CASE == 13)    // settled; the select on SRC2: the accumulator pair swapped beforehand, op_sel:[0,0,1] op_sel_hi:[1,1,0]
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED(FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]")) SETTLED(FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]"))
                     "v_mov_b32 v34, v5\n\tv_mov_b32 v35, v4\n\ts_nop 7\n\t"
                     SETTLED("v_pk_fma_f32 v[4:5], v[20:21], v[28:29], v[34:35] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\t") SETTLED(FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]")) STORE_REGS OPERANDS);
    else if (CASE == 14)    // v_pk_mul_f32 with the select on src0 -- the form the product's token kernels contain (LayerNorm scale): {1000, w1} * {b1, a1}
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED("v_pk_mul_f32 v[46:47], v[32:33], v[20:21] op_sel:[1,0]\n\t") STORE_REGS OPERANDS);
    else if (CASE == 15)    // v_pk_mul_f32 with the select on src1
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED("v_pk_mul_f32 v[46:47], v[20:21], v[32:33] op_sel:[0,1]\n\t") STORE_REGS OPERANDS);
    else if (CASE == 16)    // v_pk_add_f32 with the select on src1
        asm volatile(LOAD_REGS BUILD_PAIRS SETTLED("v_pk_add_f32 v[46:47], v[20:21], v[32:33] op_sel:[0,1]\n\t") STORE_REGS OPERANDS);
    else if (CASE == 17)    // a[0] / b[0] carry two packed f16 each: LOW result = high halves summed, HIGH result = low halves summed
        asm volatile(LOAD_REGS "v_fma_mix_f32 v46, v8, 1.0, v12 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
                     "v_fma_mix_f32 v47, v8, 1.0, v12 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t" STORE_REGS OPERANDS);
    else if (CASE == 18)    // w1 - high half of a[0], w0 - low half of a[0]
        asm volatile(LOAD_REGS "v_fma_mix_f32 v46, v8, -1.0, v1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 v47, v8, -1.0, v0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t" STORE_REGS OPERANDS);
    else                    // ... with s_nop 0 between them
        asm volatile(LOAD_REGS BUILD_PAIRS FMA_SEL("v[4:5]", "v[16:17]", "v[38:39]", "v[4:5]") "s_nop 0\n\t" FMA_SEL("v[4:5]", "v[18:19]", "v[2:3]", "v[4:5]") "s_nop 0\n\t"
                     FMA_SEL("v[4:5]", "v[20:21]", "v[28:29]", "v[4:5]") "s_nop 0\n\t" FMA_SEL("v[46:47]", "v[22:23]", "v[0:1]", "v[4:5]") STORE_REGS OPERANDS);
}

struct Sample { float a[4], b[4], w[4], lo, hi, ref_lo, ref_hi; };

// out[0] = mismatches in the LOW half (row 2), out[1] = in the HIGH half (row 1), out[2] = chains executed / 2^10
template <int CASE>
__global__ __launch_bounds__(256) void victim(unsigned long long* out, Sample* samples, int iters) {
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u + (unsigned)out[2];
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        float4v a, b, w;
        for (int i = 0; i < 4; ++i) { a[i] = rnd(s) * 2.0f; b[i] = rnd(s) * 2.0f; w[i] = rnd(s); }
        float lo, hi;
        _Float16 ah[2] = {(_Float16)a[0], (_Float16)a[1]}, bh[2] = {(_Float16)b[0], (_Float16)b[1]};
        if (CASE == 17 || CASE == 18) {      // two f16 in one register
            a[0] = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, ah[0]) | ((unsigned)__builtin_bit_cast(unsigned short, ah[1]) << 16));
            b[0] = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, bh[0]) | ((unsigned)__builtin_bit_cast(unsigned short, bh[1]) << 16));
        }
        chain<CASE>(a, b, w, lo, hi);
        float ref_lo = fmaf(b[0], w[0], fmaf(b[1], w[1], fmaf(b[2], w[2], fmaf(b[3], w[3], 0.0f))));
        float ref_hi = fmaf(a[0], w[0], fmaf(a[1], w[1], fmaf(a[2], w[2], fmaf(a[3], w[3], 0.0f))));
        if (CASE == 14 || CASE == 15) { ref_lo = __fmul_rn(b[1], w[1]); ref_hi = __fmul_rn(a[1], w[1]); }
        if (CASE == 16) { ref_lo = __fadd_rn(b[1], w[1]); ref_hi = __fadd_rn(a[1], w[1]); }
        if (CASE == 17) { ref_lo = __fadd_rn((float)ah[1], (float)bh[1]); ref_hi = __fadd_rn((float)ah[0], (float)bh[0]); }
        if (CASE == 18) { ref_lo = __fsub_rn(w[1], (float)ah[1]); ref_hi = __fsub_rn(w[0], (float)ah[0]); }
        const bool wl = __float_as_uint(lo) != __float_as_uint(ref_lo), wh = __float_as_uint(hi) != __float_as_uint(ref_hi);
        if ((wl || wh) && bad_lo + bad_hi == 0) {
            const unsigned long long slot = atomicAdd(&out[3], 1ull);
            if (slot < 8) {
                Sample& q = samples[slot];
                for (int i = 0; i < 4; ++i) { q.a[i] = a[i]; q.b[i] = b[i]; q.w[i] = w[i]; }
                q.lo = lo; q.hi = hi; q.ref_lo = ref_lo; q.ref_hi = ref_hi;
            }
        }
        bad_lo += wl;
        bad_hi += wh;
    }
    if (bad_lo) atomicAdd(&out[0], (unsigned long long)bad_lo);
    if (bad_hi) atomicAdd(&out[1], (unsigned long long)bad_hi);
    if (threadIdx.x == 0) atomicAdd(&out[2], (unsigned long long)iters * 256ull >> 10);
}

// neighbours: kind 1 = MFMA only, 2 = v_exp_f32 only, 4 = LDS + global traffic + cross-lane
__global__ __launch_bounds__(256) void neighbour_mfma(float* sink, int iters) {
    half8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.01f * (threadIdx.x + i)); y[i] = (_Float16)(0.02f * i); }
    float4v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, acc2, 0, 0, 0);
    }
    if (acc[0] + acc2[1] == 12345.678f) sink[0] = acc[1];
}
__global__ __launch_bounds__(256) void neighbour_exp(float* sink, int iters) {
    float e = 0.001f * threadIdx.x, f = 0.002f * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        e = __expf(e * 0.5f) - 1.0f;
        f = __expf(f * 0.25f) - 1.0f;
    }
    if (e + f == 12345.678f) sink[2] = e;
}
__global__ __launch_bounds__(256) void neighbour_valu(float* sink, int iters) {        // plain fp32 FMAs, no matrix / trans unit
    float e = 0.001f * threadIdx.x, f = 0.002f * threadIdx.x, g = 1.0f, h = 0.5f;
    for (int it = 0; it < iters; ++it) {
        e = fmaf(e, 0.999f, f); f = fmaf(f, 0.998f, g); g = fmaf(g, 0.997f, h); h = fmaf(h, 0.996f, e);
    }
    if (e + f + g + h == 12345.678f) sink[3] = e;
}
__global__ __launch_bounds__(256) void neighbour_mem(const float* src, float* sink, int iters, int n) {
    __shared__ float buf[256 * 4];
    float v = 0.f;
    unsigned idx = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const float4v g = *reinterpret_cast<const float4v*>(src + ((idx * 4u + (unsigned)it * 4096u) % (unsigned)n));
        *reinterpret_cast<float4v*>(&buf[threadIdx.x * 4]) = g;
        __syncthreads();
        v += buf[(threadIdx.x * 4 + 37 * it) & 1023];
        v += __shfl_xor(v, 1, 64);
        __syncthreads();
    }
    if (v == 12345.678f) sink[1] = v;
}

template <int CASE>
void run_case(const char* load, int mode, double seconds, unsigned long long* dev_out, Sample* dev_samples, float* dev_buf, int nbuf) {
    hipStream_t sv, s1, s2, s3, s4;
    for (hipStream_t* s : {&sv, &s1, &s2, &s3, &s4}) CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    CHECK(hipMemset(dev_out, 0, 4 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, sv));
    double elapsed = 0;
    int rounds = 0;
    while (elapsed < seconds) {
        // victim: 2 waves per SIMD (512 workgroups of 4 waves on 256 CUs), so neighbours fit beside it on every SIMD
        hipLaunchKernelGGL(victim<CASE>, dim3(512), dim3(256), 0, sv, dev_out, dev_samples, 20000);
        if (mode & 1) hipLaunchKernelGGL(neighbour_mfma, dim3(1024), dim3(256), 0, s1, dev_buf, 120000);
        if (mode & 2) hipLaunchKernelGGL(neighbour_exp, dim3(1024), dim3(256), 0, s2, dev_buf, 60000);
        if (mode & 4) hipLaunchKernelGGL(neighbour_mem, dim3(1024), dim3(256), 0, s3, dev_buf + 16, dev_buf, 4000, nbuf - 4096);
        if (mode & 8) hipLaunchKernelGGL(neighbour_valu, dim3(1024), dim3(256), 0, s4, dev_buf, 120000);
        CHECK(hipEventRecord(e1, sv));
        CHECK(hipStreamSynchronize(sv));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        elapsed = ms * 1e-3;
        ++rounds;
    }
    CHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    CHECK(hipMemcpy(h, dev_out, sizeof(h), hipMemcpyDeviceToHost));
    const double chains = (double)h[2] * 1024.0;
    std::printf("case %d  neighbours %-11s chains %.2fe9  wrong LOW halves %llu (%.1e per chain)  wrong HIGH halves %llu\n", CASE, load, chains / 1e9,
                h[0], (double)h[0] / chains, h[1]);
    if (h[0] + h[1]) {
        Sample smp[8];
        CHECK(hipMemcpy(smp, dev_samples, sizeof(smp), hipMemcpyDeviceToHost));
        for (int i = 0; i < (int)(h[3] < 3 ? h[3] : 3); ++i) {
            const Sample& q = smp[i];
            // which partial sums of the chain does the wrong low half equal?  (stale accumulator = a step was lost)
            const float p3 = fmaf(q.b[3], q.w[3], 0.f), p2 = fmaf(q.b[2], q.w[2], p3), p1 = fmaf(q.b[1], q.w[1], p2);
            const float skip2 = fmaf(q.b[0], q.w[0], fmaf(q.b[1], q.w[1], p3)), skip1 = fmaf(q.b[0], q.w[0], p2), skip3 = fmaf(q.b[0], q.w[0], fmaf(q.b[1], q.w[1], fmaf(q.b[2], q.w[2], 0.f)));
            std::printf("    sample: lo %.9g want %.9g | hi %.9g want %.9g | partial sums p3 %.9g p2 %.9g p1 %.9g | without step 3/2/1: %.9g %.9g %.9g | a.w with b's acc?\n",
                        q.lo, q.ref_lo, q.hi, q.ref_hi, p3, p2, p1, skip3, skip2, skip1);
        }
    }
    std::fflush(stdout);
    for (hipStream_t s : {sv, s1, s2, s3, s4}) CHECK(hipStreamDestroy(s));
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 3.0;
    unsigned long long* dev_out;
    Sample* dev_samples;
    float* dev_buf;
    const int nbuf = 64 << 20;
    CHECK(hipMalloc(&dev_out, 4 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&dev_samples, 8 * sizeof(Sample)));
    CHECK(hipMalloc(&dev_buf, (size_t)nbuf * 4));
    CHECK(hipMemset(dev_buf, 0, (size_t)nbuf * 4));
    const bool only_selects = argc > 2;          // any second argument: only the operand-select cases (8, 12-16)
    if (!only_selects) {
    // which neighbour does it take?
    run_case<0>("none", 0, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<0>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<0>("v_exp", 2, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<0>("lds+global", 4, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<0>("fp32 valu", 8, seconds, dev_out, dev_samples, dev_buf, nbuf);
    // which part of the sequence?  (beside MFMA + v_exp neighbours)
    run_case<1>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<2>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<3>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<4>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<5>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<6>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<7>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    }
    run_case<8>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<9>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<10>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<11>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<12>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<13>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<14>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<15>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<16>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<17>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<18>("mfma", 1, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<17>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    run_case<18>("mfma+v_exp", 3, seconds, dev_out, dev_samples, dev_buf, nbuf);
    return 0;
}

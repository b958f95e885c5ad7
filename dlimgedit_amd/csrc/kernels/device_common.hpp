// Shared device-side helpers for the gfx950 (CDNA4) kernels of the SAM path.
// Wave = 64 lanes; MFMA shape used throughout: v_mfma_f32_32x32x16_f16
//   A operand: lane l holds A[row = l&31][k = 8*(l>>5) + j], j = 0..7
//   B operand: lane l holds B[k = 8*(l>>5) + j][col = l&31]
//   C/D      : lane l, register r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dlimg {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float3_t __attribute__((ext_vector_type(3)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

#define DLIMG_DEVICE __device__ __forceinline__

DLIMG_DEVICE int lane_id() { return threadIdx.x & 63; }
DLIMG_DEVICE int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// row of accumulator register r for a lane in half `hi` (0/1)
DLIMG_DEVICE constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

DLIMG_DEVICE float16_t mfma32(half8_t a, half8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

DLIMG_DEVICE float16_t zero16() {
    float16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

DLIMG_DEVICE half8_t zero_h8() {
    half8_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (half_t)0.f;
    return z;
}

// exact GELU (erf form), as nn.GELU() in the encoder MLP and the decoder upscaling
DLIMG_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.28:
//   erf(z) = 1 - 1 / (1 + a1 z + ... + a6 z^6)^16,  z >= 0,  |err| <= 3e-7
// so with q = 1 / P(|x|)^16 (sqrt 2 folded into the coefficients):  GELU(x) = max(x, 0) - 0.5 |x| q  (both signs).
// One reciprocal and no exponential per value, and everything else on two values at a time (v_pk_fma_f32 /
// v_pk_mul_f32): fc1's epilogue is VALU-bound (12.6 M outputs on the CUs its tiles occupy; the 7.1.26 form with
// exp2 + rcp on single values cost 6.7 k of the tile's 11 k epilogue cycles).  Against the fp64 erf form the fp32
// evaluation is within 7.1e-7 absolute over |x| <= 12; P^16 overflowing to inf for |x| > ~25 gives q = 0, the limit.
DLIMG_DEVICE float2_t gelu_pair_erf(float2_t x) {
    const float2_t ax = {fabsf(x[0]), fabsf(x[1])};
    float2_t p = ax * 5.38297500e-6f + 4.88906359e-5f;       // a6 / 8, a5 / (4 sqrt 2)
    p = p * ax + 3.80035750e-5f;                               // a4 / 4
    p = p * ax + 3.27762634e-3f;                               // a3 / (2 sqrt 2)
    p = p * ax + 2.11410062e-2f;                               // a2 / 2
    p = p * ax + 4.98673463e-2f;                               // a1 / sqrt 2
    p = p * ax + 1.0f;
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const float2_t q = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};       // 1 ulp is plenty
    const float2_t pos = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
    return pos - (ax * 0.5f) * q;
}
// The fc1 epilogue's form (r05): GELU(x) = x * Phi(x) with Phi(x) ~ sigmoid(x (a + b x^2 + c x^4)), the three constants
// fitted to the exact erf form (max |error| 2.5e-5 over all x, at |x| ~ 2-3; the result is rounded to f16 right after,
// whose spacing at those values is 1e-3).  x is clamped to +-8 inside the polynomial (c < 0: the argument would turn
// round beyond |x| ~ 11; at 8 the sigmoid is 1 - 1e-12 / e^-27 already); the final product takes the unclamped x.
// Per value 3.5 packed instructions + v_exp_f32 + v_rcp_f32 against 8.5 + v_rcp_f32 above.
DLIMG_DEVICE float2_t gelu_pair(float2_t x) {
#if defined(DLIMG_TUNING) && defined(DLIMG_GELU_ERF)      // A/B of the two forms, tuning build only
    return gelu_pair_erf(x);
#endif
    constexpr float L2E = 1.44269504088896341f;
    const float2_t xc = {fminf(fmaxf(x[0], -8.0f), 8.0f), fminf(fmaxf(x[1], -8.0f), 8.0f)};
    const float2_t x2 = xc * xc;
    float2_t p = x2 * (0.0007030335806902641f * L2E) + (-0.07401129206536386f * L2E);       // -(c x^2 + b) log2(e)
    p = p * x2 + (-1.5950157685363155f * L2E);                                              // -(... + a) log2(e)
    const float2_t u = xc * p;                                                              // -x (a + b x^2 + c x^4) log2(e)
    const float2_t d = float2_t{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.0f;
    return x * float2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
DLIMG_DEVICE float4_t gelu4(float4_t v) {
#if defined(DLIMG_TUNING) && defined(DLIMG_NO_GELU)      // upper bound of what a cheaper GELU could return (WRONG results)
    return v;
#endif
    const float2_t lo = gelu_pair(float2_t{v[0], v[1]}), hi = gelu_pair(float2_t{v[2], v[3]});
    return float4_t{lo[0], lo[1], hi[0], hi[1]};
}
DLIMG_DEVICE float gelu_fast(float x) { return gelu_pair(float2_t{x, x})[0]; }

// value held by the partner lane in the other 32-lane half
DLIMG_DEVICE float swap_halves(float v) { return __shfl_xor(v, 32, 64); }

// Sum over each aligned group of 8 lanes, result in all 8, on the DPP path (no LDS round trips as with
// __shfl_xor): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7-i, the other quad).
template <int CTRL>
DLIMG_DEVICE float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
DLIMG_DEVICE float sum_over_8_lanes(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}

// Reductions over the 64 lanes of a wave, result in every lane.  All on the DPP path (VALU only): inside each row of
// 16 lanes by quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror; across the four rows by
// row_bcast:15 (rows 1 and 3 take lane 15 of the row before) and row_bcast:31 (rows 2 and 3 take lane 31); lane 63
// then holds the total and is broadcast through an SGPR.  A __shfl_xor butterfly costs six LDS round trips per value
// instead (measured: 18 sums per query made the decoder's token-to-image kernel 40 us long).
template <int CTRL, int ROW_MASK>
DLIMG_DEVICE float dpp_move_rows(float keep, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
DLIMG_DEVICE float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    v += dpp_move_rows<0x142, 0xa>(0.f, v);
    v += dpp_move_rows<0x143, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
DLIMG_DEVICE float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    v = fmaxf(v, dpp_move_rows<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_move_rows<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Bijective XCD-aware remap of a linear workgroup id (cdna_hip_programming.md §5 "XCD swizzle
// must be bijective"): hardware deals consecutive ids round-robin over the 8 XCDs; this gives
// every XCD one contiguous chunk of the logical tile range so neighbouring tiles share an L2.
DLIMG_DEVICE int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, k = bid / nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// 16-byte store of a RESULT: bytes that no workgroup of this launch reads again and that the next kernel reads from wherever
// it runs.  Write-through (`sc1`): the line does not stay behind in the XCD's L2 (MI355X_MICROARCH.md, "stores of each
// flavour") -- a kernel that writes 25-100 MB otherwise ends with up to 32 MB of dirty lines that the kernel boundary has to
// write back (MI355X_MICROARCH.md, price list, "boundary": + B / 6 TB/s) and that push the operand panels of its later
// workgroups out of the L2.  r06, measured alone on the chip: fc1 85.9 -> 82.0 us per four-image launch, qkv's fetch traffic
// 114 -> 103 MB.  Only for 16-byte accesses (narrower sc1 stores are one fabric write each, 2.7-12 x the time per byte).
// The store is an asm statement: the compiler's memory counter does not see it; nothing in a kernel waits for a result store.
template <typename V>
DLIMG_DEVICE void store16_result(void* p, V v) {
    static_assert(sizeof(V) == 16, "16-byte results only");
#if defined(DLIMG_TUNING) && defined(DLIMG_PLAIN_STORES)      // A/B: the plain store
    *reinterpret_cast<V*>(p) = v;
#else
    // (s_nop 1 inside the statement: a store of more than 8 bytes reads its data registers up to two states after it issues, and
    // the compiler pads that hazard only for its own stores -- cdna_hip_programming.md 5.7 item 1; without it the next
    // instruction may overwrite the data: fc1's epilogue did, the parity tests caught it)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(float4_t, v)) : "memory");
#endif
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte async global -> LDS copy: LDS destination = wave-uniform `lds_base` + lane*16.
DLIMG_DEVICE void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_base, 16, 0, 0);
}

}  // namespace dlimg

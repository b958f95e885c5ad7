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

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte async global -> LDS copy: LDS destination = wave-uniform `lds_base` + lane*16.
DLIMG_DEVICE void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_base, 16, 0, 0);
}

}  // namespace dlimg

// K17  longest-side resize: the stb_image_resize-equivalent resampler on the device
// (reference: dlimg::resize -> stbir_resize_uint8_generic, /root/reference/src/image.cpp:37-51).
// Two separable passes over host-built contributor tables (csrc/resize_tables.cpp):
//   horizontal: u8 --sRGB table--> linear float, weighted gather along x into fp32 rows of the output width
//   vertical  : weighted gather along y, float --Giesen table--> sRGB u8
// The same two passes with a linear decode table and no encode table are dlimg::resize_mask (image.cpp:53-62).
// Every multiply and add is explicitly rounded (no fma) and runs in increasing source order, so the
// result is bit-identical to oracle/stb_resize.py.  Edges clamp.
#include "device_common.hpp"
#include "kernels.hpp"

// bit-exactness contract with the oracle: no mul+add contraction anywhere in this file
#pragma clang fp contract(off)

namespace dlimg {
namespace {

__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, int w, int h, int stride, int C,
                                                       const int* __restrict__ first, const int* __restrict__ count,
                                                       const float* __restrict__ coef, int taps, int ow,
                                                       const float* __restrict__ decode, float* __restrict__ tmp) {
    __shared__ float lut[256];
    lut[threadIdx.x] = decode[threadIdx.x];
    __syncthreads();
    const long total = (long)h * ow;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int y = (int)(i / ow), ox = (int)(i % ow);
        const uint8_t* row = src + (size_t)y * stride;
        const int f = first[ox], n = count[ox];
        const float* cf = coef + (size_t)ox * taps;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < n; ++t) {
            const int j = min(max(f + t, 0), w - 1);
            const float wgt = cf[t];
            const uint8_t* px = row + (size_t)j * C;
            for (int c = 0; c < C; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(lut[px[c]], wgt));
        }
        float* dst = tmp + (size_t)i * C;
        for (int c = 0; c < C; ++c) dst[c] = acc[c];
    }
}

DLIMG_DEVICE uint8_t linear_to_srgb_uchar(float in, const uint32_t* tab4) {
    const float minval = __uint_as_float((127u - 13u) << 23);
    const float almost_one = __uint_as_float(0x3f7fffffu);
    if (!(in > minval)) in = minval;
    if (in > almost_one) in = almost_one;
    const uint32_t u = __float_as_uint(in);
    const uint32_t tab = tab4[(u - ((127u - 13u) << 23)) >> 20];
    const uint32_t bias = (tab >> 16) << 9;
    const uint32_t scale = tab & 0xffffu;
    const uint32_t t = (u >> 12) & 0xffu;
    return (uint8_t)((bias + scale * t) >> 16);
}

// STBIR_COLORSPACE_LINEAR encode: (int)(saturate(v) * 255 + 0.5)
DLIMG_DEVICE uint8_t linear_to_uchar(float v) {
    v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    return (uint8_t)(int)__fadd_rn(__fmul_rn(v, 255.0f), 0.5f);
}

template <bool SRGB>
__global__ __launch_bounds__(256) void resize_v_kernel(const float* __restrict__ tmp, int h, int ow, int C,
                                                       const int* __restrict__ first, const int* __restrict__ count,
                                                       const float* __restrict__ coef, int taps, int oh,
                                                       const uint32_t* __restrict__ encode, uint8_t* __restrict__ dst) {
    __shared__ uint32_t tab4[104];
    if (SRGB && threadIdx.x < 104) tab4[threadIdx.x] = encode[threadIdx.x];
    __syncthreads();
    const long total = (long)oh * ow;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / ow), ox = (int)(i % ow);
        const int f = first[oy], n = count[oy];
        const float* cf = coef + (size_t)oy * taps;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < n; ++t) {
            const int j = min(max(f + t, 0), h - 1);
            const float wgt = cf[t];
            const float* px = tmp + ((size_t)j * ow + ox) * C;
            for (int c = 0; c < C; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(px[c], wgt));
        }
        uint8_t* out = dst + (size_t)i * C;
        for (int c = 0; c < C; ++c) out[c] = SRGB ? linear_to_srgb_uchar(acc[c], tab4) : linear_to_uchar(acc[c]);
    }
}

}  // namespace

namespace k {

void resize_srgb(const uint8_t* src, int w, int h, int stride, int C, const ResizeAxis& ax, const ResizeAxis& ay,
                 const float* decode_lut, const uint32_t* encode_tab, float* tmp, uint8_t* dst, hipStream_t s) {
    if (w <= 0 || h <= 0 || C < 1 || C > 4 || stride < w * C) throw_error("resize_srgb: invalid source image");
    if (ax.out <= 0 || ay.out <= 0 || ax.taps <= 0 || ay.taps <= 0) throw_error("resize_srgb: invalid tables");
    const long n1 = (long)h * ax.out, n2 = (long)ay.out * ax.out;
    auto grid = [](long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); };
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid(n1)), dim3(256), 0, s, src, w, h, stride, C, ax.first, ax.count, ax.coef,
                       ax.taps, ax.out, decode_lut, tmp);
    if (encode_tab)
        hipLaunchKernelGGL(resize_v_kernel<true>, dim3(grid(n2)), dim3(256), 0, s, tmp, h, ax.out, C, ay.first, ay.count,
                           ay.coef, ay.taps, ay.out, encode_tab, dst);
    else        // linear colour space (resize_mask): decode_lut holds i / 255
        hipLaunchKernelGGL(resize_v_kernel<false>, dim3(grid(n2)), dim3(256), 0, s, tmp, h, ax.out, C, ay.first, ay.count,
                           ay.coef, ay.taps, ay.out, encode_tab, dst);
}

}  // namespace k
}  // namespace dlimg

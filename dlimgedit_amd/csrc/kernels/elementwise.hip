// HBM-bound kernels of the SAM path: pixel pre-processing, row LayerNorm, casts, im2col.
#include "device_common.hpp"
#include "kernels.hpp"

namespace dlimg {
namespace {

// ---------------------------------------------------------------------------------------------
// K1  pixel pre-processing.
//
// Restates, fused into one pass, what the reference does in three places:
//   create_image_tensor   /root/reference/src/segmentation.cpp:81-106  (channel map, u8 -> f32)
//   in-graph preprocessing of the encoder (export_models.py:26 use_preprocess=True):
//       (x - mean) / std, zero-pad bottom/right to 1024x1024, HWC -> CHW
//   the im2row of the 16x16/16 patch-embedding convolution
// Output is the A operand of the patch-embedding GEMM: [4096 patches, 768] f16,
// column = c*256 + iy*16 + ix.
//
// Mapping: one wave covers two horizontally adjacent patches; lane = iy*4 + p*2 + half reads
// 8 pixels (32 B of RGBA) so that 4 consecutive lanes read one full 128-byte line, and writes
// 16 B per channel so that the 32 lanes of one patch fill a contiguous 512-byte run per channel.

__constant__ float c_mean[3] = {123.675f, 116.28f, 103.53f};
__constant__ float c_std[3] = {58.395f, 57.12f, 57.375f};

struct ChannelMap { int bytes; int idx[3]; };

__host__ __device__ inline ChannelMap channel_map(int channels) {
    // dlimg::Channels: mask=1, rgb=3, rgba=4, bgra=5, argb=6  (segmentation.cpp:82-95)
    switch (channels) {
    case 1: return {1, {0, 0, 0}};
    case 3: return {3, {0, 1, 2}};
    case 5: return {4, {2, 1, 0}};
    case 6: return {4, {1, 2, 3}};
    default: return {4, {0, 1, 2}};
    }
}

// Up to 16 images per launch (blockIdx.y = image): the images of one batched pass share a launch, and at 16 images the
// kernel moves 168 MB, enough to be measured against the HBM rate instead of the launch floor (bench.py, hbm_kernels).
constexpr int PRE_MAX_JOBS = 16;
struct PreJob { const uint8_t* img; half_t* out; int w, h, stride, channels; };
struct PreJobs { PreJob j[PRE_MAX_JOBS]; };

__global__ __launch_bounds__(256) void preprocess_kernel(PreJobs jobs) {
    const PreJob job = jobs.j[blockIdx.y];
    const uint8_t* __restrict__ img = job.img;
    half_t* __restrict__ out = job.out;
    const int w = job.w, h = job.h, stride = job.stride, channels = job.channels;
    const int lane = lane_id();
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);       // patch pair index, 2048 per image
    const int iy = lane >> 2, p = (lane >> 1) & 1, half = lane & 1;
    const int patch = pair * 2 + p;
    const int py = patch >> 6, px = patch & 63;
    const int y = py * 16 + iy;
    const int x0 = px * 16 + half * 8;
    const ChannelMap cm = channel_map(channels);

    float v[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] = 0.f;

    if (y < h && x0 < w) {
        const uint8_t* row = img + (size_t)y * stride;
        const bool fast = cm.bytes == 4 && x0 + 8 <= w && ((((uintptr_t)row) + (size_t)x0 * 4) & 15) == 0;
        if (fast) {
            const uint4* src = reinterpret_cast<const uint4*>(row + (size_t)x0 * 4);
            uint4 q0 = src[0], q1 = src[1];
            uint32_t px32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float u = (float)((px32[i] >> (8 * cm.idx[c])) & 0xffu);
                    v[c][i] = (u - c_mean[c]) / c_std[c];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (x0 + i < w) {
                    const uint8_t* px8 = row + (size_t)(x0 + i) * cm.bytes;
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c][i] = ((float)px8[cm.idx[c]] - c_mean[c]) / c_std[c];
                }
            }
        }
    }
    half_t* dst = out + (size_t)patch * 768 + iy * 16 + half * 8;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        half8_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (half_t)v[c][i];
        store16_result(dst + c * 256, o);
    }
}

// ---------------------------------------------------------------------------------------------
// Row LayerNorm: one wave per row, values held in registers, two-pass statistics in fp32.
constexpr int LN_MAX_PER_LANE = 20;   // D <= 1280

template <int ACT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, int rows, int D,
                                                        float* out_f32, half_t* out_h) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = lane_id();
    const float* xr = x + (size_t)row * D;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        int c = i * 64 + lane;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        int c = i * 64 + lane;
        float d = c < D ? v[i] - mean : 0.f;
        v[i] = d;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        int c = i * 64 + lane;
        if (c < D) {
            float y = v[i] * rstd * w[c] + b[c];
            if (ACT == k::ACT_GELU) y = gelu_erf(y);
            if (out_f32) out_f32[(size_t)row * D + c] = y;
            if (out_h) out_h[(size_t)row * D + c] = (half_t)y;
        }
    }
}

// Same, rows whose length is a multiple of 256: 16-byte loads/stores (the encoder widths 768/1024/1280
// and the 256-channel neck / decoder rows).
constexpr int LN_MAX_VEC = 5;         // D <= 1280
template <int ACT>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, int rows, int D,
                                                            float* out_f32, half_t* out_h, int* nonfinite) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = lane_id();
    const int nv = D >> 8;
    const float4_t* xr = reinterpret_cast<const float4_t*>(x + (size_t)row * D);
    float4_t v[LN_MAX_VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        v[i] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (i < nv) v[i] = xr[i * 64 + lane];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        if (i < nv) {
            v[i] -= mean;
            q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    // a row that holds an infinity or a NaN has no finite statistics: reported (one store per such row into host-visible
    // memory) where the caller asked for it -- SamModel::encode on the last LayerNorm of the neck, whose input has seen
    // every activation of the image
    if (nonfinite && lane == 0 && !(fabsf(mean) < INFINITY && fabsf(rstd) < INFINITY)) *nonfinite = 1;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        if (i < nv) {
            const int c4 = i * 64 + lane;
            const float4_t ww = reinterpret_cast<const float4_t*>(w)[c4], bb = reinterpret_cast<const float4_t*>(b)[c4];
            float4_t y = v[i] * rstd * ww + bb;
            if (ACT == k::ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = gelu_erf(y[e]);
            }
            if (out_f32) store16_result(out_f32 + (size_t)row * D + (size_t)c4 * 4, y);
            if (out_h) {
                half4_t h = {(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
                reinterpret_cast<half4_t*>(out_h + (size_t)row * D)[c4] = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       size_t b_mod, size_t n4, float* out_f32, half_t* out_h) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4_t v = reinterpret_cast<const float4_t*>(a)[i];
        if (b) {
            float4_t u = *reinterpret_cast<const float4_t*>(b + (i * 4) % b_mod);
            v += u;
        }
        if (out_f32) store16_result(out_f32 + i * 4, v);
        if (out_h) {
            half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            reinterpret_cast<half4_t*>(out_h)[i] = o;
        }
    }
}

__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (half_t)in[i];
}

// im2col for the 3x3 neck convolution: 16-byte chunks, one per thread.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const half_t* __restrict__ in, int B, int C,
                                                        half_t* __restrict__ out) {
    const int cpr = C / 8;                               // chunks per (token, tap)
    const size_t total = (size_t)B * 4096 * 9 * cpr;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int ch = (int)(i % cpr);
        size_t t = i / cpr;
        int tap = (int)(t % 9);
        size_t tok = t / 9;
        int b = (int)(tok >> 12), y = (int)((tok >> 6) & 63), x = (int)(tok & 63);
        int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        half8_t v = zero_h8();
        if (yy >= 0 && yy < 64 && xx >= 0 && xx < 64)
            v = *reinterpret_cast<const half8_t*>(in + (((size_t)b * 4096 + yy * 64 + xx) * C) + ch * 8);
        store16_result(out + (tok * 9 + tap) * C + ch * 8, v);
    }
}

inline int grid_for(size_t work_items, int cap = 2048) {
    size_t g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace

namespace k {

void preprocess_batch(const PreImage* images, int count, hipStream_t s) {
    for (int base = 0; base < count; base += PRE_MAX_JOBS) {
        const int n = count - base < PRE_MAX_JOBS ? count - base : PRE_MAX_JOBS;
        PreJobs jobs{};
        for (int i = 0; i < n; ++i) {
            const PreImage& im = images[base + i];
            if (!im.img || !im.patches) throw_error("preprocess: null buffer");
            if (im.w <= 0 || im.h <= 0 || im.w > 1024 || im.h > 1024) throw_error("preprocess: image must be 1..1024 pixels per side");
            const int bytes = im.channels > 4 ? 4 : im.channels;
            if (!(im.channels == 1 || im.channels == 3 || im.channels == 4 || im.channels == 5 || im.channels == 6))
                throw_error("preprocess: unsupported channel order");
            if (im.stride < im.w * bytes) throw_error("preprocess: stride smaller than one row of pixels");
            jobs.j[i] = PreJob{im.img, im.patches, im.w, im.h, im.stride, im.channels};
        }
        hipLaunchKernelGGL(preprocess_kernel, dim3(512, n), dim3(256), 0, s, jobs);
    }
}

void preprocess(const uint8_t* img, int w, int h, int stride, int channels, half_t* patches, hipStream_t s) {
    const PreImage one{img, w, h, stride, channels, patches};
    preprocess_batch(&one, 1, s);
}

void layernorm(const float* x, const float* w, const float* b, float eps, int rows, int D, int act, float* out_f32,
               half_t* out_h, hipStream_t s, int* nonfinite) {
    if (rows <= 0) return;
    if (D <= 0 || D > LN_MAX_PER_LANE * 64) throw_error("layernorm: row length must be in 1..1280");
    dim3 grid((rows + 3) / 4);
    const bool aligned = !(((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)out_f32) & 15) && !((uintptr_t)out_h & 7);
    if (D % 256 == 0 && aligned) {
        if (act == ACT_GELU)
            hipLaunchKernelGGL(layernorm_vec_kernel<ACT_GELU>, grid, dim3(256), 0, s, x, w, b, eps, rows, D, out_f32, out_h, nonfinite);
        else
            hipLaunchKernelGGL(layernorm_vec_kernel<ACT_NONE>, grid, dim3(256), 0, s, x, w, b, eps, rows, D, out_f32, out_h, nonfinite);
        return;
    }
    if (nonfinite) throw_error("layernorm: the non-finite report needs rows that are a multiple of 256 long");
    if (act == ACT_GELU)
        hipLaunchKernelGGL(layernorm_kernel<ACT_GELU>, grid, dim3(256), 0, s, x, w, b, eps, rows, D, out_f32, out_h);
    else
        hipLaunchKernelGGL(layernorm_kernel<ACT_NONE>, grid, dim3(256), 0, s, x, w, b, eps, rows, D, out_f32, out_h);
}

void add_cast(const float* a, const float* b, size_t b_mod, size_t n, float* out_f32, half_t* out_h, hipStream_t s) {
    if (n == 0) return;
    if (n % 4 || (b && (b_mod % 4 || b_mod == 0))) throw_error("add_cast: lengths must be multiples of 4");
    hipLaunchKernelGGL(add_cast_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, a, b, b_mod, n / 4, out_f32, out_h);
}

void cast_f16(const float* in, half_t* out, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(cast_f16_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
}

void im2col3x3(const half_t* in, int B, int C, half_t* out, hipStream_t s) {
    if (B <= 0 || C <= 0 || C % 8) throw_error("im2col3x3: channel count must be a positive multiple of 8");
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((size_t)B * 4096 * 9 * (C / 8))), dim3(256), 0, s, in, B, C, out);
}

}  // namespace k
}  // namespace dlimg

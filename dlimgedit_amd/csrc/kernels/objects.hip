// Pre- and post-processing around the BiRefNet call of dlimgedit's segment_objects.
// The network is an ONNX graph in the reference and out of scope; what the library computes itself is here:
//   prepare_image   u8 HWC -> f32 NCHW, (x / 255 - mean) / std        /root/reference/src/segmentation.cpp:244-256
//   process_mask    u8 = uint8_t(sigmoid(x) * 255.f)                    /root/reference/src/segmentation.cpp:258-270
// (the third step, resize_mask, is the box-filter mode of kernels/resize.hip).
// Both are pure HBM streams: one pixel per thread, planar 4-byte stores / 1-byte stores coalesced over the wave.
// Float arithmetic in the reference's order with explicit rounding, so the results are bit-identical to
// oracle/birefnet_oracle.py; exp is evaluated in double and rounded once (= the correctly rounded float exp).
#include "device_common.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace dlimg {
namespace {

__global__ __launch_bounds__(256) void prepare_image_kernel(const uint8_t* __restrict__ px, int w, int h, int stride,
                                                            int bytes_pp, float3_t mean, float3_t std,
                                                            float* __restrict__ out) {
    const long total = (long)w * h;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i % w);
        const uint8_t* p = px + (size_t)y * stride + (size_t)x * bytes_pp;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float value = __fdiv_rn((float)p[c], 255.0f);
            out[(size_t)c * total + i] = __fdiv_rn(__fsub_rn(value, mean[c]), std[c]);
        }
    }
}

__global__ __launch_bounds__(256) void process_mask_kernel(const float* __restrict__ logits, long n,
                                                           uint8_t* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float e = (float)exp(-(double)logits[i]);
        const float s = __fdiv_rn(1.0f, __fadd_rn(1.0f, e));
        out[i] = (uint8_t)(int)__fmul_rn(s, 255.0f);
    }
}

unsigned grid_for(long n) {
    const long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

namespace k {

void birefnet_prepare_image(const uint8_t* pixels, int w, int h, int stride, int bytes_pp, const float mean[3],
                            const float std[3], float* out, hipStream_t s) {
    if (w <= 0 || h <= 0 || bytes_pp < 3 || bytes_pp > 4 || stride < w * bytes_pp)
        throw_error("prepare_image: needs an image with at least three channels");
    hipLaunchKernelGGL(prepare_image_kernel, dim3(grid_for((long)w * h)), dim3(256), 0, s, pixels, w, h, stride, bytes_pp,
                       float3_t{mean[0], mean[1], mean[2]}, float3_t{std[0], std[1], std[2]}, out);
}

void birefnet_process_mask(const float* logits, int w, int h, uint8_t* out, hipStream_t s) {
    if (w <= 0 || h <= 0) throw_error("process_mask: empty mask");
    hipLaunchKernelGGL(process_mask_kernel, dim3(grid_for((long)w * h)), dim3(256), 0, s, logits, (long)w * h, out);
}

}  // namespace k
}  // namespace dlimg

// Targeted probes of the three "silently wrong numbers" traps of DESIGN.md section 6 (r03, met in the attention kernels),
// each as a pair: the form that is wrong and the form the kernels use.  Standalone:
//   hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o asm_hazards tools/asm_hazards.cpp && ./asm_hazards
//   1. an asm VALU instruction reading the result of a v_exp_f32 in the next issue slot (transcendental -> VALU forwarding:
//      the compiler pads its own instructions, nobody pads inside or in front of an asm string)
//        1a  exp and consumer in ONE asm statement, back to back         1b  the same with s_nop 0 between them
//        1c  compiler-emitted exp (__builtin_amdgcn_exp2f), asm consumer  1d  all plain C++ (what the kernels do)
//   2. v_permlane32_swap_b32 on a register written by the instruction before it (two wait states after a vector write)
//        2a  v_mov + swap back to back in one asm statement               2b  with s_nop 1 in front and behind (the kernels' form)
//   3. fmax over the two results of __builtin_amdgcn_permlane32_swap(v, v): folded by hipcc 7.2 to the first result
//        3a  the builtin form                                             3b  the asm form of 2b
// Every case: 256 x 256 threads x 4000 iterations on random inputs, compared with values computed the slow way
// (__shfl for the other half's value, plain exp2f + add).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__device__ __forceinline__ float rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(s >> 8) * (1.0f / 16777216.0f) * 8.0f - 4.0f;
}

template <int CASE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
    unsigned wrong = 0;
    for (int it = 0; it < iters; ++it) {
        const float x = rnd(s), y = rnd(s);
        float got = 0.f, want = 0.f;
        if (CASE < 10) {                     // exp2(x) + y
            want = __builtin_amdgcn_exp2f(x) + y;
            if (CASE == 0) asm volatile("v_exp_f32 %0, %1\n\tv_add_f32 %0, %0, %2" : "=&v"(got) : "v"(x), "v"(y));
            if (CASE == 1) asm volatile("v_exp_f32 %0, %1\n\ts_nop 0\n\tv_add_f32 %0, %0, %2" : "=&v"(got) : "v"(x), "v"(y));
            if (CASE == 2) {
                const float e = __builtin_amdgcn_exp2f(x);
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(got) : "v"(e), "v"(y));
            }
            if (CASE == 3) got = __builtin_amdgcn_exp2f(x) + y;
        } else {                             // max(x of this lane, x of the lane 32 away)
            want = fmaxf(x, __shfl_xor(x, 32, 64));
            if (CASE == 10) {
                float ta, tb;
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\tv_permlane32_swap_b32 %0, %1" : "=&v"(ta), "=&v"(tb) : "v"(x));
                got = fmaxf(ta, tb);
            }
            if (CASE == 11 || CASE == 21) {
                float ta = x, tb = x;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ta), "+v"(tb));
                got = fmaxf(ta, tb);
            }
            if (CASE == 20) {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const unsigned bits = __float_as_uint(x);
                const v2u r = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
                got = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            }
        }
        wrong += __float_as_uint(got) != __float_as_uint(want);
    }
    if (wrong) atomicAdd(out, (unsigned long long)wrong);
}

template <int CASE>
void run(const char* what, unsigned long long* dev) {
    CHECK(hipMemset(dev, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(probe<CASE>, dim3(256), dim3(256), 0, 0, dev, 4000);
    CHECK(hipDeviceSynchronize());
    unsigned long long wrong = 0;
    CHECK(hipMemcpy(&wrong, dev, sizeof(wrong), hipMemcpyDeviceToHost));
    std::printf("%-92s wrong %10llu of %llu\n", what, wrong, 256ull * 256ull * 4000ull);
}

int main() {
    unsigned long long* dev;
    CHECK(hipMalloc(&dev, sizeof(unsigned long long)));
    run<0>("1a v_exp_f32 ; v_add_f32 on its result, one asm statement, back to back", dev);
    run<1>("1b the same with s_nop 0 between them", dev);
    run<2>("1c compiler-emitted v_exp_f32, asm v_add_f32 consumer", dev);
    run<3>("1d plain C++ (the kernels' form)", dev);
    run<10>("2a v_mov_b32 x2 ; v_permlane32_swap_b32, one asm statement, back to back", dev);
    run<11>("2b s_nop 1 ; v_permlane32_swap_b32 ; s_nop 1 as one asm statement (the kernels' form)", dev);
    run<20>("3a fmax over __builtin_amdgcn_permlane32_swap(v, v)", dev);
    run<21>("3b fmax over the asm form of 2b", dev);
    return 0;
}

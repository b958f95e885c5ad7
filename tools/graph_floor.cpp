// What a dependent launch costs on a stream and inside a hipGraph: N tiny kernels, each reading what the previous wrote.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/graph_floor tools/graph_floor.cpp && tools/_bin/graph_floor [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void step(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.0001f + 1.0f;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 20, n = 64 * 256, reps = 300;
    float *a, *b;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4));
    CHECK(hipMemset(a, 0, n * 4)); CHECK(hipMemset(b, 0, n * 4));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto chain = [&](hipStream_t st) {
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step, dim3(64), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n);
    };
    for (int w = 0; w < 10; ++w) chain(s);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) chain(s);
    CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("stream: %d dependent launches, %.2f us per launch\n", N, 1e3 * ms / (reps * N));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain(s);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 10; ++w) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("graph : %d dependent nodes,    %.2f us per node\n", N, 1e3 * ms / (reps * N));
    return 0;
}

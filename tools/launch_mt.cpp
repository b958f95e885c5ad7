// Host cost of kernel launches from several threads, each on its own non-blocking stream (what the execution lanes and
// multi-threaded callers of the drop-in table do):
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/launch_mt tools/launch_mt.cpp -lpthread
//   tools/_bin/launch_mt                                      # the system runtime (/opt/rocm)
//   LD_LIBRARY_PATH=<torch>/lib tools/_bin/launch_mt          # the runtime PyTorch bundles (what bench.py runs on)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void tiny(int* p) { if (p && threadIdx.x == 1234567) *p = 1; }

int main() {
    int ver = 0;
    (void)hipRuntimeGetVersion(&ver);
    printf("HIP runtime version %d\n", ver);
    for (int chain : {1, 20}) {
        for (int threads : {1, 2, 4, 8}) {
            std::vector<std::thread> ts;
            std::vector<double> rate(threads);
            for (int t = 0; t < threads; ++t)
                ts.emplace_back([&, t] {
                    hipStream_t s;
                    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                    hipEvent_t e;
                    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
                    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, nullptr);
                    (void)hipStreamSynchronize(s);
                    const int calls = 2000 / chain;
                    auto t0 = std::chrono::steady_clock::now();
                    for (int c = 0; c < calls; ++c) {
                        for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, nullptr);
                        (void)hipEventRecord(e, s);
                        (void)hipEventSynchronize(e);              // a caller waits for its own chain, as compute_mask does
                    }
                    rate[t] = calls / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    (void)hipStreamDestroy(s);
                });
            for (auto& th : ts) th.join();
            double sum = 0;
            for (double r : rate) sum += r;
            printf("chain of %2d launches + event wait: %d thread(s): %9.0f chains/s in total (%.1f us per chain and thread)\n", chain, threads, sum,
                   1e6 * threads / sum);
        }
    }
    return 0;
}

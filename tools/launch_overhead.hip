// Micro-benchmark: what does one dependent kernel launch cost on this box?
//   eager vs hipGraph, small vs ~640-byte by-value kernarg, trivial vs one-dependent-load body,
//   linear chain vs a chain with one forked side branch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { long long a[80]; int *p; };
__global__ void k_small(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += (int)b.a[3]; }
__global__ void k_empty(int *p) {}
__global__ void k_wide(int *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> double timeit(hipStream_t s, int reps, F f) {
    f(); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main() {
    hipStream_t s, side; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ef, ej; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    int *p; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24));
    Big b{}; b.p = p;
    const int N = 10;
    auto chain = [&](int kind) {
        for (int i = 0; i < N; ++i) {
            if (kind == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p);
            if (kind == 1) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b);
            if (kind == 2) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p);
            if (kind == 3) hipLaunchKernelGGL(k_wide, dim3(1024), dim3(256), 0, s, p, 1 << 18);
            if (kind == 4) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, b);
        }
    };
    auto forked = [&]() {
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p);
            if (i == 3) { hipEventRecord(ef, s); hipStreamWaitEvent(side, ef, 0); hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, side, p + 64); hipEventRecord(ej, side); }
        }
        hipStreamWaitEvent(s, ej, 0);
    };
    const char *names[] = {"small-arg 1 block", "640B-arg 1 block", "empty 256 blocks", "wide 1024 blocks (1MB rmw)", "640B-arg 256 blocks"};
    for (int kind = 0; kind < 5; ++kind) {
        double eager = timeit(s, 200, [&] { chain(kind); });
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(kind); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double graph = timeit(s, 200, [&] { hipGraphLaunch(ge, s); });
        printf("%-28s eager %6.2f us/kernel   graph %6.2f us/kernel\n", names[kind], eager / N, graph / N);
    }
    {
        double eager = timeit(s, 200, forked);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); forked(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double graph = timeit(s, 200, [&] { hipGraphLaunch(ge, s); });
        printf("%-28s eager %6.2f us/chain    graph %6.2f us/chain  (10 small kernels + 1 forked)\n", "fork/join", eager, graph);
    }
    return 0;
}

// Micro-benchmark: where does the time between two dependent kernels of a hipGraph chain go?
// Every block stamps the device wall clock (100 MHz) when it starts and when it ends; the gap
// between the last block of kernel i and the first block of kernel i+1 is the price of the
// launch boundary as the device sees it.  Variants isolate kernarg size, dynamic LDS, grid size
// and the amount of data a kernel reads / leaves dirty.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
struct Big { long long a[60]; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
extern __shared__ unsigned char dyn[];
__device__ __forceinline__ void body(u64 *stamps, int kidx, int maxblk, float *data, size_t n_read, size_t n_write, long long extra) {
    u64 *st = stamps + ((size_t)kidx * maxblk + blockIdx.x) * 2;
    if (threadIdx.x == 0) st[0] = wall_clock64();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    float acc = (float)extra;
    for (size_t i = tid; i < n_read; i += nth) acc += data[i];
    for (size_t i = tid; i < n_write; i += nth) data[i + (1 << 24)] = acc;
    if (acc == 123.456f) data[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) st[1] = wall_clock64();
}
__global__ void k_small(u64 *stamps, int kidx, int maxblk, float *data, size_t n_read, size_t n_write) { body(stamps, kidx, maxblk, data, n_read, n_write, 0); }
__global__ void k_big(u64 *stamps, int kidx, int maxblk, float *data, size_t n_read, size_t n_write, Big b) {
    long long s = 0;
    for (int i = 0; i < 60; ++i) s += b.a[i];
    body(stamps, kidx, maxblk, data, n_read, n_write, s);
}
__global__ void k_ptr(u64 *stamps, int kidx, int maxblk, float *data, size_t n_read, size_t n_write, const Big *b) {
    long long s = 0;
    for (int i = 0; i < 60; ++i) s += b->a[i];
    body(stamps, kidx, maxblk, data, n_read, n_write, s);
}
__global__ void k_scratch(u64 *stamps, int kidx, int maxblk, float *data, size_t n_read, size_t n_write) {
    volatile int spill[12];                        // private (scratch) memory: what a register-capped kernel gets when it spills
    for (int i = 0; i < 12; ++i) spill[i] = (int)threadIdx.x + i;
    long long s = 0;
    for (int i = 0; i < 12; ++i) s += spill[(i * 5 + kidx) % 12];
    body(stamps, kidx, maxblk, data, n_read, n_write, s);
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 20, MAXBLK = 4096;
    u64 *stamps; CK(hipMalloc(&stamps, (size_t)N * MAXBLK * 16));
    float *data; CK(hipMalloc(&data, (size_t)1 << 28)); CK(hipMemset(data, 0, (size_t)1 << 28));
    Big big{};
    Big *dbig; CK(hipMalloc(&dbig, sizeof(Big))); CK(hipMemset(dbig, 0, sizeof(Big)));
    struct V { const char *name; int blocks, lds; int bigarg; size_t rd, wr; };
    const V vs[] = {
        {"256 blocks, 8-byte args", 256, 0, 0, 0, 0},
        {"  + 480-byte by-value arg", 256, 0, 1, 0, 0},
        {"  the same 480 bytes behind a pointer", 256, 0, 2, 0, 0},
        {"  + 20 KB dynamic LDS", 256, 20480, 0, 0, 0},
        {"  + 48 bytes of scratch per lane", 256, 0, 3, 0, 0},
        {"3000 blocks, scratch", 3000, 0, 3, 0, 0},
        {"3000 blocks", 3000, 0, 0, 0, 0},
        {"256 blocks, reads 8 MB", 256, 0, 0, 2 << 20, 0},
        {"256 blocks, writes 1 MB", 256, 0, 0, 0, 1 << 18},
        {"256 blocks, writes 8 MB", 256, 0, 0, 0, 2 << 20},
        {"1024 blocks, writes 32 MB", 1024, 0, 0, 0, 8 << 20},
        {"3000 blocks, all of it, w 8 MB", 3000, 20480, 1, 2 << 20, 2 << 20},
    };
    printf("%-34s %10s %10s %10s %10s\n", "variant", "step us", "in-kernel", "gap", "start skew");
    for (const V &v : vs) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            if (v.bigarg == 3) hipLaunchKernelGGL(k_scratch, dim3(v.blocks), dim3(256), v.lds, s, stamps, i, MAXBLK, data, v.rd, v.wr);
            else if (v.bigarg == 2) hipLaunchKernelGGL(k_ptr, dim3(v.blocks), dim3(256), v.lds, s, stamps, i, MAXBLK, data, v.rd, v.wr, (const Big *)dbig);
            else if (v.bigarg) hipLaunchKernelGGL(k_big, dim3(v.blocks), dim3(256), v.lds, s, stamps, i, MAXBLK, data, v.rd, v.wr, big);
            else hipLaunchKernelGGL(k_small, dim3(v.blocks), dim3(256), v.lds, s, stamps, i, MAXBLK, data, v.rd, v.wr);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<u64> h((size_t)N * MAXBLK * 2);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double span = 0, gap = 0, skew = 0, step = 0;
        u64 prev_end = 0, prev_start = 0;
        for (int i = 0; i < N; ++i) {
            u64 first = ~0ull, last_start = 0, last = 0;
            for (int b = 0; b < v.blocks; ++b) {
                const u64 a = h[((size_t)i * MAXBLK + b) * 2], e = h[((size_t)i * MAXBLK + b) * 2 + 1];
                first = std::min(first, a); last_start = std::max(last_start, a); last = std::max(last, e);
            }
            span += (double)(last - first);
            skew += (double)(last_start - first);
            if (i) { gap += (double)(first - prev_end); step += (double)(first - prev_start); }
            prev_end = last; prev_start = first;
        }
        printf("%-34s %10.2f %10.2f %10.2f %10.2f\n", v.name, step / (N - 1) / 100, span / N / 100, gap / (N - 1) / 100, skew / N / 100);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}

// Micro-benchmark: what does an in-kernel fan-in cost on this GPU -- P producer blocks finish a piece of work, C consumer blocks of
// the SAME launch may start theirs only then -- against the 2 us of a launch boundary?  (the two-launch
// step needs one, activation -> middle role.)  Producers: a delay standing for their work, 256 bytes each written through to
// memory, the stores' acknowledgement awaited, one atomic on one of NC counters.  Consumers: wait `look` us before the first look
// (a look that comes too early costs a round trip and traffic), then wave 0 polls the counters (agent-scope loads) until their
// sum is P; the block then reads every producer's bytes and checks them.  Optionally S more blocks stream memory beside them.
// Every block stamps the device wall clock (100 MHz).
//     hipcc --offload-arch=gfx950 -O3 -o tools/fanin tools/fanin.hip && tools/fanin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void sleep_us(float us) { const u64 t0 = wall_clock64(); while ((float)(wall_clock64() - t0) < us * 100.f) __builtin_amdgcn_s_sleep(8); }

__global__ __launch_bounds__(256) void k_fanin(u64 *stamps, uint32_t *data, uint32_t *counters, float *stream, int P, int C, int NC,
                                               float work_us, float look_us, uint32_t epoch, uint32_t target, size_t stream_words) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ uint32_t s_go;
    if (tid == 0) stamps[b * 4 + 0] = wall_clock64();
    if (b < P) {                                       // ---- producer
        sleep_us(work_us + 0.002f * (float)(b % 97));                                  // (not all at the same instant)
        if (tid < 64) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(data + (size_t)b * 64 + tid), "v"(epoch + (uint32_t)tid) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            stamps[b * 4 + 1] = wall_clock64();                                       // (work done, bytes acknowledged)
            atomicAdd(&counters[(b % NC) * 32], 1u);
            stamps[b * 4 + 2] = wall_clock64();
        }
        return;
    }
    if (b < P + C) {                                   // ---- consumer
        sleep_us(look_us);
        if (tid < 64) {
            int looks = 0;
            for (;; ++looks) {
                uint32_t v = tid < NC ? __hip_atomic_load(&counters[tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (v >= target || looks > (1 << 18)) break;            // (every wave leaves: a quarter of a million looks is seconds)
                __builtin_amdgcn_s_sleep(2);
            }
            if (tid == 0) { stamps[b * 4 + 1] = wall_clock64(); stamps[b * 4 + 3] = (u64)looks; s_go = 1; }
        }
        __syncthreads();
        uint32_t bad = 0;
        for (int i = tid; i < P * 64; i += 256) bad |= __hip_atomic_load(&data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch + (uint32_t)(i & 63);
        if (__syncthreads_or((int)bad) && tid == 0) stamps[b * 4 + 3] |= 1ull << 32;   // (a producer's bytes were not there)
        if (tid == 0) stamps[b * 4 + 2] = wall_clock64();
        return;
    }
    // ---- bystanders: stream
    float acc = 0.f;
    const size_t nth = (size_t)(gridDim.x - P - C) * 256, t0 = (size_t)(b - P - C) * 256 + tid;
    for (size_t i = t0; i < stream_words; i += nth) acc += stream[i];
    if (acc == 1.2345f) stream[0] = acc;
    if (tid == 0) stamps[b * 4 + 2] = wall_clock64();
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int MAXB = 4096;
    u64 *stamps; CK(hipMalloc(&stamps, (size_t)MAXB * 32));
    uint32_t *data, *counters; CK(hipMalloc(&data, (size_t)MAXB * 256)); CK(hipMalloc(&counters, 64 * 128));
    float *stream; const size_t SW = (size_t)8 << 20; CK(hipMalloc(&stream, SW * 4)); CK(hipMemset(stream, 0, SW * 4));
    struct V { const char *name; int P, C, S, NC; float work, look; };
    const V vs[] = {
        {"164 -> 385, 1 counter, look at once", 164, 385, 0, 1, 3.0f, 0.f},
        {"164 -> 385, 16 counters, look at once", 164, 385, 0, 16, 3.0f, 0.f},
        {"164 -> 385, 16 counters, first look at 3.5 us", 164, 385, 0, 16, 3.0f, 3.5f},
        {"164 -> 385, 16 counters, first look at 4.5 us", 164, 385, 0, 16, 3.0f, 4.5f},
        {"164 -> 385, 16 counters, look at 4.5 us, 1 300 blocks streaming 32 MB beside", 164, 385, 1300, 16, 3.0f, 4.5f},
        {"164 -> 385, 16 counters, look at once, 1 300 blocks streaming 32 MB beside", 164, 385, 1300, 16, 3.0f, 0.f},
        {"32 -> 32, 16 counters, first look at 3.5 us", 32, 32, 0, 16, 3.0f, 3.5f},
    };
    std::vector<u64> h((size_t)MAXB * 4);
    uint32_t epoch = 0;
    for (const V &v : vs) {
        CK(hipMemsetAsync(counters, 0, 64 * 128, s));      // (on the kernels' stream: it is a non-blocking one)
        CK(hipStreamSynchronize(s));
        uint32_t target = 0;
        double seen = 0, seen_max = 0, done = 0, looks = 0, prod_atomic = 0;
        int bad = 0;
        const int R = 20;
        for (int rep = 0; rep < R + 3; ++rep) {
            target += (uint32_t)v.P;                  // (the counters are not reset between the launches of a variant)
            hipLaunchKernelGGL(k_fanin, dim3(v.P + v.C + v.S), dim3(256), 0, s, stamps, data, counters, stream, v.P, v.C, v.NC, v.work, v.look, epoch, target, SW);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h.data(), stamps, (size_t)(v.P + v.C + v.S) * 32, hipMemcpyDeviceToHost));
            ++epoch;
            if (rep < 3) continue;
            u64 last_prod = 0, first = ~0ull;
            for (int b = 0; b < v.P + v.C; ++b) first = std::min(first, h[b * 4]);
            for (int b = 0; b < v.P; ++b) { last_prod = std::max(last_prod, h[b * 4 + 1]); prod_atomic += (double)(h[b * 4 + 2] - h[b * 4 + 1]) / 100.0 / v.P; }
            std::vector<double> sv;
            double dmax = 0;
            for (int b = v.P; b < v.P + v.C; ++b) {
                sv.push_back(((double)h[b * 4 + 1] - (double)last_prod) / 100.0);
                dmax = std::max(dmax, ((double)h[b * 4 + 2] - (double)last_prod) / 100.0);
                looks += (double)(h[b * 4 + 3] & 0xFFFFFFFFu) / v.C;
                bad += (int)(h[b * 4 + 3] >> 32);
            }
            std::sort(sv.begin(), sv.end());
            seen += sv[sv.size() / 2]; seen_max += sv.back(); done += dmax;
            (void)first;
        }
        printf("%-84s producers' bytes out -> consumers see the count: median %.2f us, last %.2f us; all consumers have read the bytes %.2f us after the last producer; "
               "%.1f failed looks per consumer; a producer's atomic returns in %.2f us%s\n", v.name, seen / R, seen_max / R, done / R, looks / R, prod_atomic / R, bad ? "  ** STALE BYTES SEEN **" : "");
    }
    return 0;
}

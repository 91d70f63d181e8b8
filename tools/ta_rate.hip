// Micro-benchmark: what does a vector-memory instruction cost a CU's address/texture path when its data is
// already near (L1 / L2 hits)?  The segment scan issues ~40 of them per wave and batch; if they are not free,
// six resident blocks per CU queue behind each other there.
//   every wave of a full-chip grid (6 blocks of 256 threads per CU) runs ITER rounds of 8 independent loads and
//   one wait; reported: nanoseconds and clocks (2.4 GHz) per wave-instruction and CU
//   kinds: 0 dword, all lanes the same address        1 dword, lanes consecutive (256 B per wave)
//          2 dword, ~2 % of the lanes a random line of a 256 KB table, the others one shared word (the scan's lookup)
//          3 as 2, but only the ~2 % lanes active (exec-masked)
//          4 dwordx4, lanes consecutive (1 KB per wave), 64 KB working set per block (L2 hits)
//          5 the same as non-temporal loads (do streaming loads take another path?)
//          6..10 the same through buffer loads with cache-policy bits: none / sc0 / sc1 / sc0 sc1 / nt
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 64;

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int KIND>
__global__ __launch_bounds__(256) void k_ta(const unsigned *__restrict__ table, unsigned *out) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < ITER; ++it) {
        unsigned v[8];
        uint4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned h = hash32(tid * 977u + it * 8u + j);
            const bool on = (h & 63u) == 0;                       // ~1.6 % of the lanes
            const unsigned far = (h >> 8) & 0xFFFFu;              // a word of the 256 KB table
            if (KIND == 0) v[j] = table[(it * 8 + j) & 1023];
            if (KIND == 1) v[j] = table[((it * 8 + j) * 64 + (threadIdx.x & 63)) & 0xFFFF];
            if (KIND == 2) v[j] = table[on ? far : 0];
            if (KIND == 3) { v[j] = 0; if (on) v[j] = table[far]; }
            const unsigned *src = table + ((((blockIdx.x * 8 + j) * 256 + threadIdx.x) * 4 + it * 64) & 0xFFFC);
            if (KIND == 4) w[j] = *(const uint4 *)src;
            if (KIND >= 6) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, 1 << 18, 0x00020000);
                const int aux = KIND == 6 ? 0 : KIND == 7 ? 1 : KIND == 8 ? 16 : KIND == 9 ? 17 : 2;
                const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((const char *)src - (const char *)table), 0, aux);
                w[j] = make_uint4(t.x, t.y, t.z, t.w);
            }
            if (KIND == 5) {                                      // compiler-managed streaming load (nt bit)
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_nontemporal_load((const v4u *)src);
                w[j] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += KIND >= 4 ? w[j].x ^ w[j].y ^ w[j].z ^ w[j].w : v[j];
    }
    if (acc == 0x12345u) out[tid] = acc;
}

int main() {
    unsigned *table, *out;
    CK(hipMalloc(&table, 1 << 18)); CK(hipMemset(table, 0, 1 << 18)); CK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 1536;
    const char *names[] = {"dword, one address", "dword, consecutive lanes", "dword, 2% lanes scattered + shared word", "dword, only the 2% lanes (exec-masked)", "dwordx4, consecutive lanes", "dwordx4, consecutive lanes, non-temporal", "buffer dwordx4", "buffer dwordx4 sc0", "buffer dwordx4 sc1", "buffer dwordx4 sc0 sc1", "buffer dwordx4 nt"};
    for (int kind = 0; kind < 11; ++kind) {
        auto launch = [&] {
            if (kind == 0) hipLaunchKernelGGL(k_ta<0>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 1) hipLaunchKernelGGL(k_ta<1>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 2) hipLaunchKernelGGL(k_ta<2>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 3) hipLaunchKernelGGL(k_ta<3>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 4) hipLaunchKernelGGL(k_ta<4>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 5) hipLaunchKernelGGL(k_ta<5>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 6) hipLaunchKernelGGL(k_ta<6>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 7) hipLaunchKernelGGL(k_ta<7>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 8) hipLaunchKernelGGL(k_ta<8>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 9) hipLaunchKernelGGL(k_ta<9>, dim3(blocks), dim3(256), 0, 0, table, out);
            if (kind == 10) hipLaunchKernelGGL(k_ta<10>, dim3(blocks), dim3(256), 0, 0, table, out);
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 100.0 * ms;                                   // per launch
        const double instr_per_cu = 6.0 * 4 * ITER * 8;                 // 6 blocks x 4 waves x ITER x 8 loads
        printf("%-42s %8.1f us per launch  %6.2f ns = %5.1f clk per wave-instruction and CU\n", names[kind], us, 1e3 * us / instr_per_cu, 2.4e3 * us / instr_per_cu);
    }
    return 0;
}

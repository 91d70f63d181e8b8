// Micro-benchmark: what read bandwidth does this MI355X deliver to a kernel shaped like the segment
// scan?  The bar `tools/scan_stress.py` should be read against (the 8 TB/s of the data sheet is not
// reachable by any kernel).
//   rows of `stride` bytes, the first `used` bytes of each read with 16-byte loads by 8 lanes per
//   row (the scan's access shape), `inflight` rows per lane group issued before the first use,
//   xor-reduced and written once per block
//     stride = used = 128: a contiguous stream;  stride 256 / used 128: the scan on 64-slot rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_rows(const char *__restrict__ base, long long n_rows, int stride, int used, unsigned *out) {
    const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
    const int per_row = used / 128;                // 128-byte chunks of a row
    unsigned acc = 0;
    for (long long r0 = (long long)blockIdx.x * 32 * INFLIGHT; r0 < n_rows; r0 += (long long)gridDim.x * 32 * INFLIGHT) {
        int4 v[INFLIGHT][2];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            long long r = r0 + u * 32 + g;
            if (r >= n_rows) r = n_rows - 1;
            const char *row = base + r * stride + l * 16;
            v[u][0] = *(const int4 *)row;
            v[u][1] = per_row > 1 ? *(const int4 *)(row + 128) : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u)
            acc ^= (unsigned)(v[u][0].x ^ v[u][0].y ^ v[u][0].z ^ v[u][0].w ^ v[u][1].x ^ v[u][1].w);
    }
    if (acc == 0x12345u) out[blockIdx.x] = acc;    // (never: keeps the loads alive)
}

// The scan's instruction mix, added one ingredient at a time to the 256-byte-stride / 128-bytes-used stream
// (two rows in flight per lane group, like k_tm_scan): which ingredient costs the bandwidth?
//   MIX 1: + two 4-byte side loads per row (synapse count, owner cell; the 8 lanes of a row share the address)
//   MIX 2: + a bitmap lookup in LDS per synapse (4 per lane and row) behind a block barrier per batch
//   MIX 3: + a dependent 4-byte gather per synapse from a 256 KB table (all but ~2 % of the lanes read word 0)
//   MIX 4: + one 4-byte result per row stored by the row's first lane
//   MIX 5: the same results passed through LDS and stored by one wave, 256 contiguous bytes per batch
template <int MIX>
__global__ __launch_bounds__(256) void k_mix(const char *__restrict__ base, const int *__restrict__ side, const unsigned *__restrict__ table,
                                             long long n_rows, unsigned *out, unsigned *result) {
    __shared__ unsigned bitmap[2048];
    __shared__ unsigned res[64];
    const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
    for (int i = threadIdx.x; i < 2048; i += 256) bitmap[i] = (i * 2654435761u >> 7) & (i * 40503u >> 3) & (i * 9176u >> 2) & 0x11111111u;   // sparse
    __syncthreads();
    unsigned acc = 0;
    for (long long r0 = (long long)blockIdx.x * 64; r0 < n_rows; r0 += (long long)gridDim.x * 64) {
        int4 v[2];
        int n[2] = {0, 0}, c[2] = {0, 0};
        long long r[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            r[u] = r0 + u * 32 + g;
            if (r[u] >= n_rows) r[u] = n_rows - 1;
            if (MIX >= 1) { n[u] = side[r[u]]; c[u] = side[n_rows + r[u]]; }
            v[u] = *(const int4 *)(base + r[u] * 256 + l * 16);
        }
        if (MIX >= 2) __syncthreads();
        unsigned e[2][4], on[2][4], aw[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) { e[u][0] = v[u].x; e[u][1] = v[u].y; e[u][2] = v[u].z; e[u][3] = v[u].w; }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned col = (e[u][q] >> 5) & 0xFFFFu;
                on[u][q] = MIX >= 2 ? (bitmap[col >> 5] >> (col & 31)) & 1u : e[u][q] & 1u;
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) aw[u][q] = MIX >= 3 ? table[on[u][q] ? (e[u][q] >> 5) & 0xFFFFu : 0] : e[u][q];
        unsigned sum[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sum[u] = (unsigned)n[u] + (unsigned)c[u];
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[u] += on[u][q] & (aw[u][q] >> (e[u][q] & 31));
            sum[u] += __shfl_xor(sum[u], 1); sum[u] += __shfl_xor(sum[u], 2); sum[u] += __shfl_xor(sum[u], 4);
            if (MIX == 4) { if (l == 0) result[r[u]] = sum[u]; } else if (MIX == 5) { if (l == 0) res[u * 32 + g] = sum[u]; } else acc ^= sum[u];
        }
        if (MIX >= 2) __syncthreads();
        if (MIX == 5 && threadIdx.x < 64 && r0 + threadIdx.x < n_rows) result[r0 + threadIdx.x] = res[threadIdx.x];
    }
    if (acc == 0x12345u) out[blockIdx.x] = acc;
}

static int run_mix(const char *buf, long long bytes, unsigned *out) {
    const long long n_rows = bytes / 256;
    int *side; unsigned *table, *result;
    CK(hipMalloc(&side, n_rows * 8)); CK(hipMemset(side, 0, n_rows * 8));
    CK(hipMalloc(&table, 1 << 18)); CK(hipMemset(table, 0xFF, 1 << 18));
    CK(hipMalloc(&result, n_rows * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mix = 0; mix <= 5; ++mix) {
        for (int blocks : {1024, 1536, 2048}) {
            auto launch = [&] {
                if (mix == 0) hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(256), 0, 0, buf, side, table, n_rows, out, result);
                if (mix == 1) hipLaunchKernelGGL(k_mix<1>, dim3(blocks), dim3(256), 0, 0, buf, side, table, n_rows, out, result);
                if (mix == 2) hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(256), 0, 0, buf, side, table, n_rows, out, result);
                if (mix == 3) hipLaunchKernelGGL(k_mix<3>, dim3(blocks), dim3(256), 0, 0, buf, side, table, n_rows, out, result);
                if (mix == 4) hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(256), 0, 0, buf, side, table, n_rows, out, result);
                if (mix == 5) hipLaunchKernelGGL(k_mix<5>, dim3(blocks), dim3(256), 0, 0, buf, side, table, n_rows, out, result);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double read = (double)n_rows * (128 + (mix >= 1 ? 8 : 0));
            printf("mix %d  blocks %4d : %7.1f us  %7.1f GB/s (rows + side loads)\n", mix, blocks, 100.0 * ms, read / (ms / 10 * 1e-3) / 1e9);
        }
    }
    return 0;
}

int main(int argc, char **argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 2048ll) << 20;
    char *buf; unsigned *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[3][2] = {{128, 128}, {256, 128}, {256, 256}};
    for (auto &sh : shapes) {
        for (int blocks : {1024, 2048, 4096}) {
            for (int inflight : {2, 4, 8}) {
                const long long n_rows = bytes / sh[0];
                auto launch = [&] {
                    if (inflight == 2) hipLaunchKernelGGL(k_rows<2>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, sh[0], sh[1], out);
                    if (inflight == 4) hipLaunchKernelGGL(k_rows<4>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, sh[0], sh[1], out);
                    if (inflight == 8) hipLaunchKernelGGL(k_rows<8>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, sh[0], sh[1], out);
                };
                launch(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double read = (double)n_rows * sh[1];
                printf("stride %3d used %3d  blocks %4d  rows in flight/group %d : %7.1f us  %7.1f GB/s read\n",
                       sh[0], sh[1], blocks, inflight, 100.0 * ms, read / (ms / 10 * 1e-3) / 1e9);
            }
        }
    }
    return run_mix(buf, bytes, out);
}

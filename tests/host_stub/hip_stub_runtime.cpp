// A HIP runtime made of host memory, for the sanitizer build of the engine's HOST code (tests/test_host_sanitizers.py).
// htm_engine.hip is compiled with `hipcc --offload-host-only -fsanitize=address,undefined` and linked against this file
// instead of libamdhip64: "device" memory is the host heap (so every hipMemcpy / hipMemset of the engine is bounds-checked
// by AddressSanitizer on both sides), kernel launches and graph replays do nothing (device state stays what the host wrote:
// zeros, or an imported state), streams and events are tokens.  What runs for real is everything the library does on the
// host: argument checks, launch plans and graph keys, state import / export with its conversions, the registry of
// handles, the sharded import's row selection.  Test infrastructure only -- nothing under bithtm_amd/ knows about it.
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

namespace {
struct CallConfig { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallConfig g_cfg;
long g_launches = 0, g_graph_launches = 0;
}

extern "C" {
// what the kernel-launch stubs of the host object call
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) { g_cfg = CallConfig{grid, block, shmem, stream}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
    *grid = g_cfg.grid; *block = g_cfg.block; *shmem = g_cfg.shmem; *stream = g_cfg.stream;
    return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) { static void *handle = nullptr; return &handle; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
long bithtm_stub_kernel_launches(void) { return g_launches; }
long bithtm_stub_graph_launches(void) { return g_graph_launches; }
}

static bool grid_ok(dim3 grid, dim3 block) {
    return grid.x >= 1 && grid.y == 1 && grid.z == 1 && block.x >= 1 && block.x <= 1024 && block.y == 1 && block.z == 1;
}

hipError_t hipLaunchKernel(const void *f, dim3 grid, dim3 block, void **, size_t shmem, hipStream_t) {
    if (!f || !grid_ok(grid, block) || shmem > 64 * 1024) abort();      // a launch the device would refuse is a bug of the host code
    ++g_launches;
    return hipSuccess;
}
extern "C" hipError_t hipExtLaunchKernel(const void *f, dim3 grid, dim3 block, void **, size_t shmem, hipStream_t, hipEvent_t, hipEvent_t, int) {
    if (!f || !grid_ok(grid, block) || shmem > 64 * 1024) abort();
    ++g_launches;
    return hipSuccess;
}

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *prop, int) { memset(prop, 0, sizeof(*prop)); prop->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *f, int block, size_t shmem) {
    if (!f || block < 64) abort();
    *n = shmem > 32 * 1024 ? 4 : 6;
    return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t bytes) { *p = calloc(1, bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t n) { memset(dst, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { memset(dst, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = (hipGraph_t)malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t) { *e = (hipGraphExec_t)malloc(8); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { if (!e) abort(); ++g_graph_launches; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

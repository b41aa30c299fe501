// Host stand-in for the HIP runtime, for the AddressSanitizer build of the library's HOST side only (tools/build_asan_host.sh,
// tests/test_asan_host.py).  "Device" memory is host memory, copies are memcpy, kernel launches do nothing: every packer,
// planner and launcher runs its host code (geometry checks, descriptor set-up, pointer arithmetic, buffer sizes) under ASan —
// a hipMemcpy that reads or writes past a buffer is then an ASan report.  Test infrastructure; never linked into the product.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 0;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip stub"; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
// stream capture / graphs: handles are small heap blocks (so leaks and double frees of the library's graph cache are visible to the
// sanitizer); a "captured" stream still executes nothing, like every launch of this stand-in
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = reinterpret_cast<hipGraph_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipGraphGetNodes(hipGraph_t, hipGraphNode_t*, size_t* n) { *n = 7; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = reinterpret_cast<hipGraphExec_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { std::free(e); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { std::free(g); return hipSuccess; }
// registration of the (absent) device code objects
void** __hipRegisterFatBinary(const void*) { static void* h[4]; return h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
static thread_local struct { dim3 g, b; size_t s; hipStream_t st; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t s, hipStream_t st) { g_cfg = {g, b, s, st}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* s, hipStream_t* st) { *g = g_cfg.g; *b = g_cfg.b; *s = g_cfg.s; *st = g_cfg.st; return hipSuccess; }
}

// A HOST-ONLY stand-in for <hip/hip_runtime.h>, for tests/test_host_threads.py: csrc/solver.hip compiled by g++ against
// this header runs its whole host side -- lanes, tickets, staged host batches, finisher threads, the speculative bound
// stage's bookkeeping -- under ThreadSanitizer / AddressSanitizer WITHOUT a GPU.  "Device" memory is host memory, streams
// and events are tokens (everything completes at once), kernel launches do nothing (tests/host_stub_launchers.cpp defines
// the launchers of the other .hip files as no-ops).  The numbers that come out are meaningless; the bookkeeping is real.
// TEST INFRASTRUCTURE ONLY -- never on the product's include path.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
struct stub_stream_ { int id; };
struct stub_event_ { int id; };
typedef stub_stream_* hipStream_t;
typedef stub_event_* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { hipMemoryType type; void* devicePointer; void* hostPointer; };
struct hipDeviceProp_t { int multiProcessorCount; char name[256]; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
// The three kernels solver.hip launches itself (header fetch, state push, host inputs) are grid-stride copy loops: they
// are RUN here, every (block, thread) of the launch one after the other on the calling thread, so that the "device"
// header and the page-locked state mirror hold what they hold on a GPU.
#define hipLaunchKernelGGL(k, g, b, l, s, ...) stub_run_grid((g), (b), [&] { k(__VA_ARGS__); })
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __threadfence() ((void)0)
#define __syncthreads() ((void)0)
struct stub_dim3_ { unsigned x, y, z; };
extern thread_local stub_dim3_ blockIdx, threadIdx, gridDim, blockDim;  // (thread-local: lanes launch from several threads)

template <class F>
inline void stub_run_grid(dim3 g, dim3 b, F f) {
  gridDim = {g.x, g.y, g.z};
  blockDim = {b.x, b.y, b.z};
  for (unsigned bx = 0; bx < g.x; ++bx)
    for (unsigned tx = 0; tx < b.x; ++tx) {
      blockIdx = {bx, 0, 0};
      threadIdx = {tx, 0, 0};
      f();
    }
}

inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new stub_stream_{0}; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new stub_stream_{0}; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = new stub_stream_{0}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new stub_event_{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new stub_event_{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "stub"; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) { a->type = hipMemoryTypeHost; a->devicePointer = const_cast<void*>(p); a->hostPointer = const_cast<void*>(p); return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

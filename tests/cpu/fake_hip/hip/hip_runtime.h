// TEST INFRASTRUCTURE: a host-only stand-in for <hip/hip_runtime.h>, for ONE purpose -- compiling the host side of librtd.so
// (csrc/rtd_api.hip: plan arenas, window offsets, the device-memory pool, pinned slabs, gathered-array offsets) with g++ and the
// address / undefined-behaviour sanitizers, and running it on the CPU (tests/test_host_asan.py; SURVEY section 5 lists a
// sanitizer build of the host side, the GPU pool has no device sanitizer).  "Device" memory is host heap memory, so every
// copy, fill and pointer the host code hands to a kernel is bounds-checked; streams and events are tokens (everything is
// synchronous); a __global__ kernel of rtd_api.hip itself is RUN, thread by thread, by hipLaunchKernelGGL.  The kernels of the
// other translation units are replaced by shadow launchers that touch exactly the extents the real kernels read and write
// (tests/cpu/host_asan_shadow.cpp).  Nothing here is product code and nothing of it is measured.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct fake_stream* hipStream_t;
typedef struct fake_event* hipEvent_t;
typedef void* hipDeviceptr_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct fake_idx { unsigned x, y, z; };
extern thread_local fake_idx blockIdx, threadIdx, blockDim, gridDim;

// allocation bookkeeping of the fake device (FAKE_HIP_TOTAL bytes; an allocation beyond what is "free" fails like the real one)
extern "C" {
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
}
inline hipError_t hipMalloc(double** p, size_t n) { return hipMalloc((void**)p, n); }
template <typename T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }

inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "error"; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

// inter-process handles do not exist here: the tests' stand-in transport (tests/stub/rccl_stub.cpp), built against this header for
// the CPU, finds its IPC probe failing and stages through shared memory instead
typedef struct { char reserved[64]; } hipIpcMemHandle_t;
enum { hipIpcMemLazyEnablePeerAccess = 1 };
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorInvalidValue; }
inline hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorInvalidValue; }
inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
inline hipError_t hipMemGetAddressRange(hipDeviceptr_t*, size_t*, hipDeviceptr_t) { return hipErrorInvalidValue; }

// a kernel of the translation unit itself runs here, one thread after the other (none of rtd_api.hip's kernels synchronises)
template <typename K, typename... A>
inline void fake_launch(K kernel, dim3 grid, dim3 block, A... args) {
  gridDim = {grid.x, grid.y, grid.z};
  blockDim = {block.x, block.y, block.z};
  for (unsigned b = 0; b < grid.x; ++b)
    for (unsigned t = 0; t < block.x; ++t) {
      blockIdx = {b, 0, 0};
      threadIdx = {t, 0, 0};
      kernel(args...);
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) fake_launch(kernel, grid, block, __VA_ARGS__)

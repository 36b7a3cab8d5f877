// TEST INFRASTRUCTURE ONLY -- a minimal stand-in for <hip/hip_runtime.h> that
// lets g++ compile hector_amd/csrc/*.hip/*.cpp for the HOST so the `-m "not gpu"`
// suite can execute the exact kernel source, one lane at a time, against the
// oracle.  Lanes of these kernels never exchange data (except the statistics
// kernel, which is NOT emulated faithfully: __shfl_down returns its input), so
// sequential execution with __any(x) == x reproduces each member's result.
// The library built with this shim reports hx_backend() == "host-emulation" and
// the product loader (hector_amd/_lib.py) refuses to use it outside tests.
#pragma once
#define HX_HOST_EMULATION 1
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__

using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hx_emul_idx { unsigned x, y, z; };
extern hx_emul_idx threadIdx, blockIdx, blockDim, gridDim;

inline bool __any(bool p) { return p; }
template <class T> inline T __shfl_down(T v, int, int = 64) { return v; }
inline void __syncthreads() {}
inline double rsqrt(double x) { return 1.0 / sqrt(x); }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef struct hx_emul_stream *hipStream_t;
struct hx_emul_event { std::chrono::steady_clock::time_point t; };
typedef hx_emul_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

inline const char *hipGetErrorString(hipError_t) { return "host-emulation error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
struct double2 { double x, y; };
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline long long __double_as_longlong(double x) { long long r; std::memcpy(&r, &x, 8); return r; }
inline double __longlong_as_double(long long x) { double r; std::memcpy(&r, &x, 8); return r; }
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  *free_b = *total_b = (size_t)1 << 40; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hx_emul_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) {
  *p = (T *)malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
  memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
  memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h,
                              hipMemcpyKind) {
  for (size_t r = 0; r < h; ++r) memcpy((char *)d + r * dp, (const char *)s + r * sp, w);
  return hipSuccess;
}

inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h,
                                   hipMemcpyKind k, hipStream_t) {
  return hipMemcpy2D(d, dp, s, sp, w, h, k);
}

template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
  gridDim = {grid.x, grid.y, grid.z};
  blockDim = {block.x, block.y, block.z};
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx)
      for (unsigned tx = 0; tx < block.x; ++tx) {
        blockIdx = {bx, by, 0};
        threadIdx = {tx, 0, 0};
        kernel(args...);
      }
}

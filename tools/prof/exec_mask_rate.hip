// Does the vector ALU skip the 16-lane passes of an fp64 instruction whose lanes are all masked
// off?  (If it did, an ensemble of exactly one wavefront per SIMD could run as half-filled
// wavefronts, two per SIMD.)  Eight independent v_fma_f64 chains, one wavefront per SIMD, with
// EXEC = all 64 lanes, the lower 32, the lower 16; core-clock ticks per instruction.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o gpuwork/exec_mask_rate tools/prof/exec_mask_rate.hip
//   gpurun -- gpuwork/exec_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int CH>
__global__ void k(double *out, unsigned long long *cyc, double a, double b, int iters, int lanes) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i * 1e-3 + threadIdx.x * 1e-6;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < lanes) {      // EXEC narrowed for the whole timed region
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
      REP16(
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[0]) : "v"(a), "v"(b));
        if (CH > 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[1]) : "v"(a), "v"(b));
        if (CH > 2) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[2]) : "v"(a), "v"(b));
                      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[3]) : "v"(a), "v"(b)); }
        if (CH > 4) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[4]) : "v"(a), "v"(b));
                      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[5]) : "v"(a), "v"(b));
                      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[6]) : "v"(a), "v"(b));
                      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[7]) : "v"(a), "v"(b)); }
      )
    }
    t1 = clock64();
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
void run(int lanes, int waves_per_simd) {
  const int blocks = 1024 * waves_per_simd, threads = 64, iters = 200;
  double *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<blocks, threads>>>(out, cyc, 0.999, 1e-3, iters, lanes);
  hipEventRecord(e0);
  k<CH><<<blocks, threads>>>(out, cyc, 0.999, 1e-3, iters, lanes);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * CH;
  printf("chains %d  active lanes %2d  waves/SIMD %d: %6.2f ticks per v_fma_f64 per wave, %7.3f ns per instruction (event, whole launch)\n",
         CH, lanes, waves_per_simd, h[0] / n, ms * 1e6 / n);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 2; ++w)
    for (int lanes : {64, 32, 16}) { run<1>(lanes, w); run<8>(lanes, w); }
  return 0;
}

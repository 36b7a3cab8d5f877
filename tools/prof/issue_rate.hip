// Issue-rate microbenchmark behind the cost model in DESIGN.md section 6: one wavefront per SIMD, chains of
// fp64 ops, dependent or independent; LDS round trip; workgroup barrier.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o gpuwork/issue_rate tools/prof/issue_rate.hip
//   gpurun -- gpuwork/issue_rate      (results: profiles/r02_issue_rate.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int CH, int OP>
__global__ void k(double *out, unsigned long long *cyc, double a, double b, int iters) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i * 1e-3 + threadIdx.x * 1e-6;
  unsigned long long t0 = __builtin_readcyclecounter();
  t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {  // fma
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
    } else if (OP == 1) {  // rcp
      REP16(
        asm volatile("v_rcp_f64 %0, %0" : "+v"(v[0]));
        if (CH > 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[1]));
        if (CH > 2) { asm volatile("v_rcp_f64 %0, %0" : "+v"(v[2])); asm volatile("v_rcp_f64 %0, %0" : "+v"(v[3])); }
      )
    } else if (OP == 2) {  // add
      REP16(
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[0]) : "v"(b));
        if (CH > 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[1]) : "v"(b));
        if (CH > 2) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[2]) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[3]) : "v"(b)); }
      )
    } else if (OP == 3) {  // fp32 fma for comparison
      float *f = reinterpret_cast<float *>(v);
      float fa = (float)a, fb = (float)b;
      REP16(
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(fa), "v"(fb));
        if (CH > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[2]) : "v"(fa), "v"(fb));
        if (CH > 2) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[4]) : "v"(fa), "v"(fb)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[6]) : "v"(fa), "v"(fb)); }
      )
    } else if (OP == 4) {  // cndmask pair (a 64-bit select) after a compare
      REP16(
        asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[0]) : "v"(b));
      )
    } else if (OP == 5) {  // LDS round trip: write, read back, dependent
      __shared__ double s[64];
      REP16(
        s[threadIdx.x] = v[0]; asm volatile("s_waitcnt lgkmcnt(0)"); v[0] = ((volatile double *)s)[threadIdx.x ^ 1];
      )
    } else if (OP == 6) {  // barrier (workgroup of 128)
      REP16( __syncthreads(); )
    }
  }
  unsigned long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH, int OP>
void run(const char *name, int blocks, int threads, int per) {
  double *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8 * blocks);
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH, OP><<<blocks, threads>>>(out, cyc, 0.999, 1e-3, iters);
  hipEventRecord(e0);
  k<CH, OP><<<blocks, threads>>>(out, cyc, 0.999, 1e-3, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * per;
  printf("%-34s blocks %5d x %3d: %7.2f clock64 ticks/op  %8.3f ns/op (event)\n", name, blocks, threads, h[0] / n, ms * 1e6 / n);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1, 1024}) {
    run<1, 0>("fma f64, 1 dependent chain", blocks, 64, 1);
    run<2, 0>("fma f64, 2 chains", blocks, 64, 2);
    run<4, 0>("fma f64, 4 chains", blocks, 64, 4);
    run<8, 0>("fma f64, 8 chains", blocks, 64, 8);
    run<1, 2>("add f64, 1 chain", blocks, 64, 1);
    run<4, 2>("add f64, 4 chains", blocks, 64, 4);
    run<1, 1>("rcp f64, 1 chain", blocks, 64, 1);
    run<4, 1>("rcp f64, 4 chains", blocks, 64, 4);
    run<1, 3>("fma f32, 1 chain", blocks, 64, 1);
    run<4, 3>("fma f32, 4 chains", blocks, 64, 4);
    run<1, 4>("max f64, 1 chain", blocks, 64, 1);
    run<1, 5>("LDS write/wait/read round trip", blocks, 64, 1);
    run<1, 6>("s_barrier, 128 threads", blocks, 128, 1);
    run<4, 0>("fma f64, 4 chains, 2 waves/SIMD?", blocks * 2, 64, 4);
  }
  return 0;
}

// Scalar-load latency microbenchmark: what an EXPOSED s_load + s_waitcnt costs the one resident
// wavefront of a SIMD (the year loop has ~16 of them per model year: tools/isa_stats.py).
//   hipcc --offload-arch=gfx950 -O2 -o gpuwork/smem_latency tools/prof/smem_latency.hip
//   gpurun -- gpuwork/smem_latency
// A chain of s_load_dwordx2 through a table whose every entry holds its own address (scalar
// cache hits after the first touch); the same with x16 loads; and a load issued K independent
// VALU instructions ahead of its wait (how much cover hides it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE, int COVER>
__global__ void k(const unsigned long long *tab, unsigned long long *cyc, double *out, int iters) {
  const unsigned long long *p = tab + (blockIdx.x & 7);   // uniform
  unsigned long long q = (unsigned long long)p;
  double v = threadIdx.x * 1e-6, a = 1.0000001, b = 1e-9;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      REP16(asm volatile("s_load_dwordx2 %0, %0, 0x0\n\ts_waitcnt lgkmcnt(0)" : "+s"(q));)
    } else if (MODE == 1) {
      v16i r;
      REP16(asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(q));
            asm volatile("" :: "s"(r));)
    } else {
      // load, COVER independent fp64 FMAs, wait: the chain continues through the loaded pointer
      REP16(asm volatile("s_load_dwordx2 %0, %0, 0x0" : "+s"(q));
            for (int c = 0; c < COVER; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q));)
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
  out[blockIdx.x * 64 + threadIdx.x] = v + (double)(q & 1);
}

template <int MODE, int COVER>
void run(const char *what, int blocks, const unsigned long long *d_tab, unsigned long long *d_cyc, double *d_out) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, COVER>), dim3(blocks), dim3(64), 0, 0, d_tab, d_cyc, d_out, iters);
  hipLaunchKernelGGL((k<MODE, COVER>), dim3(blocks), dim3(64), 0, 0, d_tab, d_cyc, d_out, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(blocks);
  hipMemcpy(c.data(), d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : c) s += (double)x;
  printf("%-58s blocks %5d: %8.1f clock64 ticks per load\n", what, blocks, s / blocks / (16.0 * iters));
}

int main() {
  unsigned long long *d_tab, *d_cyc; double *d_out;
  hipMalloc(&d_tab, 64 * 8); hipMalloc(&d_cyc, 8 * 4096); hipMalloc(&d_out, 8 * 64 * 4096);
  std::vector<unsigned long long> h(64);
  for (int i = 0; i < 64; ++i) h[i] = (unsigned long long)(d_tab + i);
  hipMemcpy(d_tab, h.data(), 64 * 8, hipMemcpyHostToDevice);
  for (int blocks : {1, 1024}) {
    run<0, 0>("s_load_dwordx2 + wait, dependent chain (cache hit)", blocks, d_tab, d_cyc, d_out);
    run<1, 0>("s_load_dwordx16 + wait", blocks, d_tab, d_cyc, d_out);
    run<2, 0>("s_load_dwordx2, 0 FMAs, wait", blocks, d_tab, d_cyc, d_out);
    run<2, 4>("s_load_dwordx2, 4 independent-of-load FMAs (one chain), wait", blocks, d_tab, d_cyc, d_out);
    run<2, 8>("s_load_dwordx2, 8 FMAs, wait", blocks, d_tab, d_cyc, d_out);
    run<2, 16>("s_load_dwordx2, 16 FMAs, wait", blocks, d_tab, d_cyc, d_out);
    run<2, 32>("s_load_dwordx2, 32 FMAs, wait", blocks, d_tab, d_cyc, d_out);
  }
  return 0;
}

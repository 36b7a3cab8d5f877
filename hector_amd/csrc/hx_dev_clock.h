// hx_dev_clock.h -- section clock of the profiling build (tools/prof/phase_clock.py)
// Part of the device code of hx_kernels.hip (one translation unit).
#pragma once

namespace {

// Profiling build only (-DHX_PHASE_CLOCK, tools/prof/phase_clock.sh): s_memtime stamps at the
// boundaries of the sections of a model year, accumulated per wavefront in LDS and written over
// the first HX_NCLK year rows of the global_tas output at the end of the launch.  The product
// build compiles none of it.
// Product build: start (which = 0) and end (1) of a wavefront of a year-loop launch in ticks of the
// constant 100 MHz clock (s_memrealtime), one store by lane 0 each -- HxBuffers::wave_clk,
// hx_wave_clock(): the launch lasts as long as its last wavefront.
__device__ __forceinline__ void hx_wave_stamp(const HxBuffers &buf, int wave, int which, int lane) {
#ifndef HX_HOST_EMULATION
  if (buf.wave_clk && lane == 0)
    ((long long HX_GLOBAL *)buf.wave_clk)[2 * wave + which] = (long long)__builtin_amdgcn_s_memrealtime();
#else
  if (buf.wave_clk && lane == 0) buf.wave_clk[2 * wave + which] = which;   // (no clock on the host)
#endif
}

#ifdef HX_PHASE_CLOCK
#define HX_NCLK 24
__shared__ long long hx_s_clk[HX_NCLK + 1];
__device__ __forceinline__ void hx_stamp(int k) {
  volatile long long *clk = hx_s_clk;
  asm volatile("" ::: "memory");
  const long long now = (long long)__builtin_readcyclecounter();
  clk[k] += now - clk[HX_NCLK];
  clk[HX_NCLK] = now;
  asm volatile("" ::: "memory");
}
#define HX_STAMP(m, k) hx_stamp((k))
#define HX_COUNT(m, k) (((volatile long long *)hx_s_clk)[(k)] += 1)
#else
#define HX_STAMP(m, k)
#define HX_COUNT(m, k)
#endif


}  // namespace

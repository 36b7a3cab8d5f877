// hx_addrspace.h -- pointers fetched from the kernel-argument block are "generic"
// to the compiler, which then emits flat_load/flat_store (slower, and they tie up
// both wait counters).  Everything per-member lives in HBM: say so.
#pragma once
#define HX_GLOBAL __attribute__((address_space(1)))
typedef const double HX_GLOBAL *hx_gcd;  // global const double*
typedef double HX_GLOBAL *hx_gd;         // global double*
typedef unsigned HX_GLOBAL *hx_gu;
// wave-uniform read-only tables (scenario series, shared DOECLIM kernel): the constant
// address space lets the compiler use scalar loads (s_load) and SGPR operands
#define HX_CONSTANT __attribute__((address_space(4)))
typedef const double HX_CONSTANT *hx_ccd;
#define HX_CCD(p) ((hx_ccd)(p))
#define HX_GCD(p) ((hx_gcd)(p))
#define HX_GD(p) ((hx_gd)(p))
#define HX_GU(p) ((hx_gu)(p))
// hardware reciprocal seed (v_rcp_f64: ~4.6e-8 relative, measured), refined in hx_recip()
#define HX_RCP(x) __builtin_amdgcn_rcp(x)
// hardware reciprocal-square-root seed (v_rsq_f64), refined in hx_sqrt()
#define HX_RSQ(x) __builtin_amdgcn_rsq(x)
// v_log_f32 / v_exp_f32 themselves (base 2, normal-range arguments: no denormal scaling around them)
#define HX_LOG2F(x) __builtin_amdgcn_logf(x)
#define HX_EXP2F(x) __builtin_amdgcn_exp2f(x)
// the fp64 matrix pipe (v_mfma_f64_16x16x4_f64) is there for DOECLIM's history contraction
#ifndef HX_HAS_MFMA
#define HX_HAS_MFMA 1
#endif

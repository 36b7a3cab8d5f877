// TEST INFRASTRUCTURE ONLY: host stand-in for hector_amd/csrc/hx_addrspace.h
#pragma once
#define HX_GLOBAL
typedef const double *hx_gcd;
typedef double *hx_gd;
typedef unsigned *hx_gu;
#define HX_CONSTANT
typedef const double *hx_ccd;
#define HX_CCD(p) ((hx_ccd)(p))
#define HX_GCD(p) ((hx_gcd)(p))
#define HX_GD(p) ((hx_gd)(p))
#define HX_GU(p) ((hx_gu)(p))
#define HX_RCP(x) (1.0 / (x))
#define HX_RSQ(x) (1.0 / sqrt(x))
#define HX_LOG2F(x) log2f(x)
#define HX_EXP2F(x) exp2f(x)
#define HX_HAS_MFMA 0

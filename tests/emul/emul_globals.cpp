// TEST INFRASTRUCTURE ONLY: storage for the host-emulation shim.
#include <hip/hip_runtime.h>
hx_emul_idx threadIdx, blockIdx, blockDim, gridDim;
double s_lds[3 * 64 * 64];  // the run kernel's dynamic LDS (extern __shared__)

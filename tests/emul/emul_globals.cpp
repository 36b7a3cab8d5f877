// TEST INFRASTRUCTURE ONLY: storage for the host-emulation shim.
#include <hip/hip_runtime.h>
hx_emul_idx threadIdx, blockIdx, blockDim, gridDim;


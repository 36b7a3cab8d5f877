// TEST INFRASTRUCTURE ONLY: storage for the host-emulation shim.
#include <hip/hip_runtime.h>
hx_emul_idx threadIdx, blockIdx, blockDim, gridDim;


// the looped kernels' dynamic LDS park (hx_dev_member.h), sized for 16 biomes
double hx_dyn_park[256][64];

// Explicit instantiations of the mid-size step kernels (see slk_inst_big.hip).
#define SLK_INST_UNIT 1
#include <hip/hip_runtime.h>
#include "../../include/slk.h"
#include "slk_kernels.hpp"

namespace slk {
template __global__ void msckf_step_kernel<10, 512, -1, 0>(KArgs);
template __global__ void msckf_step_kernel<8, 256, -1, 0>(KArgs);
template __global__ void msckf_step_kernel<6, 256, -1, 0>(KArgs);
template __global__ void msckf_step_kernel<5, 256, -1, 0>(KArgs);
} // namespace slk

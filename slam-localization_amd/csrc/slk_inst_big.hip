// Explicit instantiations of the largest step kernels in a translation unit of their own: the library builds its
// translation units in parallel (slam-localization_amd/build.py); slk_api.hip declares these `extern template`.
#define SLK_INST_UNIT 1
#include <hip/hip_runtime.h>
#include "../../include/slk.h"
#include "slk_kernels.hpp"

namespace slk {
template __global__ void msckf_step_kernel<13, 512, -1, 0>(KArgs);
template __global__ void msckf_step_kernel<13, 512, 31, 8>(KArgs);
} // namespace slk

// Optional per-launch timing of the HBM-side kernels (bench.py's roofline_hbm block): when switched on, a launch goes through
// hipExtLaunchKernelGGL, which stamps the kernel's own start/stop into two events on the launch stream (the same mechanism as
// the conv launches' profile slots); off (the default) it is a plain launch.  Process-wide, instrumentation only.  The records
// live in flow_ops.hip (hbm_profile_begin); this header is the launch wrapper shared by flow_ops.hip and sampler_ops.hip.
#pragma once
#include "vstab_internal.h"
#include <hip/hip_ext.h>

namespace vstab {

// profiling on: creates the two events of one launch, books `alg_bytes` under `slot` and returns them; off: *a = *b = nullptr
hipError_t hbm_profile_begin(int slot, double alg_bytes, hipEvent_t *a, hipEvent_t *b);

template <typename... KArgs, typename... Args>
static hipError_t launch_timed(int slot, double alg_bytes, void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t stream, Args... args)
{
    hipEvent_t a = nullptr, b = nullptr;
    const hipError_t e = hbm_profile_begin(slot, alg_bytes, &a, &b);
    if (e != hipSuccess) return e;
    if (a) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, a, b, 0, static_cast<KArgs>(args)...);
    else kernel<<<grid, block, 0, stream>>>(static_cast<KArgs>(args)...);
    return hipGetLastError();
}

}  // namespace vstab

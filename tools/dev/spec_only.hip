// Development aid (never shipped): the configs[1] kernel alone -- axis-aligned member of the 4-slot latency kernel.
#include <hip/hip_runtime.h>
#include "../../dyobav-mpcnwta-warehouse_amd/csrc/nmpc_device.h"
#include "../../dyobav-mpcnwta-warehouse_amd/csrc/nmpc_spec.h"
#ifndef DEV_RS
#define DEV_RS 4
#endif
extern "C" __global__ __launch_bounds__(64 * 4, 3) void spec_axis(nmpc::KParams<float> kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int inst = nmpc::dispatch_index(kp);
    if (nmpc::finished_in_pilot<float>(inst)) return;
    nmpc::solve_instance_spec<float, 3, false, DEV_RS, true>(kp, inst, reinterpret_cast<float*>(smem));
}

// rdv_general.h — host entry of the general rigid-body step kernels (csrc/rdv_general.hip)
#pragma once
#include "rdv_kernels.h"
namespace rdv {
// partner_waves (and no diagnostics): step_kernel_general, 512-thread workgroups of 256 envs, the target's RK45 on waves 4-7;
// otherwise step_kernel<ST, diag, true> on `fused_grid` workgroups of kBlock threads (A.xcd_per as set by the caller)
void launch_step_general(bool f32, bool diag, bool partner_waves, int64_t n, dim3 fused_grid, hipStream_t s, const DevParams* dev_params, const StepArgs& A);
}  // namespace rdv

// rdv_tiles.h — host entry of the tile-loop step kernel (csrc/rdv_tiles.hip)
#pragma once
#include "rdv_kernels.h"
namespace rdv {
// launches step_kernel_tiles<float | double> on `grid` workgroups of kBlock threads (A.xcd_per != 0: grid is a multiple of 8)
void launch_step_tiles(bool f32, dim3 grid, hipStream_t s, const DevParams* dev_params, const StepArgs& A);
}  // namespace rdv

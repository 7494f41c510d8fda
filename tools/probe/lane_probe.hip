// device-only compile of the lane-per-problem tree kernel and the derivative kernel (stream form) for ISA inspection (see deriv2_probe.hip)
#include "smpc_engine.h"
using namespace smpc;
typedef Dims<13, 4> D;
template __global__ void smpc::kernel_entry<LaneKernelArgs<D>, lane_tree_body<D, 1, true>, 64, SMPC_LANE_MINW, 0>(LaneKernelArgs<D>);
template __global__ void smpc::kernel_entry<LaneKernelArgs<D>, lane_tree_body<D, 1, false>, 64, SMPC_LANE_MINW, 0>(LaneKernelArgs<D>);
template __global__ void smpc::kernel_entry<StageKernelArgs<D>, deriv2_body<D, true>, 64, 2, 0>(StageKernelArgs<D>);

// device-only compile of the lane-per-problem tree kernel for ISA inspection (see deriv2_probe.hip)
#include "smpc_engine.h"
using namespace smpc;
typedef Dims<13, 4> D;
template __global__ void smpc::kernel_entry<LaneKernelArgs<D>, lane_tree_body<D, 1>, 64, SMPC_LANE_MINW, 0>(LaneKernelArgs<D>);

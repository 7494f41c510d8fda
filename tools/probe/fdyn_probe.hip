// device-only compile of the Go2 full-dynamics stage kernel for ISA inspection (see deriv2_probe.hip)
#include "smpc_full_engine.h"
using namespace smpc;
typedef FullDims<13, 4, 3> D;
template __global__ void smpc::kernel_entry<StageKernelArgs<D>, fdyn_deriv_body<D>, 64, 1, 0>(StageKernelArgs<D>);

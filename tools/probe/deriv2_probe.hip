// device-only compile of one kernel for ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Isimple-mpc_amd/csrc -Iinclude --cuda-device-only -S tools/probe/deriv2_probe.hip -o /tmp/deriv2.s
#include "smpc_engine.h"
using namespace smpc;
typedef Dims<13, 4> D;
template __global__ void smpc::kernel_entry<StageKernelArgs<D>, deriv2_body<D, false>, 64, 2, 0>(StageKernelArgs<D>);

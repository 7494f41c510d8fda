// device-only compile of the fused centroidal control-step kernel for ISA inspection (see deriv2_probe.hip)
#include "smpc_cent_engine.h"
using namespace smpc;
typedef CentDims<4> DC;
template __global__ void smpc::kernel_entry<CentStepArgs<DC>, cent_step_body<DC>, 64, 2, 0>(CentStepArgs<DC>);

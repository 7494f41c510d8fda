// device-only compile of the kinodynamics Riccati sweep for ISA inspection (see deriv2_probe.hip)
#include "smpc_engine.h"
using namespace smpc;
typedef Dims<13, 4> D;
template __global__ void smpc::kernel_entry<SolverArgs<D>, riccati_kino_body<D, false>, 64, 2, 0>(SolverArgs<D>);

"""Per-kernel profile driver of the inverse-dynamics QP path: `rocprofv3 --kernel-trace --stats -- python3 tools/id_profile.py [batch] [calls]`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
import simple_mpc  # noqa: E402
from simple_mpc import presets as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fixed = len(sys.argv) > 3 and sys.argv[3] == "fixed"
lib = simple_mpc._capi.SmpcLib(os.environ["SMPC_VARIANT_LIB"]) if os.environ.get("SMPC_VARIANT_LIB") else None  # (kernel experiments)
mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
for n in P.GO2_FEET:
    mh.addPointFoot(n, "root_joint")
eff, vmax = np.array([23.7, 23.7, 45.43] * 4), np.array([30.1, 30.1, 15.7] * 4)
st = dict(kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1e-3, w_contact_motion=1.0)
kw = dict(admm_iters=int(os.environ.get("ADMM_ITERS", "100")), admm_tol=-1.0) if fixed else {}
kid = simple_mpc.KinodynamicsID(mh, 1e-3, st, eff, vmax, batch=B, lib=lib, **kw)
X = P.random_states(mh, B, scale=0.3)
rng = np.random.default_rng(5)
for k in range(calls):
    Xk = X + np.concatenate([np.zeros((B, 7)), rng.normal(0.0, 2e-3, (B, X.shape[1] - 7))], axis=1)
    kid.solve(0.0, Xk[:, : mh.nq], Xk[:, mh.nq :])
print("max residual", kid.resid.max())

"""Error trajectory of the kinodynamics solve with active friction cones, HIP library vs oracle: python tools/cone_err.py [H] [iters] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
H = int(sys.argv[1]) if len(sys.argv) > 1 else 50
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lib = S.emu_lib() if os.environ.get('SMPC_USE_EMU') else None
om, gm, rb = S.make_pair(3, max_iters=iters, lib=lib, horizon=H, settings_override={"force_cone": True, "mu": 0.1}, walk=(0.6, 0.4, 0, 0, 0, 0.5))
X = S.random_states(rb, 3)
for it in range(steps):
    om.iterate(X); gm.iterate(X)
    print(it, 'xs %.1e us %.1e' % (S.rel_err(om.xs, gm.xs), S.rel_err(om.us, gm.us)), 'alpha', om.info[:, 2], gm.info[:, 2], 'active', int((om.vs[:, :, 24:] != 0).sum()))
    X = om.xs[:, 1, :].copy()

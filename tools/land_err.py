"""Error trajectory of the kinodynamics solve with land_cstr rows, product vs oracle: python tools/land_err.py [H] [iters] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
H = int(sys.argv[1]) if len(sys.argv) > 1 else 50
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lib = S.emu_lib() if os.environ.get('SMPC_USE_EMU') else None
SO = {"land_cstr": True}
MO = {"T_fly": 8, "T_contact": 4}
om, rb, _ = S.make_oracle(2, iters, H, settings_override=SO, mpc_override=MO)
gm, _, _, _ = S.make_product(2, iters, lib, H, settings_override=SO, mpc_override=MO)
cs = np.ones((24, 4), np.uint8)  # short trot: 4 all feet, 8 with FL + RR in the air, 4 all feet, 8 with FR + RL in the air
cs[4:12, [0, 3]] = 0
cs[16:24, [1, 2]] = 0
for m in (om, gm):
    m.generateCycleHorizon(cs)
    m.switchToWalk(np.array([0.3, 0, 0, 0, 0, 0.1]))
X = S.random_states(rb, 2)
for it in range(steps):
    om.iterate(X); gm.iterate(X)
    print(it, 'xs %.1e us %.1e' % (S.rel_err(om.xs, gm.xs), S.rel_err(om.us, gm.us)), 'alpha', om.info[:, 2], gm.info[:, 2], 'land mult', int((om.vs[:, :, 24:] != 0).sum()),
          'prim %.2e %.2e dphi %.6e %.6e' % (om.info[0][4], gm.info[0][4], om.info[0][1], gm.info[0][1]))
    X = om.xs[:, 1, :].copy()

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else 'tile'
gm, rb, _, _ = S.make_product(B, max_iters=iters)
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))
if mode == 'tile':
    X = np.tile(S.random_states(rb, 64), (B//64, 1))
else:
    X = S.random_states(rb, B)
for step in range(4):
    gm.iterate(X)
    xs = gm.xs; info = gm.info
    bad = np.where(~np.isfinite(xs).all(axis=(1,2)))[0]
    print('step', step, 'nan instances', len(bad), bad[:10], 'info nonfinite', (~np.isfinite(info)).any(axis=1).sum(),
          'alpha hist', np.unique(info[:,2], return_counts=True), 'prim_new max', np.nanmax(info[:,8]), 'phi0 max', np.nanmax(info[:,0]))
    if len(bad):
        i = bad[0]; print(' first bad: info', info[i,:12]); t_bad = np.where(~np.isfinite(xs[i]).all(axis=1))[0]; print(' bad t', t_bad[:10])
    X = xs[:,1,:].copy()
    if len(bad): break

import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import mpc_setup as S, oracle_lib as O
B=1024
gm, rb, _, _ = S.make_talos_kino_product(B, max_iters=3)
gm.generateCycleHorizon(O.walk_cycle()); gm.switchToWalk(np.array([0.1,0,0,0,0,0.]))
X = S.talos_random_states(rb, B, scale=0.7)
gm.iterate(X); X = gm.xs[:,1,:].copy()
gm.set_profiling(True); gm.reset_kernel_times()
for _ in range(3): gm.iterate(X)
kt = gm.kernel_times()
print(' '.join('%s %.2f' % (k, v[0]/max(1,v[1])) for k,v in kt.items()))
if not os.environ.get('SMPC_PHASE_PROFILE'):
    sys.exit(0)
out = np.zeros(64); gm._lib.check(gm._lib.L.smpc_debug_get_phase_cycles(gm._h, out))
names = ['load','kin','composite','M/J','cholM/W','G/Gi','prox/a','eval-tail','forces','dk/Ak/Jc','R1','R2','solve','-','tables/grad','AB','Hessian stores','Bc','stage-out','GN rows']
extra = [(26, 'H0'), (27, 'W JT'), (28, 'JT^T W JT'), (30, 'knot rows'), (29, 'dual max / end')]
nd = 4*3
print('deriv phase cycles:', ' '.join('%s %.0f' % (nm, out[i]/nd) for i,nm in enumerate(names)), ' '.join('%s %.0f' % (nm, out[i]/nd) for i,nm in extra), '| total %.0f' % ((out[:20].sum() + sum(out[i] for i,_ in extra))/nd))

"""Per-kernel timing of the full-dynamics engine on the GPU (no torch): python tools/quick_bench_full.py [B] [iters] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H = int(sys.argv[4]) if len(sys.argv) > 4 else 50
gm, rb, _, _ = S.make_full_product(B, max_iters=iters, horizon=H)
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))
X = S.random_states(rb, B)
gm.iterate(X); X = gm.xs[:,1,:].copy()
gm.set_profiling(True); gm.reset_kernel_times()
t0 = time.time()
for _ in range(steps):
    gm.iterate(X)
dt = (time.time()-t0)/steps
kt = gm.kernel_times()
print('FULL B=%d H=%d k=%d: %.1f ms/step  %.0f steps/s |' % (B, H, iters, dt*1e3, B/dt), ' '.join('%s %.2f' % (k, v[0]/max(1,v[1])) for k,v in kt.items()), '| finite', bool(np.isfinite(gm.info).all()))
if os.environ.get('SMPC_PHASE_PROFILE'):
    out = np.zeros(64); gm._lib.check(gm._lib.L.smpc_debug_get_phase_cycles(gm._h, out))
    names = ['load','kin','composite','M/J','cholM/W','G/Gi','prox/a','eval-tail','forces','dk/Ak/Jc','R1','R2','solve','-','tables/grad','AB','Hessian stores','Bc','stage-out','GN rows']
    extra = [(26, 'H0'), (27, 'W JT'), (28, 'JT^T W JT'), (30, 'knot rows'), (29, 'dual max / end')]
    nd = (steps+1)*iters
    print('eval-tail parts:', ' '.join('%s %.0f' % (nm, out[20+i]/nd) for i,nm in enumerate(['se3','defect','resid','weighted','cost'])))
    print('deriv phase cycles (inst 0, stage 17):', ' '.join('%s %.0f' % (nm, out[i]/nd) for i,nm in enumerate(names)), ' '.join('%s %.0f' % (nm, out[i]/nd) for i,nm in extra), '| total %.0f' % ((out[:20].sum() + sum(out[i] for i,_ in extra))/nd))

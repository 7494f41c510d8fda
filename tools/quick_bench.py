"""Quick per-kernel timing on the GPU (no torch): python tools/quick_bench.py [B] [iters] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gm, rb, _, _ = S.make_product(B, max_iters=iters)
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))
X = S.random_states(rb, B)
gm.iterate(X); X = gm.xs[:,1,:].copy()
gm.set_profiling(True); gm.reset_kernel_times()
t0 = time.time()
for _ in range(steps):
    gm.iterate(X)
dt = (time.time()-t0)/steps
kt = gm.kernel_times()
print('NT=%s B=%d k=%d: %.1f ms/step  %.0f steps/s |' % (os.environ.get('SMPC_RICCATI_NT','256'), B, iters, dt*1e3, B/dt), ' '.join('%s %.2f' % (k, v[0]/max(1,v[1])) for k,v in kt.items()), '| finite', bool(np.isfinite(gm.info).all()))
if os.environ.get('SMPC_PHASE_PROFILE'):
    out = np.zeros(64); gm._lib.check(gm._lib.L.smpc_debug_get_phase_cycles(gm._h, out))
    names = ['c','-','AB_prefetch','build','Ptilde_out','Hprefetch','colpass','TG','products','Cphase','-','-','-','Pt_out']  # riccati_kino_body slots (a -DSMPC_KINO_RICCATI_PROF build)
    n = (steps+1)*iters*gm.H
    print('phase cycles per stage (block 0):', ' '.join('%s %.0f' % (nm, out[i]/n) for i,nm in enumerate(names)), '| total %.0f' % (out[:14].sum()/n))
    print('  sweep parts per stage (15 panels): gather %.0f  invert+U %.0f  operands+mfma %.0f  fixup %.0f' % tuple(out[36:40]/n))
if os.environ.get('SMPC_PHASE_PROFILE'):
    names2 = {15:'load',16:'FK',17:'S/I',18:'vel',19:'composite',20:'com/Ag',21:'GJ',22:'M1/Agbi',23:'a',24:'xnext',25:'acc',26:'Fc',27:'columns',28:'ab_d',29:'defect',30:'cost/cstr',31:'mult',32:'tables',33:'AB',34:'QS',35:'CR'}
    nd = (steps+1)*iters
    print('deriv phase cycles (block 17):', ' '.join('%s %.0f' % (nm, out[i]/nd) for i,nm in names2.items()), '| total %.0f' % (out[15:36].sum()/nd))

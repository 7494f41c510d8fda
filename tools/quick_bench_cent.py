"""Quick timing of the centroidal control step on the GPU: python tools/quick_bench_cent.py [B] [iters] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gm, rb, _, _ = S.make_cent_product(B, max_iters=iters)
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.]))
X = S.random_states(rb, B)
gm.iterate(X)
prof = not os.environ.get('NOPROF')
gm.set_profiling(prof); gm.reset_kernel_times()
t0 = time.time()
for _ in range(steps):
    gm.iterate(X)
dt = (time.time() - t0) / steps
kt = gm.kernel_times()
print('centroidal B=%d k=%d: %.2f ms/step  %.0f steps/s |' % (B, iters, dt * 1e3, B / dt), ' '.join('%s %.3f' % (k, v[0] / max(1, v[1])) for k, v in kt.items() if k != '-'),
      '| finite', bool(np.isfinite(gm.info).all()), 'ls idx max', gm.info[:, 11].max())
if os.environ.get('SMPC_PHASE_PROFILE'):
    out = np.zeros(64); gm._lib.check(gm._lib.L.smpc_debug_get_phase_cycles(gm._h, out))
    names = ['inputs', 'point', 'small+AB', 'grad', 'knot+zero', 'M1', 'load', 'sweep1+ext', 'M2', 'mfma', 'sweep2+epi(bwd)', 'forward', 'reduce', 'linesearch', 'accept']
    n = (steps + 1) * iters
    print('cycles per iteration (block 0):', ' '.join('%s %.0f' % (nm, out[i] / n) for i, nm in enumerate(names)), '| total %.0f' % (out[:40].sum() / n))
    print('  one tick costs %.0f cycles per stage-iteration' % (out[15] / n / 50))
    print('  sweep parts per iteration: gather %.0f invert %.0f mfma %.0f fixup %.0f' % tuple(out[36:40] / n))

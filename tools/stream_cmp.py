"""stream vs tile hand-over of the derivative pass on the GPU: python tools/stream_cmp.py <batch> (diagnostic)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np, mpc_setup as S, oracle_lib as O
    B = int(sys.argv[1])
    gm, rb, _, _ = S.make_product(B, max_iters=int(os.environ.get("ITERS", "3")), horizon=20)
    gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.3, 0.05, 0, 0, 0, 0.1]))
    X = S.random_states(rb, B, seed=5)
    outs = []
    for _ in range(int(os.environ.get("STEPS", "3"))):
        gm.iterate(X); X = gm.xs[:, 1, :].copy(); outs.append(gm.xs.copy())
    np.save(sys.argv[2], np.stack(outs))
else:
    import numpy as np
    B = sys.argv[1]
    for v in ("0", "1"):
        subprocess.check_call([sys.executable, __file__, B, "/tmp/stream_%s.npy" % v], env=dict(os.environ, SMPC_LANE_STREAM=v))
    a, b = np.load("/tmp/stream_0.npy"), np.load("/tmp/stream_1.npy")
    print("finite", np.isfinite(b).all(), "identical", np.array_equal(a, b))
    for s in range(a.shape[0]):
        d = np.abs(a[s] - b[s]).max(axis=(1, 2))
        bad = np.nonzero(d > 0)[0]
        print("step", s, "max diff", d.max(), "instances differing:", bad[:20], "of", len(bad))

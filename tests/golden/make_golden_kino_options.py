#!/usr/bin/env python3
"""Golden closed loops of the optional constraint blocks of the kinodynamics OCP (tests/golden/go2_kino_options_golden.npz), produced by
the CPU oracle in the build container (the reference cannot be built or imported here):

    python tests/golden/make_golden_kino_options.py

H = 20 so that the CPU tier replays them in seconds: (tc) terminal DCM constraint, gait of record, k = 2, 8 control steps;
(cone) friction cones mu = 0.1 under a sideways, turning command, k = 2, 8 steps; (land) land rows on the short trot 4 / 8 / 4 / 8,
k = 2, 18 steps (the first landing stage enters the horizon at step 13).  Inputs and expected outputs only, no code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402

CASES = {
    "tc": dict(so=None, mo={"terminal_constraint": True}, cycle="trot", walk=(0.2, 0, 0, 0, 0, 0.0), steps=8),
    "cone": dict(so={"force_cone": True, "mu": 0.1}, mo=None, cycle="trot", walk=(0.6, 0.4, 0, 0, 0, 0.5), steps=8),
    "land": dict(so={"land_cstr": True}, mo={"T_fly": 8, "T_contact": 4}, cycle="short", walk=(0.3, 0, 0, 0, 0, 0.1), steps=18),
}


def cycle_of(name):
    if name == "trot":
        return O.trot_cycle()
    cs = np.ones((24, 4), np.uint8)
    cs[4:12, [0, 3]] = 0
    cs[16:24, [1, 2]] = 0
    return cs


def main():
    out = {}
    for tag, c in CASES.items():
        om, rb, _ = S.make_oracle(2, 2, 20, settings_override=c["so"], mpc_override=c["mo"])
        om.generateCycleHorizon(cycle_of(c["cycle"]))
        om.switchToWalk(np.array(c["walk"], float))
        X = S.random_states(rb, 2)
        out[tag + "_X0"] = X
        for _ in range(c["steps"]):
            om.iterate(X)
            X = om.xs[:, 1, :].copy()
        out[tag + "_xs"], out[tag + "_us"], out[tag + "_alpha"], out[tag + "_vs"] = om.xs, om.us, om.info[:, 2], om.vs
    np.savez_compressed(os.path.join(HERE, "go2_kino_options_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

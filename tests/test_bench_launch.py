"""`bench.py --gpus N` must really start N ranks (the parent touches no GPU, spawns torch.distributed.run as a child and
relays rank 0's line).  Rehearsed on CPU with --dry-run: gloo + the test-only emulation build of the kernel bodies."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1"] + args,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks(built):
    one = _run([])
    two = _run(["--gpus", "2"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["global_batch"] == 2 * two["config"]["batch_per_gpu"]
    for o in (one, two):
        assert o["scaling"] == "weak" and o["steps"] == 2 and o["warmup"] == 1 and o["config"]["finite"]
        assert "DRY RUN" in o["data"]
    # N > 1: the return set is exchanged inside the timed region (rank 0 holds every rank's rows), and a run that could not do it does not
    # report a value at all (bench.py exits non-zero at the first warm-up step)
    g = two["gather"]
    assert g["in_timed_region"] is True and "error" not in g and g["rows_ok"] is True


def test_default_batch_is_the_baseline_configuration():
    sys.path.insert(0, ROOT)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "8192 if args.gpus >= 8 else 4096" in src  # 8 x 8192 = 65536 (BASELINE configs[4]); 4096 on one GPU (configs[2])
    assert 'default=100' in src and 'default=20' in src  # >= 100 timed steps after 20 warm-up steps

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
    # the driver parses the LAST stdout line: it must be that object, and short (round 5 printed 24 KB and was not parsed)
    last = p.stdout.rstrip("\n").splitlines()[-1]
    assert last == lines[0] and len(last) < 4096, len(last)
    return json.loads(last)


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


def test_the_printed_line_stays_compact():
    """compact_line on the largest line this repo has produced (round 5's 24 KB object): under the limit, every contract key kept, the
    dominant kernel's roofline flat, the cpu_baseline present."""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05g_bench.json")))
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT <= 3072 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "kernel_ms", "cfg2_centroidal_steps_per_s", "cfg4_talos_fulldynamics_steps_per_s"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    rl = line["roofline"]
    assert set(rl) <= {"bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "kernel", "traffic", "traffic_source", "peak_measured"}
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-4 and not any(isinstance(v, dict) for v in rl.values())
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    # a line that would not fit sheds its optional blocks, never the contract keys
    fat = dict(full, kernel_ms={("k%d" % i): 1.0 for i in range(400)})
    slim = json.loads(bench.compact_line(fat))
    assert "kernel_ms" not in slim and "roofline" in slim and "cpu_baseline" in slim

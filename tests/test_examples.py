"""The example scripts run end to end: on the CPU test build of the kernel bodies here (short horizon), on the HIP library in the GPU tier."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, emu, env_extra=None):
    src = open(os.path.join(ROOT, "examples", script)).read()
    if emu:
        assert "LIB = None" in src
        src = src.replace("LIB = None", "LIB = __import__('mpc_setup').emu_lib()")
    code = "import sys; sys.path.insert(0, %r); sys.argv = ['x'] + %r; __file__ = %r\n" % (
        os.path.join(ROOT, "tests"), [str(a) for a in args], os.path.join(ROOT, "examples", script)) + src
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


def test_talos_kinodynamics_id_stack_on_the_cpu_build(built):
    out = _run("talos_kinodynamics_id_batched.py", [2, 2], True, {"SMPC_EXAMPLE_HORIZON": "12"})
    assert "flat-foot inverse dynamics" in out and "895 N of 895 N weight" in out


@pytest.mark.gpu
def test_talos_kinodynamics_id_stack(built):
    out = _run("talos_kinodynamics_id_batched.py", [16, 120], False)  # 1.2 s: double support, then the whole first swing of the left foot
    assert "16 bipeds" in out and "simulated robots: base height" in out


def test_talos_centroidal_id_stack_on_the_cpu_build(built):
    out = _run("talos_centroidal_id_batched.py", [2, 2], True, {"SMPC_EXAMPLE_HORIZON": "12"})
    assert "flat-foot inverse dynamics" in out and "base height 1.026 .. 1.026" in out


@pytest.mark.gpu
def test_talos_centroidal_id_stack(built):
    out = _run("talos_centroidal_id_batched.py", [16, 120], False)
    assert "16 bipeds" in out

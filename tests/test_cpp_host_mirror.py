"""The C++ host mirror (include/simple-mpc/batched-mpc.hpp over the C ABI) exercised by a C++ program written like the
reference's tests/mpc.cpp (sizes, foot-timing known answers, iterate).  CPU tier: linked against the sequential-lane test
build of the kernel bodies; GPU tier: against libsmpc_hip.so."""
import os
import subprocess

import pytest

import mpc_setup as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_kat.cpp")
HIP_LIB = os.path.join(ROOT, "simple-mpc_amd", "csrc", "libsmpc_hip.so")


def _build_and_run(lib_path, exe):
    libdir, libname = os.path.split(lib_path)
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + libdir, "-l:" + libname,
           "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror KAT: OK" in out.stdout


def test_cpp_program_against_the_cpu_test_build(built, tmp_path):
    S.emu_lib()
    _build_and_run(S.EMU_LIB, str(tmp_path / "host_mirror_emu"))


@pytest.mark.gpu
def test_cpp_program_against_the_hip_library(built, tmp_path):
    _build_and_run(HIP_LIB, str(tmp_path / "host_mirror_hip"))

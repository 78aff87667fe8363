"""The C ABI from C: compile examples/fir_mc_demo.c against include/*.h with gcc, link libllzfilter_hip.so, run it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    exe = os.path.join(str(tmp_path), "fir_mc_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "fir_mc_demo.c"),
                           "-L" + os.path.join(ROOT, "llzlab_amd"), "-lllzfilter_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "llzlab_amd"), "-lm", "-o", exe])
    return exe


def test_c_caller_compiles_and_links(tmp_path):
    """headers are valid C99 and every symbol the example uses resolves (no GPU needed to link)"""
    from llzlab_amd import capi
    capi.build()
    assert os.path.exists(_compile(tmp_path))


@pytest.mark.gpu
def test_c_caller_runs_on_gpu(tmp_path):
    exe = _compile(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout

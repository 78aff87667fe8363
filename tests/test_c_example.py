"""The C ABI from C: compile examples/fir_mc_demo.c against include/*.h with gcc, link libllzfilter_hip.so, run it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path, name="fir_mc_demo"):
    exe = os.path.join(str(tmp_path), name)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"),
                           "-L" + os.path.join(ROOT, "llzlab_amd"), "-lllzfilter_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "llzlab_amd"), "-lm", "-o", exe])
    return exe


def test_c_caller_compiles_and_links(tmp_path):
    """headers are valid C99 and every symbol the example uses resolves (no GPU needed to link)"""
    from llzlab_amd import capi
    capi.build()
    assert os.path.exists(_compile(tmp_path))
    assert os.path.exists(_compile(tmp_path, "sharded_demo"))


@pytest.mark.gpu
def test_c_caller_runs_on_gpu(tmp_path):
    exe = _compile(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


@pytest.mark.gpu
def test_c_sharded_caller_runs_on_gpu(tmp_path):
    """include/llz_shard.h from plain C: 4 and 1 shard(s) on the box's device(s), sharded output identical to one handle"""
    exe = _compile(tmp_path, "sharded_demo")
    for shards in ("4", "1"):
        out = subprocess.run([exe, shards], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "identical to one handle" in out.stdout and out.stdout.strip().endswith("OK")

"""SURVEY section 5, row 2 (race detection / sanitizers): the host side of librgcn.so under AddressSanitizer and
UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the GPU pool, so what is checked is the code that runs on
the CPU: csrc/sampler.hip, the O(log V) port of the reference's sample_edge_neighborhood (code/train.py:161-198) -- pure
host C++ behind the C ABI of include/rgcn.h, compiled here with g++ instead of hipcc."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_sampler_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sampler_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-x", "c++", os.path.join(ROOT, "relationprediction_amd", "csrc", "sampler.hip"),
           os.path.join(ROOT, "tests", "sanitize", "sampler_driver.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "sampler_driver: ok" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]

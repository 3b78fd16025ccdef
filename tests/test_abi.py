"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/rgcn.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from relationprediction_amd import build
    return build.build()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rgcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgcn_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ["rgcn_create", "rgcn_destroy", "rgcn_set_graph", "rgcn_forward", "rgcn_backward",
                 "rgcn_get_grad", "rgcn_set_param", "rgcn_comm_init", "rgcn_last_error"]:
        assert must in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for s in declared_symbols():
        assert hasattr(lib, s), "librgcn.so does not export %s" % s
    lib.rgcn_abi_version.restype = ctypes.c_int32
    assert lib.rgcn_abi_version() == 1


def test_binding_matches_header(lib_path):
    from relationprediction_amd import _native
    assert _native.exported_symbols() == declared_symbols()
    _native.load_library()          # attaches prototypes; raises if a symbol is missing
    assert ctypes.sizeof(_native.RgcnConfig) == 64


def test_experiment_hooks_live_in_the_devtools_build_only(lib_path):
    """the product library exports nothing but include/rgcn.h; the stand-alone GEMM entry points of
    include/rgcn_devtools.h exist in librgcn_devtools.so (-DRGCN_DEVTOOLS) only, and no ablation switch is left in
    the product sources"""
    from relationprediction_amd import _native, build
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rgcn_devtools.h")).read(), flags=re.S)
    dev = sorted(set(re.findall(r"\b(rgcn_[a-z_0-9]+)\s*\(", text)))
    assert dev == _native.exported_devtools_symbols() and dev
    product, devtools = ctypes.CDLL(lib_path), ctypes.CDLL(build.LIB_DEVTOOLS)
    for s in dev:
        assert not hasattr(product, s), "product library exports %s" % s
        assert hasattr(devtools, s)
    for s in declared_symbols():
        assert hasattr(devtools, s)
    for name in os.listdir(build.CSRC):
        src = open(os.path.join(build.CSRC, name)).read()
        assert "RGCN_GEMM_ABLATE" not in src and "ablate" not in src, name


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "rgcn.h"\n#include "rgcn_devtools.h"\nint main(void){ rgcn_config c; (void)c; return sizeof(rgcn_config) == 64 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_no_gpu_means_loud_failure(lib_path):
    """Without a GPU the product path must raise, not fall back to anything."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    from relationprediction_amd import _native
    with pytest.raises(_native.RgcnError):
        _native.Engine(16, 9, 10, 1, "block", 2, max_edges=50)


def test_library_has_no_packed_fp32_instructions():
    """gfx950: packed-FP32 VALU ops return wrong low halves beside waves that issue bf16 MFMAs
    (tools/mfma_corun.hip); the build switches the feature off and this checks the linked code objects."""
    from relationprediction_amd import build
    build.build()
    assert build.check_no_packed_fp32()


def test_bench_workloads_are_backed_by_fixtures_and_fail_loudly_without_a_gpu():
    """every bench.py workload names a graph in tests/golden/graphs.npz of the stated size and a legal geometry;
    on a box without a GPU the bench exits non-zero with the library's error instead of falling back to anything"""
    import importlib.util
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name, (graph, V, R, d, L, kind, nb, E) in bench.WORKLOADS.items():
        t = bench.load_graph(graph)        # a fixture of tests/golden/graphs.npz, or drawn from one's histograms
        assert t.shape == (E, 3) and t.dtype == np.int32, name
        assert t[:, [0, 2]].max() < V and t[:, 1].max() < R and t.min() >= 0, name
        if graph.startswith("synth:"):
            assert len(np.unique(t, axis=0)) == E, name
            assert np.array_equal(t, bench.load_graph(graph)), name      # deterministic
        if kind == "block":
            assert d % nb == 0 and d // nb in (1, 2, 3, 4, 5, 8), name
    assert set(bench.EXTRA_WORKLOADS) <= set(bench.WORKLOADS)
    if not os.path.exists("/dev/kfd"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--cpu-steps", "0"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "librgcn status" in r.stderr and not r.stdout.strip()

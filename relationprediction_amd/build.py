"""Build librgcn.so (hand-written HIP for gfx950) in-tree.

    python -m relationprediction_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  Objects go to build/, the library to
relationprediction_amd/lib/librgcn.so (git-ignored, but it travels to the GPU box).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(ROOT, "build", "obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librgcn.so")
ARCH = "gfx950"
SOURCES = ["rgcn_api.hip", "graph_prep.hip", "gemm_f32.hip", "block_msgs.hip", "basis.hip",
           "elementwise.hip", "decoder.hip", "optimizer.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "rgcn_internal.h"), os.path.join(ROOT, "include", "rgcn.h")]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the R-GCN HIP library cannot be built")


def _digest(paths):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([sp] + HEADERS)
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp)
                 and open(stamp).read() == dig)
        jobs.append((sp, obj, stamp, dig, fresh))

    def compile_one(job):
        sp, obj, stamp, dig, fresh = job
        if fresh:
            return None
        cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (sp, r.stderr[-4000:]))
        with open(stamp, "w") as f:
            f.write(dig)
        return r.stderr if verbose else None

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        outs = list(ex.map(compile_one, jobs))
    for o in outs:
        if o:
            sys.stderr.write(o)
    if force or not os.path.exists(LIB) or any(not j[4] for j in jobs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + [j[1] for j in jobs] + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))

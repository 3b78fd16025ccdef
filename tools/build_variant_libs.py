#!/usr/bin/env python
"""Throw-away variant libraries for timing experiments: ONE translation unit recompiled with extra -D flags and linked
with the cached objects of the product build.  The variants go to tools/experiments/_libs/ (git-ignored; they travel to
the GPU box), where a script copies one over relationprediction_amd/lib/librgcn.so of its scratch tree.
Usage: python tools/build_variant_libs.py decoder.hip NAME=-DFOO=1 NAME2="-DFOO=2 -DBAR" ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relationprediction_amd import build  # noqa: E402

build.build()
tu = sys.argv[1]
out = os.path.join(ROOT, "tools", "experiments", "_libs")
os.makedirs(out, exist_ok=True)
objs = [os.path.join(build.OBJDIR, s.replace(".hip", ".o")) for s in build.SOURCES]
vs = os.path.join(build.OBJDIR, "exports.map")
for spec in sys.argv[2:]:
    name, flags = spec.split("=", 1)
    obj = os.path.join(out, name + ".o")
    subprocess.check_call([build._hipcc()] + build.FLAGS + flags.split() + ["-c", os.path.join(build.CSRC, tu), "-o", obj],
                          stderr=subprocess.DEVNULL)
    lib = os.path.join(out, "librgcn_%s.so" % name)
    link = [o if os.path.basename(o) != tu.replace(".hip", ".o") else obj for o in objs]
    subprocess.check_call([build._hipcc(), "--offload-arch=" + build.ARCH, "-shared", "-fPIC",
                           "-Wl,--version-script=" + vs, "-o", lib] + link + ["-ldl"])
    os.remove(obj)
    print(lib)

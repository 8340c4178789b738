"""In-tree build of libsurya_amd.so for gfx950 (hipcc cross-compiles without a GPU).

`python -m surya_amd.build` or `__graft_entry__.build()`. The .so stays next to this file so it travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsurya_amd.so")
SOURCES = ["rec_model.hip", "det_model.hip", "layout_model.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "surya_amd.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    # experiment knobs: extra compiler flags (-DSA_...=1) and an alternative output path, loaded through SURYA_AMD_LIB
    extra = os.environ.get("SURYA_AMD_CXXFLAGS", "").split()
    out = os.environ.get("SURYA_AMD_LIB_OUT", LIB)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), *extra]
    # one object per translation unit, compiled side by side (the two model files take ~40 s each), then one link
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    import zlib
    tag = "%08x" % zlib.crc32((out + " ".join(extra)).encode())    # per output / flag set: A/B builds do not share objects
    objs, procs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + "." + tag + ".o")
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        if verbose:
            print("[surya_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out + ".tmp"]
    if verbose:
        print("[surya_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

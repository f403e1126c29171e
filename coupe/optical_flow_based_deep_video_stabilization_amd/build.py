"""In-tree build of libvstab_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvstab_hip.so")
SOURCES = ("conv_mfma.hip", "conv_rowwin.hip", "flow_ops.hip", "sampler_ops.hip", "nldf_ops.hip", "clip_ops.hip", "pack.cpp", "api.cpp", "nldf_api.cpp")
HEADERS = ("vstab_internal.h", "api_internal.h", os.path.join("..", "..", "..", "include", "vstab.h"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    objs = []
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(obj_dir, s + ".o")
        cmd = [hipcc()] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    tmp = LIB + ".tmp"
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp])
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

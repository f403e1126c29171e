"""In-tree build of libvstab_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvstab_hip.so")
SOURCES = ("conv_mfma.hip", "conv_rowwin.hip", "conv_skinny.hip", "tap_panel.hip", "wino_gemm_stream.hip", "flow_ops.hip", "sampler_ops.hip", "nldf_ops.hip", "clip_ops.hip", "train_ops.hip", "wgrad_mfma.hip", "winograd_ops.hip", "homography_ops.hip", "pack.cpp", "api.cpp", "nldf_api.cpp", "train_api.cpp")
HEADERS = ("vstab_internal.h", "api_internal.h", "hbm_profile.h", "conv_kloop_gfx950.inc", os.path.join("..", "..", "..", "include", "vstab.h"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


STAMP = os.path.join(HERE, "libvstab_hip.so.srchash")


def source_hash() -> str:
    """Content hash of everything the library is built from (mtimes do not survive the copy to a GPU box)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for dep in [os.path.join(CSRC, s) for s in SOURCES + HEADERS]:
        with open(dep, "rb") as f:
            h.update(os.path.basename(dep).encode() + b"\0" + f.read())
    return h.hexdigest()


def stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    # one builder at a time: several ranks of a torchrun job may import the package at once
    import fcntl
    lock = open(os.path.join(HERE, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not stale():          # another process built it while we waited
            return LIB
        return _build_locked(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(verbose: bool) -> str:
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
    # -z defs: an unresolved symbol fails the link here, not the dlopen on the GPU box (hipcc can silently drop the host
    # stub of a kernel template whose body it could not digest in the host pass)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs"] + objs + ["-o", tmp])
    os.replace(tmp, LIB)
    with open(STAMP + ".tmp", "w") as f:
        f.write(source_hash())
    os.replace(STAMP + ".tmp", STAMP)
    return LIB


# ---- sanitizer build of the HOST side (sanitizers run on the CPU build only): pack.cpp and the planning /
# validation code of the three API files compiled with AddressSanitizer + UndefinedBehaviorSanitizer, linked with the regular
# objects of the kernels.  tests/test_host_sanitize.py drives it with the vstab_host_* CPU tests.
ASAN_LIB = os.path.join(HERE, "libvstab_hip_asan.so")
ASAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"]


def asan_runtime() -> str:
    out = subprocess.check_output([hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not os.path.isabs(out) or not os.path.exists(out):
        import glob
        cands = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        if not cands:
            raise RuntimeError("AddressSanitizer runtime of the ROCm clang not found")
        out = sorted(cands)[-1]
    return out


def build_sanitized(verbose: bool = False) -> str:
    build()                                             # the kernels' regular objects
    stamp = ASAN_LIB + ".srchash"
    want = source_hash() + "+asan"
    if os.path.exists(ASAN_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return ASAN_LIB
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)                 # build() may have been satisfied by its stamp alone
    objs = []
    for s in SOURCES:
        if s.endswith(".cpp"):
            o = os.path.join(obj_dir, s + ".asan.o")
            cmd = [hipcc()] + [f for f in FLAGS if f != "-O3"] + ASAN_FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        else:
            o = os.path.join(obj_dir, s + ".o")
            if not os.path.exists(o):                   # build() was satisfied by its stamp but the objects are gone
                subprocess.check_call([hipcc()] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan"]
                          + objs + ["-o", ASAN_LIB + ".tmp"])
    os.replace(ASAN_LIB + ".tmp", ASAN_LIB)
    with open(stamp, "w") as f:
        f.write(want)
    return ASAN_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

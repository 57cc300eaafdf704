"""Build libscl_hip.so (the gfx950 kernel library) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so sits next to this file so that
it travels with the repo snapshot to the GPU box.  Objects are rebuilt only when a source or
header is newer than the object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SCL_BUILD_TAG=<tag>: an A/B build (usually with SCL_BUILD_DEFINES) into build_<tag>/, library included — the shipped
# libscl_hip.so next to this file is left alone; run a tool against it with LD_LIBRARY_PATH / SCL_LIB_PATH
TAG = os.environ.get("SCL_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build_" + TAG if TAG else "build")
LIB = os.path.join(OBJ if TAG else HERE, "libscl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
         "-fno-gpu-rdc"] + os.environ.get("SCL_BUILD_DEFINES", "").split()      # extra -D switches for A/B builds of one kernel (same-box comparisons)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "scl_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build_library(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("[scl build] %s (%d objects, rebuilt=%s)" % (LIB, len(objs), rebuilt))
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)

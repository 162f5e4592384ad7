"""Build the HIP shared library of hypre_amd in-tree (hipcc, gfx950 only).

    python -m hypre_amd.build            # incremental
    python -m hypre_amd.build --force

Produces hypre_amd/lib/libhypre_amd.so (git-ignored; it travels to the GPU box
with the repo snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libhypre_amd.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-I" + INC, "-I" + CSRC,
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")))


def _newest_header():
    t = 0.0
    for d in (CSRC, INC):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, force):
    obj = os.path.join(OBJDIR, src + ".o")
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(sp)
            and os.path.getmtime(obj) > _newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fopenmp", "-o", LIB] + objs + ["-lrccl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built" if changed else "up to date", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)

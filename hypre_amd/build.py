"""Build the HIP shared library of hypre_amd in-tree (hipcc, gfx950 only).

    python -m hypre_amd.build            # incremental
    python -m hypre_amd.build --force

Produces hypre_amd/lib/libhypre_amd.so (git-ignored; it travels to the GPU box
with the repo snapshot).  hipcc cross-compiles without a GPU.
"""
import json
import os
import re
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
    kernels = src.endswith(".hip")
    if kernels:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    err = r.stderr
    if kernels:
        # registers / scratch / occupancy of every kernel, kept beside the object for
        # tests/test_abi.py::test_hot_kernels_keep_full_occupancy
        rows, cur = [], None
        for line in err.splitlines():
            m = re.search(r"remark:\s+(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = {"name": m.group(2)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(1).split(" [")[0]] = int(m.group(2))
        with open(obj + ".resources.json", "w") as fh:
            json.dump(rows, fh)
        err = "\n".join(l for l in err.splitlines()
                        if "kernel-resource-usage" not in l and not re.match(r"^\s*\d+ \| |^\s*\| *\^", l))
        err = re.sub(r"\d+ warnings? generated when compiling for gfx950\.\n?", "", err + "\n") if "warning:" not in err else err
    if err.strip():
        sys.stderr.write(err)
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
    # one report for the whole library
    report = []
    for o in objs:
        if os.path.exists(o + ".resources.json"):
            with open(o + ".resources.json") as fh:
                report += json.load(fh)
    with open(os.path.join(LIBDIR, "kernel_resources.json"), "w") as fh:
        json.dump(report, fh, indent=0)
    if verbose:
        print("built" if changed else "up to date", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)

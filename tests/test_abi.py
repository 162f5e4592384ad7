"""CPU: the C-ABI library loads, exports every function include/*.h declares, and the
ctypes prototype tables cover exactly that set (no compute call is made)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    pat = re.compile(r"^\s*(?:[A-Za-z_][\w\s\*]*?)\b((?:hypre|HYPRE|hypreDevice|Generate)\w*)\s*\(", re.M)
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith("#"))
        txt = re.sub(r"typedef\s[^;{]*\(\s*\*[^;]*;", "", txt)          # function-pointer typedefs
        txt = re.sub(r"typedef\s+struct[^{;]*\{.*?\}[^;]*;", "", txt, flags=re.S)   # struct bodies (callback members)
        for m in pat.finditer(txt):
            name = m.group(1)
            if name.startswith("HYPRE_PtrTo"):
                continue
            names.add(name)
    return names


def test_every_declared_symbol_is_exported_and_bound(lib):
    from hypre_amd import binding, parcsr_ls_binding
    declared = _declared()
    assert len(declared) > 150
    bound = set(binding.PROTOTYPES) | set(parcsr_ls_binding.PROTOTYPES)
    missing_export = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing_export, missing_export
    assert not (declared - bound), sorted(declared - bound)
    assert not (bound - declared), sorted(bound - declared)


def test_struct_sizes_match_headers(lib):
    """The ctypes mirrors must have the C layout (LP64)."""
    import subprocess
    import tempfile
    from hypre_amd import binding as B
    # the headers must be plain C: compile a probe with gcc and compare layouts
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "hypre_amd_parcsr_ls.h"
#include "hypre_amd_IJ_mv.h"
int main(void) {
  printf("%zu %zu %zu %zu ", sizeof(hypre_IJMatrix), offsetof(hypre_IJMatrix, global_first_row),
         sizeof(hypre_IJVector), offsetof(hypre_IJVector, global_first_row));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(hypre_CSRMatrix), sizeof(hypre_Vector),
         sizeof(hypre_ParCSRMatrix), sizeof(hypre_ParVector), sizeof(hypre_ParCSRCommPkg),
         offsetof(hypre_ParCSRMatrix, comm_pkg), offsetof(hypre_ParVector, all_zeros),
         sizeof(hypre_amd_CommOps));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "probe.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(td, "probe.c"), "-o", os.path.join(td, "probe")], check=True)
        out = subprocess.run([os.path.join(td, "probe")], capture_output=True, text=True, check=True).stdout.split()
    ij, out = [int(v) for v in out[:4]], out[4:]
    assert ij == [C.sizeof(B.IJMatrix), B.IJMatrix.global_first_row.offset,
                  C.sizeof(B.IJVector), B.IJVector.global_first_row.offset]
    sizes = [int(v) for v in out]
    assert sizes[0] == C.sizeof(B.CSRMatrix)
    assert sizes[1] == C.sizeof(B.Vector)
    assert sizes[2] == C.sizeof(B.ParCSRMatrix)
    assert sizes[3] == C.sizeof(B.ParVector)
    assert sizes[4] == C.sizeof(B.CommPkg)
    assert sizes[5] == B.ParCSRMatrix.comm_pkg.offset
    assert sizes[6] == B.ParVector.all_zeros.offset
    assert sizes[7] == C.sizeof(B.CommOps)
    m = lib.hypre_CSRMatrixCreate(3, 4, 5)
    assert (m.contents.num_rows, m.contents.num_cols, m.contents.num_nonzeros, m.contents.owns_data) == (3, 4, 5, 1)
    lib.hypre_CSRMatrixDestroy(m)
    v = lib.hypre_ParVectorCreate(0, 10, None)
    assert v.contents.global_size == 10 and v.contents.local_vector.contents.size == 10
    lib.hypre_ParVectorDestroy(v)


def test_error_word_convention(lib):
    from hypre_amd import binding as B
    lib.HYPRE_ClearAllErrors()
    assert lib.HYPRE_BoomerAMGSolve(None, None, None, None) != 0       # arg 2 (A) is checked first
    assert lib.HYPRE_GetError() & 4 and lib.HYPRE_GetErrorArg() == 2
    lib.HYPRE_ClearAllErrors()
    s = C.c_void_p()
    lib.HYPRE_BoomerAMGCreate(C.byref(s))
    lib.HYPRE_BoomerAMGSetNumSweeps(s, 0)                               # invalid -> arg 2
    assert lib.HYPRE_GetErrorArg() == 2
    lib.HYPRE_ClearAllErrors()
    lib.HYPRE_BoomerAMGDestroy(s)


def test_hot_kernels_keep_full_occupancy():
    """Build-time guard: every tiled SpMV variant must fit 8 waves per SIMD (<= 64 VGPRs) with no scratch —
    one innocent-looking loop once cost the family 35 registers and 11 % of its bandwidth."""
    import json
    path = os.path.join(ROOT, "hypre_amd", "lib", "kernel_resources.json")
    if not os.path.exists(path):
        pytest.skip("library not built by hypre_amd/build.py in this tree")
    rows = json.load(open(path))
    tiled = [r for r in rows if "spmv_tiled_kernel" in r["name"]]
    assert len(tiled) >= 16
    for r in tiled:
        assert r["VGPRs"] <= 64 and r["Occupancy"] == 8, r
    for r in rows:
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, r


_listing_cache = {}


def _device_assembly(extra_flags):
    """device assembly of spmv_kernels.hip as the build's flags compile it (one compile per set of extra flags)"""
    import subprocess
    import tempfile
    key = tuple(extra_flags)
    if key not in _listing_cache:
        src = os.path.join(ROOT, "hypre_amd", "csrc", "spmv_kernels.hip")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-I" + os.path.join(ROOT, "include"),
                   "-I" + os.path.join(ROOT, "hypre_amd", "csrc"), "-x", "hip", "-S", "--cuda-device-only", "-o", out, src] + list(extra_flags)
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            _listing_cache[key] = open(out).read().splitlines()
    return _listing_cache[key]


def _xs_kernel_listing(extra_flags, vf=0, symbol=None):
    """instructions of spmv_xs_kernel<OP_AXPBY, value form vf (0 fp64, 2 one-byte codes), no fill> — or of the kernel whose
    mangled name starts with `symbol` — as the build's flags compile it (device code only)"""
    lines = _device_assembly(extra_flags)
    symbol = symbol or "_ZN4hamd14spmv_xs_kernelILi0ELi%dELb0E" % vf
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol) and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return [l.strip() for l in lines[start:end + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def _xs_schedule(ins):
    """where the loads of one tile sit: indices of the first matrix-stream load, of the wait that ends the scalar batch
    with the tile's piece descriptors, of the first LDS-DMA load of x, of the first barrier; and the waits between them"""
    first_stream = next(i for i, l in enumerate(ins) if l.startswith(("global_load_dwordx4", "global_load_dwordx2", "global_load_dword ")))
    first_dma = next(i for i, l in enumerate(ins) if l.startswith("global_load_lds_dwordx4"))
    # the descriptors are the scalar loads at a register offset (tile * 96 ints into the plan's table)
    import re
    pat = re.compile(r"s_load_dwordx8 s\[\d+:\d+\], s\[\d+:\d+\], s\d+ offset:")
    desc = [i for i, l in enumerate(ins[:first_dma]) if pat.match(l)]
    assert desc, "no descriptor batch found in front of the LDS-DMA loads"
    batch_wait = next(i for i in range(desc[-1], len(ins)) if ins[i].startswith("s_waitcnt") and "lgkmcnt(0)" in ins[i])
    first_barrier = next(i for i in range(first_dma, len(ins)) if ins[i].startswith("s_barrier"))
    return dict(first_stream=first_stream, batch_wait=batch_wait, first_dma=first_dma, first_barrier=first_barrier,
                vm_waits_before_dma=[l for l in ins[first_stream:first_dma] if l.startswith("s_waitcnt") and "vmcnt" in l],
                vm_waits_dma_to_barrier=[l for l in ins[first_dma:first_barrier] if l.startswith("s_waitcnt") and "vmcnt" in l],
                stream_loads_before_batch_wait=sum(1 for l in ins[:batch_wait] if l.startswith(("global_load_dwordx4", "global_load_dwordx2", "global_load_dword "))),
                dma_loads=sum(1 for l in ins[first_dma:first_barrier] if l.startswith("global_load_lds_dwordx4")))


def test_the_x_staged_kernel_keeps_its_load_schedule():
    """One trip through the vector-memory pipeline per tile is what spmv_xs_kernel is built on: the
    (value, index) stream of the tile is requested BEFORE the scalar batch with the tile's bounds and piece descriptors
    has come back, nothing waits for a vector load until the twelve LDS-DMA loads of the x pieces are out, and exactly one
    wait for everything stands in front of the first barrier.  The compiler is free to sink loads below the branches
    that follow them — it did, until an empty asm with a memory clobber fenced the batch — so the build's own compile of the
    kernel is disassembled here and the order checked; the same check must FAIL for the -DXS_EARLY_STREAM=0 variant
    (stream loads behind the empty-tile test), or it checks nothing."""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    good = _xs_schedule(_xs_kernel_listing([]))
    assert good["first_stream"] < good["batch_wait"] < good["first_dma"] < good["first_barrier"], good
    assert good["stream_loads_before_batch_wait"] >= 6, good          # 4 x 16 bytes of values + 2 x 8 bytes of indices per lane
    assert good["vm_waits_before_dma"] == [], good
    assert good["dma_loads"] == 12, good
    assert len(good["vm_waits_dma_to_barrier"]) == 1 and "vmcnt(0)" in good["vm_waits_dma_to_barrier"][0], good
    late = _xs_schedule(_xs_kernel_listing(["-DXS_EARLY_STREAM=0"]))
    assert not (late["first_stream"] < late["batch_wait"]), late
    # the coded form (one-byte value codes and their table): 2 x 4 bytes of codes + 2 x 8 bytes of indices per lane, and the
    # table rides with the x pieces as a thirteenth LDS-DMA load
    coded = _xs_schedule(_xs_kernel_listing([], vf=2))
    assert coded["first_stream"] < coded["batch_wait"] < coded["first_dma"] < coded["first_barrier"], coded
    assert coded["stream_loads_before_batch_wait"] >= 4, coded
    assert coded["vm_waits_before_dma"] == [], coded
    assert coded["dma_loads"] == 13, coded
    assert len(coded["vm_waits_dma_to_barrier"]) == 1 and "vmcnt(0)" in coded["vm_waits_dma_to_barrier"][0], coded
    # the slice form (a lane per row, eight entries a lane): 2 words of codes + 4 of indices per lane before the batch's wait,
    # the same thirteen LDS-DMA loads, one wait
    sl = _xs_schedule(_xs_kernel_listing([], symbol="_ZN4hamd14spmv_sl_kernelILi0ELi1ELi8E"))
    assert sl["first_stream"] < sl["batch_wait"] < sl["first_dma"] < sl["first_barrier"], sl
    assert sl["stream_loads_before_batch_wait"] >= 6, sl
    assert sl["vm_waits_before_dma"] == [], sl
    assert sl["dma_loads"] == 13, sl
    assert len(sl["vm_waits_dma_to_barrier"]) == 1 and "vmcnt(0)" in sl["vm_waits_dma_to_barrier"][0], sl


def test_option_gates_accept_the_built_branch_and_refuse_the_rest(lib):
    """Setters for options whose other branches are not built: the supported value passes, anything else is an
    argument error with a message (never silently ignored)."""
    import ctypes as C
    s = C.c_void_p()
    lib.HYPRE_BoomerAMGCreate(C.byref(s))
    lib.HYPRE_ClearAllErrors()
    for name, ok, bad in (("AggNumLevels", 0, 1), ("Nodal", 0, 4), ("SeqThreshold", 0, 100), ("RAP2", 0, 1),
                          ("Restriction", 0, 1), ("SmoothNumLevels", 0, 3), ("Additive", -1, 0), ("MultAdditive", -1, 0),
                          ("Simple", -1, 0), ("Redundant", 0, 1)):
        fn = getattr(lib, "HYPRE_BoomerAMGSet" + name)
        assert fn(s, ok) == 0 and lib.HYPRE_GetError() == 0, name
        fn(s, bad)
        assert lib.HYPRE_GetError() & 4 and lib.HYPRE_GetErrorArg() == 2, name
        assert name.encode() in lib.hypre_amd_LastErrorMessage()
        lib.HYPRE_ClearAllErrors()
    for name in ("NonGalerkinTol", "ADropTol"):
        fn = getattr(lib, "HYPRE_BoomerAMGSet" + name)
        assert fn(s, 0.0) == 0 and lib.HYPRE_GetError() == 0
        fn(s, 0.05)
        assert lib.HYPRE_GetError() & 4
        lib.HYPRE_ClearAllErrors()
    for name, v in (("MeasureType", 1), ("DebugFlag", 3), ("NumPaths", 2), ("SmoothType", 6), ("SmoothNumSweeps", 2)):
        assert getattr(lib, "HYPRE_BoomerAMGSet" + name)(s, v) == 0 and lib.HYPRE_GetError() == 0
    lib.HYPRE_BoomerAMGSetMeasureType(s, 5)
    assert lib.HYPRE_GetError() & 4
    lib.HYPRE_ClearAllErrors()
    lib.HYPRE_BoomerAMGDestroy(s)

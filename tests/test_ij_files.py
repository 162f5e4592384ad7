"""CPU: hypre's IJ text format (IJ_mv/IJMatrix.c:112-247, parcsr_mv/par_csr_matrix.c:888-1047,
IJ_mv/HYPRE_IJVector.c:641-782) through the library's HYPRE_IJMatrixRead / Print and
HYPRE_IJVectorRead / Print: the reference's own input files parse to the matrices their headers
describe, print -> read is the identity, and the reference's error behaviour is kept.  The multi-rank
files (ghost columns, entries sent to the owning rank, index ranges that do not start at 0) are covered
by the goldens that solve with them (tests/test_dist_golden.py: matrix.out.3, matrix.out.11,
solvers.out.404/405)."""
import ctypes as C
import os

import numpy as np
import pytest

from hypre_amd import binding as B, ij

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = os.path.join(HERE, "golden", "ij_files")


@pytest.fixture(scope="module")
def lib():
    return B.load_library()


def _diag_arrays(A):
    return B.csr_to_arrays(A.contents.diag)


def test_print_then_read_is_the_identity(lib, tmp_path):
    A = B.laplacian(5, 4, 3)                            # host-resident
    name = str(tmp_path / "lap")
    lib.hypre_ParCSRMatrixPrintIJ(A, 0, 0, name.encode())
    B.check()
    text = open(name + ".00000").read().splitlines()
    assert text[0] == "0 59 0 59"
    assert text[1] == "0 0 %.14e" % 6.0                 # the reference's "%b %b %.14e" line
    A2 = ij.read_matrix(name)
    for x, y in zip(_diag_arrays(A), _diag_arrays(A2)):
        assert np.array_equal(x, y)
    assert A2.contents.global_num_rows == 60 and A2.contents.offd.contents.num_nonzeros == 0
    # through the IJ shell as well, with a base-1 print read back (indices are relative to the first row)
    lib.hypre_ParCSRMatrixPrintIJ(A, 1, 1, (name + "1").encode())
    assert open(name + "1.00000").readline().split() == ["1", "60", "1", "60"]
    A3 = ij.read_matrix(name + "1")
    for x, y in zip(_diag_arrays(A), _diag_arrays(A3)):
        assert np.array_equal(x, y)
    shell = lib.hypre_amd_IJMatrixWrap(A3)
    lib.HYPRE_IJMatrixPrint(shell, (name + "2").encode())
    lib.HYPRE_IJMatrixDestroy(shell)                    # borrowed object: A3 stays alive
    assert open(name + "2.00000").read() == open(name + ".00000").read()
    for m in (A, A2, A3):
        lib.hypre_ParCSRMatrixDestroy(m)


def test_diagonal_moves_to_the_front_and_duplicates_overwrite(lib, tmp_path):
    name = str(tmp_path / "m")
    open(name + ".00000", "w").write("0 2 0 2\n0 1 -1.0\n0 0 4.0\n1 2 -2\n1 0 -3\n1 1 5e0\n2 2 1.0\n2 2 7.0\n2 0 0.5\n")
    A = ij.read_matrix(name)
    ii, jj, aa = _diag_arrays(A)
    assert list(ii) == [0, 2, 5, 7]
    assert list(jj) == [0, 1, 1, 2, 0, 2, 0]            # IJMatrix_parcsr.c:2803-2821: diagonal first, rest in file order
    assert list(aa) == [4.0, -1.0, 5.0, -2.0, -3.0, 7.0, 0.5]
    lib.hypre_ParCSRMatrixDestroy(A)


def test_vector_round_trip_and_reference_vector_file(lib, tmp_path):
    b = ij.read_vector(os.path.join(FILES, "b_tstoffd"))  # rank 0's file as a single-rank vector
    head = open(os.path.join(FILES, "b_tstoffd.00000")).readline().split()
    assert len(b) == int(head[1]) - int(head[0]) + 1
    v = B.parvec_from_numpy(np.linspace(-1.0, 2.0, 7), location=B.HYPRE_MEMORY_HOST)
    shell = lib.hypre_amd_IJVectorWrap(v)
    name = str(tmp_path / "v")
    lib.HYPRE_IJVectorPrint(shell, name.encode())
    lib.HYPRE_IJVectorDestroy(shell)
    lines = open(name + ".00000").read().splitlines()
    assert lines[0] == "0 6" and lines[1] == "0 %.14e" % -1.0
    assert np.allclose(ij.read_vector(name), np.linspace(-1.0, 2.0, 7), rtol=0, atol=1e-14)
    lib.hypre_ParVectorDestroy(v)


def test_reference_matrix_file_matches_its_header(lib):
    # rank 0's block of the 4-rank tucker matrix read as a 1-rank file: columns beyond the block are ghosts
    path = os.path.join(FILES, "data", "tucker21935", "IJ.A")
    head = [int(t) for t in open(path + ".00000").readline().split()]
    nlines = sum(1 for _ in open(path + ".00000")) - 1
    A = ij.read_matrix(path)
    m = A.contents
    assert m.diag.contents.num_rows == head[1] - head[0] + 1
    assert m.diag.contents.num_nonzeros + m.offd.contents.num_nonzeros == nlines
    cmap = np.ctypeslib.as_array(m.col_map_offd, shape=(max(m.offd.contents.num_cols, 1),))[:m.offd.contents.num_cols]
    assert np.all(np.diff(cmap) > 0) and (len(cmap) == 0 or cmap[0] > head[3])
    ii, jj, aa = _diag_arrays(A)
    assert np.array_equal(jj[ii[:-1]], np.arange(len(ii) - 1))       # diagonal entry first in every row
    lib.hypre_ParCSRMatrixDestroy(A)


def test_error_behaviour(lib, tmp_path):
    ijm = C.POINTER(B.IJMatrix)()
    lib.HYPRE_IJMatrixRead(str(tmp_path / "missing").encode(), 0, B.HYPRE_PARCSR, C.byref(ijm))
    flag = lib.HYPRE_GetError()
    assert flag & 4 and lib.HYPRE_GetErrorArg() == 1                 # hypre_error_in_arg(1), IJMatrix.c:138-142
    lib.HYPRE_ClearAllErrors()
    assert not ijm
    name = str(tmp_path / "bad")
    open(name + ".00000", "w").write("0 1 0 1\n0 0 1.0\n1 1\n")     # value missing
    lib.HYPRE_IJMatrixRead(name.encode(), 0, B.HYPRE_PARCSR, C.byref(ijm))
    assert lib.HYPRE_GetError() & 1                                  # HYPRE_ERROR_GENERIC, IJMatrix.c:205-209
    assert b"Error in IJ matrix input file." in lib.hypre_amd_LastErrorMessage()
    lib.HYPRE_ClearAllErrors()
    open(name + "v.00000", "w").write("0 1\n0 1.0\n2.5\n")           # a lone decimal is not "index value"
    ijv = C.POINTER(B.IJVector)()
    lib.HYPRE_IJVectorRead((name + "v").encode(), 0, B.HYPRE_PARCSR, C.byref(ijv))
    assert lib.HYPRE_GetError() & 1
    lib.HYPRE_ClearAllErrors()

"""The `ij`-compatible command line (hypre_amd/ij.py): the reference's own job lines
(test/TEST_ij/*.jobs, kept verbatim in tests/golden/ij_saved.json as `cmd`) are parsed into the same
options the golden tests use, and — on a GPU — replayed end to end through the driver, whose closing
lines are compared with the reference's `.saved` text."""
import json
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "ij_saved.json")))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_reference_job_line_parses_to_the_golden_options(name):
    from hypre_amd import ij
    case = GOLD[name]
    opt = ij.parse_cli(case["cmd"].split())
    ref = ij.IJOptions(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in case["options"].items()})
    for key, val in vars(ref).items():
        if key == "level_ow":
            continue            # `-owl 1.0 0` sets the default value explicitly; the golden options omit it
        assert getattr(opt, key) == val, (name, key, getattr(opt, key), val)


def test_out_of_scope_flags_are_refused():
    from hypre_amd import ij
    for bad in (["-agg_nl", "1"], ["-solver", "5"], ["-nc", "3"], ["-solver", "2", "-nc", "3", "-rhsrand"], ["-cljp"], ["-smtype", "6"], ["-interptype", "7"], ["-rlx", "5"], ["-rap", "1"],
                ["-w", "-10"], ["-owl", "-10", "0"], ["-rlx_coarse", "29"]):
        with pytest.raises(SystemExit):
            ij.parse_cli(bad)


def _replay(case, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    from conftest import free_port
    args = case["cmd"].split()
    if case["np"] == 1:
        cmd = [sys.executable, "-m", "hypre_amd.ij"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(case["np"]),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
               "-m", "hypre_amd.ij"] + args
    # jobs that read matrix files run where the reference's input files were copied (test/TEST_ij layout)
    cwd = os.path.join(HERE, "golden", "ij_files") if "-fromfile" in args else ROOT
    p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default.out.0", "fsai.out.103", "smoother.out.0", "smoother.out.3", "smoother.out.9",
                                  "smoother.out.11.1", "solvers.out.19", "solvers.out.23", "survey.C1", "smoother.out.13", "smoother.out.24",
                                  "coarsening.out.4", "interp.out.0", "matrix.out.0", "matrix.out.11", "solvers.out.405", "solvers.out.24", "solvers.out.28",
                                  "vector.out.B0", "vector.out.B7", "vector.out.B10", "vector.out.B100", "vector.out.B1", "vector.out.B101", "fsai.out.3", "solvers.out.3"])
def test_replay_reference_job_on_the_device(name):
    case = GOLD[name]
    out = _replay(case)
    exp = case["expect"]
    if "iterations" in exp:
        label = {1: "Iterations", 2: "Iterations", 3: "GMRES Iterations", 4: "GMRES Iterations"}.get(case["options"].get("solver", 0), "BoomerAMG Iterations")
        assert re.search(r"^%s = %d$" % (label, exp["iterations"]), out, re.M), out
        m = re.search(r"^Final %sRelative Residual Norm = (\S+)$" % ("GMRES " if label.startswith("GMRES") else ""), out, re.M)
        assert m and abs(float(m.group(1)) - exp["rel_resid"]) <= 1.5e-6 * exp["rel_resid"], out
    if "conv_factor" in exp:
        # the reference's own layout and precision (par_amg_solve.c:411-414)
        assert " Average Convergence Factor = %f" % exp["conv_factor"] in out, out
        assert "     Complexity:    grid = %f" % exp["grid"] in out, out
        assert "                operator = %f" % exp["operator"] in out, out
        assert "                   cycle = %f" % exp["cycle"] in out, out

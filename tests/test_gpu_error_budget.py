"""GPU test (-m gpu): the rounding-error budget, asserted.

oracle/libebm_oracle_ld.so is the oracle's own C source evaluated in 80-bit extended precision between
fp64 inputs and outputs — the same discrete model with ~2000x less rounding, i.e. a stand-in for its
exact evaluation.  From the same fp64 state the GPU, the fp64 oracle and the extended build each take N
steps (tests/tools/error_budget.py; the round's table is profiles/r02_error_budget.txt).  Two things
are asserted per configuration:

  * the GPU is as close to the extended-precision result as the fp64 oracle is (within a factor 5 —
    measured ratios 0.5 ... 3.1): its partition + cyclic-reduction solves are as accurate as Thomas;
  * GPU vs oracle stays within 10x the value measured on MI355X.

The extended build measures ROUNDING only; it shares the oracle's reading of the reference and says
nothing about the transcription (parity stays unpinned).
"""
import importlib.util
import os

import pytest

from conftest import ROOT, record_error

pytestmark = pytest.mark.gpu

spec = importlib.util.spec_from_file_location("error_budget", os.path.join(ROOT, "tests", "tools", "error_budget.py"))
eb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(eb)

# (name, runner args, {steps: measured GPU-vs-oracle error on MI355X})      profiles/r02_error_budget.txt
MIZ_CASES = [
    ("reference test config", ("sin", 180, 1, 2000, 0, (1, 2, 10, 50)), {1: 2.2e-16, 2: 1.6e-14, 10: 4.5e-14, 50: 1.0e-12}),
    ("cfg2 columns", ("sin", 1440, 2, 131072, 50, (1, 20, 60)), {1: 2.2e-12, 20: 2.3e-12, 60: 2.8e-12}),
    ("cfg5 columns", ("sin", 1024, 8, 65536, 0, (1, 24)), {1: 2.3e-16, 24: 8.7e-13}),
    ("cfg4 columns", ("sin", 4096, 6, 1048576, 50, (1, 10, 40)), {1: 8.8e-14, 10: 2.6e-13, 40: 1.9e-11}),
    ("identity 1024", ("identity", 1024, 8, 262144, 50, (1, 20)), {1: 5.4e-14, 20: 1.2e-13}),
]


@pytest.fixture(autouse=True)
def _restore_geometry_knob():
    """error_budget.miz_case sets EBM_CELLS_PER_THREAD (4: the geometry of every throughput-sized run)."""
    old = os.environ.get("EBM_CELLS_PER_THREAD")
    yield
    if old is None:
        os.environ.pop("EBM_CELLS_PER_THREAD", None)
    else:
        os.environ["EBM_CELLS_PER_THREAD"] = old


@pytest.fixture(scope="module")
def oracles():
    import __graft_entry__ as graft
    _, c_oracle = graft.load_oracle()
    return c_oracle.COracle(), c_oracle.COracle(extended=True)


@pytest.mark.parametrize("name,args,measured", MIZ_CASES, ids=[c[0].replace(" ", "_") for c in MIZ_CASES])
def test_miz_gpu_is_as_accurate_as_the_oracle(pkg, oracles, name, args, measured):
    co, cl = oracles
    for n, g_e, r_e, g_r, _ in eb.miz_case(pkg, co, cl, *args):
        assert g_e[2] and r_e[2] and g_r[2], f"{name} step {n}: NaN sentinels differ"
        record_error(f"{name}, {n} steps: GPU vs 80-bit", g_e[1], g_e[0], 5.0 * r_e[0] + 1e-14)
        record_error(f"{name}, {n} steps: oracle vs 80-bit", r_e[1], r_e[0], float("nan"))
        record_error(f"{name}, {n} steps: GPU vs oracle", g_r[1], g_r[0], 10.0 * measured[n])
        assert g_e[0] <= 5.0 * r_e[0] + 1e-14, f"{name} step {n}: GPU {g_e[0]:.2e} vs oracle {r_e[0]:.2e} from the 80-bit result"
        assert g_r[0] <= 10.0 * measured[n], f"{name} step {n}: {g_r[0]:.2e} > 10 x {measured[n]:.1e}"
        assert g_r[0] <= 1e-10


def test_classic_gpu_is_as_accurate_as_the_oracle(pkg, oracles):
    co, cl = oracles
    measured = {1: 1.7e-14, 40: 8.2e-12, 120: 5.7e-12}
    for n, g_e, r_e, g_r, _ in eb.classic_case(pkg, co, cl, 1024, 16, (1, 40, 120)):
        record_error(f"cfg3 columns, {n} steps: GPU vs 80-bit", g_e[1], g_e[0], 5.0 * r_e[0] + 1e-14)
        record_error(f"cfg3 columns, {n} steps: GPU vs oracle", g_r[1], g_r[0], 10.0 * measured[n])
        assert g_e[0] <= 5.0 * r_e[0] + 1e-14, (n, g_e, r_e)
        assert g_r[0] <= 10.0 * measured[n], (n, g_r)

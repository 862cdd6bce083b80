import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The 2-process Engine test needs a fresh pair of ranks started BEFORE this process has touched the
# GPU (a process that has initialised HIP must not spawn the launcher: on this pool an exec from such
# a process is refused).  So the launcher is started here, at session start, whenever GPU tests are
# selected and a device is present; tests/test_gpu_configs.py waits for it and checks its output.
TWO_RANK = {"proc": None, "out": None, "log": None}
C_EXAMPLE = {"proc": None, "out": None, "err": None}      # examples/c_abi_example.c, same reason
BENCH_LINES = {"proc": None, "out": None}                 # tests/bench_lines_worker.py, same reason


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    markexpr = session.config.getoption("-m") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    if os.environ.get("EBM_TEST_NO_CHILDREN") == "1":
        # runners that start many sessions back to back (tests/tools/mutants_run.sh): a session that stops at its first
        # failure would leave its children on the GPU while the next session starts its own — the box allows six
        # processes on the card; the tests that read the children's output skip
        return
    # is there a GPU?  Asked of the kernel driver's topology, not of torch / HIP: this process must not have
    # initialised the GPU when it starts the children below (energybalancemodel.jl_amd/_devices.py)
    if graft.load_package().visible_gpu_count() < 1:
        return
    tmp = tempfile.mkdtemp(prefix="ebm_two_rank_")
    TWO_RANK["out"] = os.path.join(tmp, "gathered.npz")
    TWO_RANK["log"] = os.path.join(tmp, "two_rank.log")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the plain-C caller of the ABI: build with gcc, run as a child (tests/test_gpu_configs.py checks it)
    exe = os.path.join(tmp, "c_abi_example")
    libdir = os.path.join(ROOT, "energybalancemodel.jl_amd")
    build = subprocess.run(
        ["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"),
         "-o", exe, "-L", libdir, "-lebm_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
         "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    C_EXAMPLE["err"] = build.stderr
    if build.returncode == 0:
        C_EXAMPLE["out"] = os.path.join(tmp, "c_abi_example.out")
        C_EXAMPLE["proc"] = subprocess.Popen([exe], stdout=open(C_EXAMPLE["out"], "w"), stderr=subprocess.STDOUT)
    BENCH_LINES["out"] = os.path.join(tmp, "bench_lines.json")
    BENCH_LINES["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_lines_worker.py"), BENCH_LINES["out"]],
                                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(TWO_RANK["log"], "w") as log:
        TWO_RANK["proc"] = subprocess.Popen(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
             "--master-addr", "127.0.0.1", "--master-port", str(graft.load_package().free_port()),
             os.path.join(ROOT, "tests", "two_rank_worker.py"), TWO_RANK["out"]],
            env=env, stdout=log, stderr=subprocess.STDOUT)


def pytest_sessionfinish(session, exitstatus):
    """A session that ends early (-x) must not leave its children on the GPU: end exactly the processes started above."""
    for child in (TWO_RANK, C_EXAMPLE, BENCH_LINES):
        proc = child.get("proc")
        if proc is not None and proc.poll() is None:
            proc.terminate()
            try:
                proc.wait(timeout=30)
            except Exception:
                proc.kill()


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    o, _ = graft.load_oracle()
    return o


@pytest.fixture(scope="session")
def coracle():
    _, c = graft.load_oracle()
    return c.COracle()


@pytest.fixture(params=[2, 4], ids=["2cells", "4cells"])
def cells(request, monkeypatch):
    """Cells per thread of the launch geometry: the library picks 2 for a few short meridians
    (latency-bound) and 4 otherwise; tests that take this fixture run under both (EBM_CELLS_PER_THREAD;
    meridians of more than 1024 cells always use 4)."""
    monkeypatch.setenv("EBM_CELLS_PER_THREAD", str(request.param))
    return request.param


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def scaled_err(a, b):
    """max |a-b| / max(1, |b|) with NaN sentinels required to coincide."""
    a, b = np.asarray(a), np.asarray(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN sentinels differ"
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


# Every parity test records the error it measured (test id, variable, scaled error, bar): the run's
# table lands in gpurun_out/measured_errors.jsonl and, copied to profiles/, is what the bars in the
# tests are derived from (bar <= 10 x the measured error).
_ERRLOG = os.path.join(ROOT, "gpurun_out", "measured_errors.jsonl")


def record_error(what, var, err, bar):
    import json
    try:
        os.makedirs(os.path.dirname(_ERRLOG), exist_ok=True)
        with open(_ERRLOG, "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0],
                                 "what": what, "var": var, "err": err, "bar": bar}) + "\n")
    except OSError:
        pass

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


# Several GPU tests need OTHER processes on the GPU — a fresh pair of torch.distributed ranks, a plain-C program, bench.py
# itself, an example — and a process that has initialised HIP must not start a launcher (on this pool an exec from such a
# process is refused).  So ONE child, tests/session_children.py, is started here, at session start, before this process
# touches the GPU; it runs those jobs one after the other (the box allows six processes on the card: pytest plus at most two
# ranks of the child's current job) and leaves a `<stage>.done` file as each finishes.  The tests wait for their stage.
class Stage:
    """`proc`-like handle of one stage of the session's child: wait() returns the stage's return code."""

    def __init__(self, worker, done_path):
        self.worker, self.done_path = worker, done_path

    def poll(self):
        if os.path.exists(self.done_path):
            import json
            try:
                return json.load(open(self.done_path))["rc"]
            except ValueError:
                return None                     # being written
        return None

    def wait(self, timeout=900):
        import time
        t0 = time.time()
        while time.time() - t0 < timeout:
            rc = self.poll()
            if rc is not None:
                return rc
            if self.worker.poll() is not None and self.poll() is None:
                return -999                     # the child ended without reaching this stage
            time.sleep(0.2)
        raise TimeoutError(self.done_path)


TWO_RANK = {"proc": None, "out": None, "log": None}
C_EXAMPLE = {"proc": None, "out": None, "err": None}      # examples/c_abi_example.c
BENCH_LINES = {"proc": None, "out": None}                 # the child itself: its last stage writes the bench lines


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    markexpr = session.config.getoption("-m") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    if os.environ.get("EBM_TEST_NO_CHILDREN") == "1":
        # runners that start many sessions back to back (tests/tools/mutants_run.sh): the tests that read the child's
        # output skip
        return
    # is there a GPU?  Asked of the kernel driver's topology, not of torch / HIP: this process must not have
    # initialised the GPU when it starts the child below (energybalancemodel.jl_amd/_devices.py)
    if graft.load_package().visible_gpu_count() < 1:
        return
    tmp = tempfile.mkdtemp(prefix="ebm_session_")
    # the plain-C caller of the ABI: build with gcc here, the child runs it (tests/test_gpu_configs.py checks it)
    exe = os.path.join(tmp, "c_abi_example")
    libdir = os.path.join(ROOT, "energybalancemodel.jl_amd")
    build = subprocess.run(
        ["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"),
         "-o", exe, "-L", libdir, "-lebm_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
         "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    C_EXAMPLE["err"] = build.stderr
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "session_children.py"), tmp, exe if build.returncode == 0 else "-"],
                              env=env, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "session_children.err"), "w"))
    if build.returncode == 0:
        C_EXAMPLE["out"] = os.path.join(tmp, "c_abi_example.out")
        C_EXAMPLE["proc"] = Stage(worker, os.path.join(tmp, "c_example.done"))
    TWO_RANK["out"] = os.path.join(tmp, "gathered.npz")
    TWO_RANK["log"] = os.path.join(tmp, "two_rank.log")
    TWO_RANK["proc"] = Stage(worker, os.path.join(tmp, "two_rank.done"))
    BENCH_LINES["out"] = os.path.join(tmp, "bench_lines.json")
    BENCH_LINES["proc"] = worker


def pytest_sessionfinish(session, exitstatus):
    """A session that ends early (-x) must not leave its child on the GPU: end exactly the process started above (it ends
    the job it is running, tests/session_children.py)."""
    proc = BENCH_LINES.get("proc")
    if proc is not None and proc.poll() is None:
        proc.terminate()
        try:
            proc.wait(timeout=40)
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

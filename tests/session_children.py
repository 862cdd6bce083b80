"""The ONE child process of a `-m gpu` test session (started by tests/conftest.py before the pytest process touches the GPU:
a process that has initialised HIP may not start a launcher).  It runs, ONE AFTER THE OTHER — the box allows six processes
on the card, so the session keeps to the pytest process plus at most two ranks of this worker's current job —

  1. examples/c_abi_example (built by conftest)                        -> test_plain_c_caller_of_the_abi
  2. the two-rank torch.distributed.run pair of tests/two_rank_worker.py -> test_two_process_sharded_engine
  3. bench.py four ways with short settings                             -> test_bench_lines_are_self_consistent
  4. examples/hysteresis_ensemble.py, small                             -> test_hysteresis_example_runs
  5. bench.py --gpus 2 (RCCL: refused on one GPU; gloo: rehearsal), one rank over RCCL -> test_bench_gpus_n_starts_n_ranks

and writes `<stage>.done` (a JSON with the return code) as each of the first two finishes, and the bench lines at the end.
This process never initialises the GPU itself; every job is its own child.  SIGTERM ends the job that is running.

    session_children.py TMPDIR C_EXE|- """
import json, os, signal, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

TMP, C_EXE = sys.argv[1], sys.argv[2]
CURRENT = {"proc": None}


def on_term(signum, frame):
    p = CURRENT["proc"]
    if p is not None and p.poll() is None:
        p.terminate()
        try:
            p.wait(timeout=20)
        except Exception:
            p.kill()
    sys.exit(143)


signal.signal(signal.SIGTERM, on_term)


def run(cmd, env=None, stdout=None, stderr=None, cwd=ROOT):
    """subprocess.run with the child registered for the SIGTERM handler; returns (rc, stdout text, stderr text)."""
    p = subprocess.Popen(cmd, env=env, cwd=cwd, text=True, stdout=stdout if stdout is not None else subprocess.PIPE,
                         stderr=stderr if stderr is not None else subprocess.PIPE)
    CURRENT["proc"] = p
    o, e = p.communicate()
    CURRENT["proc"] = None
    return p.returncode, o or "", e or ""


def done(stage, rc):
    with open(os.path.join(TMP, stage + ".done"), "w") as fh:
        json.dump({"rc": rc}, fh)


pkg = graft.load_package()
ENV = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
# 1. the plain-C caller of the ABI
if C_EXE != "-":
    with open(os.path.join(TMP, "c_abi_example.out"), "w") as fh:
        rc, _, _ = run([C_EXE], stdout=fh, stderr=subprocess.STDOUT)
    done("c_example", rc)
# 2. two ranks driving the HIP library on their shards of one ensemble (gloo: both share the one GPU)
with open(os.path.join(TMP, "two_rank.log"), "w") as log:
    rc, _, _ = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(pkg.free_port()), os.path.join(ROOT, "tests", "two_rank_worker.py"), os.path.join(TMP, "gathered.npz")],
                   env=ENV, stdout=log, stderr=subprocess.STDOUT)
done("two_rank", rc)

RUNS = [
    ["--steps", "20", "--repeats", "3", "--spinup", "200", "--preroll", "0.02", "--cpu-budget", "1"],
    ["--workload", "miz_180x1", "--steps", "64", "--steps-per-launch", "16", "--repeats", "2", "--spinup", "100", "--cpu-budget", "0"],
    ["--workload", "miz_1024x512x32_integrate", "--steps", "4", "--repeats", "2", "--spinup", "20", "--cpu-budget", "0"],
    ["--workload", "classic_1024x512", "--steps", "50", "--repeats", "2", "--cpu-budget", "0"],
]
out = []
for args in RUNS:
    rc, so, se = run([sys.executable, os.path.join(ROOT, "bench.py")] + args)
    lines = [l for l in so.splitlines() if l.strip()]
    out.append({"args": args, "rc": rc, "nlines": len(lines), "line": json.loads(lines[-1]) if lines and rc == 0 else None,
                "stderr_tail": se[-600:]})
# and the hysteresis example end to end, small
rc, so, se = run([sys.executable, os.path.join(ROOT, "examples", "hysteresis_ensemble.py"), "--members", "8", "--nlat", "90", "--nt", "500"])
out.append({"args": ["hysteresis_ensemble.py"], "rc": rc, "stdout": so, "stderr_tail": se[-600:]})
# bench.py --gpus 2 without a launcher: the parent starts the two ranks itself.  On this one-GPU box (a) the default
# backend (RCCL) must refuse loudly — two ranks cannot share a device, and an N-GPU number is never reported from fewer
# devices; (b) EBM_BENCH_BACKEND=gloo rehearses the N > 1 path with both ranks on the one GPU and says so in its line.
two = ["--gpus", "2", "--workload", "miz_1024x512x32", "--steps", "10", "--repeats", "2", "--spinup", "50", "--preroll", "0.02", "--cpu-budget", "0"]
for backend in ("nccl", "gloo"):
    rc, so, se = run([sys.executable, os.path.join(ROOT, "bench.py")] + two, env=dict(os.environ, EBM_BENCH_BACKEND=backend))
    lines = [l for l in so.splitlines() if l.strip()]
    out.append({"args": two, "backend": backend, "rc": rc, "nlines": len(lines),
                "line": json.loads(lines[-1]) if lines and rc == 0 else None, "stderr_tail": se[-1500:]})
# the RCCL code path with ONE rank on the one GPU (launcher + EBM_BENCH_FORCE_DIST=1): communicator set-up on the device,
# barrier, all_reduce of a device tensor, the object gather — everything the N > 1 run does, minus a second device
one = ["--gpus", "1", "--steps", "10", "--repeats", "2", "--spinup", "50", "--preroll", "0.02", "--cpu-budget", "0"]
rc, so, se = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", str(pkg.free_port()), os.path.join(ROOT, "bench.py")] + one,
                 env=dict(ENV, EBM_BENCH_FORCE_DIST="1", EBM_BENCH_BACKEND="nccl"))
lines = [l for l in so.splitlines() if l.strip().startswith("{")]
out.append({"args": one, "backend": "nccl-one-rank", "rc": rc, "nlines": len(lines),
            "line": json.loads(lines[-1]) if lines and rc == 0 else None, "stderr_tail": se[-1500:]})
with open(os.path.join(TMP, "bench_lines.json"), "w") as fh:
    json.dump(out, fh)

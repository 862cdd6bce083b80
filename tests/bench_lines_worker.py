"""Child of tests/conftest.py (started before the pytest process touches the GPU): runs bench.py a few times with short
settings, one after the other, and collects the JSON lines for tests/test_gpu_configs.py::test_bench_lines_are_self_consistent;
then examples/hysteresis_ensemble.py, small, for test_hysteresis_example_runs.
This process never initialises the GPU itself; every bench run is its own child."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    ["--steps", "20", "--repeats", "3", "--spinup", "200", "--preroll", "0.02", "--cpu-budget", "1"],
    ["--workload", "miz_180x1", "--steps", "64", "--steps-per-launch", "16", "--repeats", "2", "--spinup", "100", "--cpu-budget", "0"],
    ["--workload", "miz_1024x512x32_integrate", "--steps", "4", "--repeats", "2", "--spinup", "20", "--cpu-budget", "0"],
    ["--workload", "classic_1024x512", "--steps", "50", "--repeats", "2", "--cpu-budget", "0"],
]
out = []
for args in RUNS:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out.append({"args": args, "rc": r.returncode, "nlines": len(lines), "line": json.loads(lines[-1]) if lines and r.returncode == 0 else None,
                "stderr_tail": r.stderr[-600:]})
# and the hysteresis example end to end, small
r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "hysteresis_ensemble.py"), "--members", "8", "--nlat", "90", "--nt", "500"],
                   capture_output=True, text=True, cwd=ROOT)
out.append({"args": ["hysteresis_ensemble.py"], "rc": r.returncode, "stdout": r.stdout, "stderr_tail": r.stderr[-600:]})
# bench.py --gpus 2 without a launcher: the parent starts the two ranks itself.  On this one-GPU box (a) the default
# backend (RCCL) must refuse loudly — two ranks cannot share a device, and an N-GPU number is never reported from fewer
# devices; (b) EBM_BENCH_BACKEND=gloo rehearses the N > 1 path with both ranks on the one GPU and says so in its line.
two = ["--gpus", "2", "--workload", "miz_1024x512x32", "--steps", "10", "--repeats", "2", "--spinup", "50", "--preroll", "0.02", "--cpu-budget", "0"]
for backend in ("nccl", "gloo"):
    env = dict(os.environ, EBM_BENCH_BACKEND=backend)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + two, capture_output=True, text=True, cwd=ROOT, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out.append({"args": two, "backend": backend, "rc": r.returncode, "nlines": len(lines),
                "line": json.loads(lines[-1]) if lines and r.returncode == 0 else None, "stderr_tail": r.stderr[-1500:]})
# the RCCL code path with ONE rank on the one GPU (launcher + EBM_BENCH_FORCE_DIST=1): communicator set-up on the device,
# barrier, all_reduce of a device tensor, the object gather — everything the N > 1 run does, minus a second device
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
port = graft.load_package().free_port()
one = ["--gpus", "1", "--steps", "10", "--repeats", "2", "--spinup", "50", "--preroll", "0.02", "--cpu-budget", "0"]
r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), os.path.join(ROOT, "bench.py")] + one, capture_output=True, text=True, cwd=ROOT,
                   env=dict(os.environ, EBM_BENCH_FORCE_DIST="1", EBM_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1"))
lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
out.append({"args": one, "backend": "nccl-one-rank", "rc": r.returncode, "nlines": len(lines),
            "line": json.loads(lines[-1]) if lines and r.returncode == 0 else None, "stderr_tail": r.stderr[-1500:]})
with open(sys.argv[1], "w") as fh:
    json.dump(out, fh)

"""Counting the GPUs a process WOULD see, without initialising HIP in the caller.

A process that has initialised the GPU must not start a launcher that re-execs (on this pool the
box refuses such an exec), and `torch.cuda.device_count()` may fall back to `hipGetDeviceCount`,
which initialises the runtime.  So launch decisions (bench.py's `--gpus N` parent, the test
session's pre-launched children) count devices from the kernel driver's topology instead:
`/sys/class/kfd/kfd/topology/nodes/*/properties` lists every KFD node; the ones with
`simd_count > 0` are GPUs.  `HIP_VISIBLE_DEVICES` / `ROCR_VISIBLE_DEVICES` / `CUDA_VISIBLE_DEVICES`
narrow the count the way the runtime would.  If the topology cannot be read although `/dev/kfd`
exists, a throw-away child process asks torch.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

_KFD_NODES = "/sys/class/kfd/kfd/topology/nodes/*/properties"


def _kfd_gpu_nodes():
    """Number of KFD topology nodes that are GPUs, or None if the topology is unreadable."""
    paths = glob.glob(_KFD_NODES)
    if not paths:
        return None
    n = 0
    readable = False
    for p in paths:
        try:
            with open(p) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        readable = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n if readable else None


def _visible_filter(n: int, env) -> int:
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(key)
        if v is None:
            continue
        ids = [s for s in v.split(",") if s.strip() != ""]
        n = min(n, len(ids))
    return n


def visible_gpu_count(env=None) -> int:
    """GPUs a fresh process started with `env` would see.  Never touches HIP in this process."""
    env = os.environ if env is None else env
    if not os.path.exists("/dev/kfd"):
        return 0
    n = _kfd_gpu_nodes()
    if n is None:
        # topology unreadable: ask a child (its HIP initialisation dies with it)
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               capture_output=True, text=True, timeout=600, env=dict(env))
            return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else 0
        except Exception:
            return 0
    return _visible_filter(n, env)


def free_port() -> int:
    """A TCP port that is free right now on 127.0.0.1 (rendezvous of a local torch.distributed.run)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])

"""The one-step state fuzz of tests/test_gpu_parity.py::test_randomized_states_one_step over many more seeds than the suite
runs (GPU box): python tests/tools/fuzz_many.py [first] [count] [miz|classic|imex] — prints the seeds that fail and the worst error seen."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
import test_gpu_parity as T
import __graft_entry__ as g

pkg = g.load_package()
_, c = g.load_oracle()
coracle = c.COracle()


class Env:                                                            # the one thing the test uses monkeypatch for
    def setenv(self, k, v):
        os.environ[k] = v


first = int(sys.argv[1]) if len(sys.argv) > 1 else 24
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
which = sys.argv[3] if len(sys.argv) > 3 else "miz"                    # "miz", "classic" or "imex" (the extension: same generator)
if which == "imex":
    T.make_engine = lambda pkg_, model, st, par, ncol=1: pkg_.Engine("MIZ_IMEX", st.grid_kind, st.x, pkg_.engine.param_vector(par, pkg_.default_parval),
                                                                   st.dt, ncol, device=0)
    _run = coracle.miz_run
    coracle.miz_run = lambda *a, **k: _run(*a, **dict(k, imex=True))
conftest.record_error = lambda *a, **k: None
T.record_error = lambda *a, **k: None
bad = []
for seed in range(first, first + count):
    try:
        if which == "classic":
            T.test_classic_randomized_states_one_step(pkg, coracle, seed, Env())
        else:
            T.test_randomized_states_one_step(pkg, coracle, seed, Env())
    except AssertionError as e:
        bad.append(seed)
        print("seed", seed, "FAILED:", str(e).splitlines()[0][:200], flush=True)
print(f"{count} seeds from {first}: {len(bad)} failures {bad[:20]}")

"""Mutation check of the Python glue between the host mirror and the C ABI, on the GPU box (no build needed):
python tests/tools/mutants_glue.py  — patches one line of energybalancemodel.jl_amd/*.py at a time, runs the -m gpu suite
with -x, restores the file.  One line per mutant: KILLED by <first failing test> or SURVIVED."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "energybalancemodel.jl_amd")
MUTANTS = [
    ("parameter_vector_swaps_A_and_B", "_lib.py", 'PARAM_ORDER = ("D", "A", "B",', 'PARAM_ORDER = ("D", "B", "A",'),
    ("field_ids_swap_Tw_and_Ti", "_lib.py", '"T0": 5, "Tw": 6, "Ti": 7,', '"T0": 5, "Tw": 7, "Ti": 6,'),
    ("time_table_half_a_step_late", "engine.py", "tab = np.array([cos2pit(float(t)) for t in t_in_year], dtype=np.float64)",
     "tab = np.array([cos2pit(float(t) + 0.5 / len(t_in_year)) for t in t_in_year], dtype=np.float64)"),
    ("column_forcing_negated", "engine.py", "a = None if fcol is None else as_f64(fcol, (self.ncol,))", "a = None if fcol is None else -as_f64(fcol, (self.ncol,))"),
    ("run_inverts_diag_last", "engine.py", 'check(self.lib.ebm_run(self._h, int(first_step), int(nsteps), dptr(a), int(diag_last)),',
     'check(self.lib.ebm_run(self._h, int(first_step), int(nsteps), dptr(a), int(not diag_last)),'),
    ("integrate_inverts_lastonly", "engine.py", "check(self.lib.ebm_integrate(self._h, nt, dur, dptr(f), int(lastonly), int(winter_inx),",
     "check(self.lib.ebm_integrate(self._h, nt, dur, dptr(f), int(not lastonly), int(winter_inx),"),
    ("solutions_winter_filled_with_summer", "infrastructure.py", 'sols.seasonal.winter[v] = out["winter"][vi, :, 0, :]', 'sols.seasonal.winter[v] = out["summer"][vi, :, 0, :]'),
]


def main():
    for name, fname, old, new in MUTANTS:
        path = os.path.join(PKG, fname)
        text = open(path).read()
        assert text.count(old) == 1, (name, text.count(old))
        try:
            open(path, "w").write(text.replace(old, new))
            r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                               cwd=ROOT, capture_output=True, text=True, timeout=600)
        finally:
            open(path, "w").write(text)
        first = next((l for l in r.stdout.splitlines() if l.startswith(("FAILED", "ERROR"))), "")
        print(f"{name}: " + ("SURVIVED the GPU suite" if r.returncode == 0 else f"KILLED, first: {first[:150]}"), flush=True)


if __name__ == "__main__":
    main()

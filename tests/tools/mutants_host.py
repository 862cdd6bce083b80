"""Mutation check of the HOST mirror (energybalancemodel.jl_amd/*.py) against the CPU suite: python tests/tools/mutants_host.py
Each mutant is one textual change in a temporary copy of the package (EBM_PKG_DIR is not needed: the copy shadows the
package through PYTHONPATH order is NOT used — the file is patched in place and restored).  One line per mutant: KILLED by
<first failing test> or SURVIVED (then the -m gpu suite is the next line of defence)."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "energybalancemodel.jl_amd")
MUTANTS = [
    ("winter_index_truncated", "infrastructure.py", "self.winter = Collection(t=winter, inx=int(round(self.nt * winter)))", "self.winter = Collection(t=winter, inx=int(self.nt * winter))"),
    ("time_axis_at_step_starts", "infrastructure.py", "self.t = np.array([float(Fraction(2 * i + 1, 2 * self.nt)) for i in range(self.nt)])", "self.t = np.array([float(Fraction(2 * i, 2 * self.nt)) for i in range(self.nt)])"),
    ("sin_grid_to_the_other_pole", "infrastructure.py", "urange = (0.0, math.pi / 2.0)", "urange = (0.0, math.pi)"),
    ("forcing_warming_time_from_cooling_rate", "infrastructure.py", "warming = (peak - base) / rates[0]", "warming = (peak - base) / -rates[1]"),
    ("forcing_ramp_ignores_its_start", "infrastructure.py", "return self.base + self.rates[0] * (T - d[1])", "return self.base + self.rates[0] * T"),
    ("forcing_cooling_from_base", "infrastructure.py", "return self.peak + self.rates[1] * (T - d[3])", "return self.base + self.rates[1] * (T - d[3])"),
    ("classic_time_index_without_the_half_step", "infrastructure.py", "y = (t + dt / 2.0) * nt", "y = t * nt"),
    ("shards_overlap_by_one", "ensemble.py", "return slice(start, start + base + (1 if rank < extra else 0))", "return slice(start, start + base + 1)"),
    ("shards_ignore_the_remainder", "ensemble.py", "start = rank * base + min(rank, extra)", "start = rank * base"),
]


def main():
    for name, fname, old, new in MUTANTS:
        path = os.path.join(PKG, fname)
        text = open(path).read()
        assert text.count(old) == 1, (name, text.count(old))
        try:
            open(path, "w").write(text.replace(old, new))
            r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"],
                               cwd=ROOT, capture_output=True, text=True)
        finally:
            open(path, "w").write(text)
        first = next((l for l in r.stdout.splitlines() if l.startswith(("FAILED", "ERROR"))), "")
        print(f"{name}: " + ("SURVIVED the CPU suite" if r.returncode == 0 else f"KILLED, first: {first[:140]}"), flush=True)


if __name__ == "__main__":
    main()

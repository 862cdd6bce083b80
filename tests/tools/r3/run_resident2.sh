#!/bin/bash
# GPU box: fused-K bench lines of the large shapes (LDS-resident kernel) next to their K = 1 lines
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3p
: > gpurun_out/r3p/fused_lines.jsonl
run() {
  timeout -k 10 200 python bench.py --cpu-budget 0 "$@" > gpurun_out/r3p/line.json 2> gpurun_out/r3p/line.err || { tail -5 gpurun_out/r3p/line.err; exit 1; }
  cat gpurun_out/r3p/line.json >> gpurun_out/r3p/fused_lines.jsonl
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3p/line.json").read())
print(d["config"]["workload"], "|", round(d["ms_per_step"], 5), "ms/step |", d["roofline"]["kernel"], "| solves/col-step", d["config"].get("solves_per_column_step"))
PY
}
run --workload miz_imex_4096x2048
run --workload miz_imex_4096x2048 --steps-per-launch 16
run --workload miz_imex_4096x2048 --steps-per-launch 64
run --workload miz_4096x2048 --steps-per-launch 64 --launch-chains 2
run --workload miz_2048x4096
run --workload miz_2048x4096 --steps-per-launch 64
run --workload miz_4096x2048 --steps-per-launch 256 --steps 1024

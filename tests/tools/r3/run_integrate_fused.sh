#!/bin/bash
# GPU box: the integrate workload with one launch per step and fused (default), plus a 4096-latitude variant
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3p
: > gpurun_out/r3p/integrate_lines.jsonl
run() {
  timeout -k 10 300 python bench.py --cpu-budget 0 "$@" > gpurun_out/r3p/line.json 2> gpurun_out/r3p/line.err || { tail -5 gpurun_out/r3p/line.err; exit 1; }
  cat gpurun_out/r3p/line.json >> gpurun_out/r3p/integrate_lines.jsonl
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3p/line.json").read())
print(d["config"]["workload"][:50], "|", round(d["ms_per_step"], 5), "ms/step | year_end", d.get("year_end_ms"), "|", d["roofline"]["kernel"], "frac", round(d["roofline"]["frac"], 3), "spl", d["config"].get("steps_per_launch"))
PY
}
run --workload miz_1024x512x32_integrate --steps 256 --repeats 3 --integrate-steps-per-launch 1
run --workload miz_1024x512x32_integrate --steps 256 --repeats 3
run --workload miz_1024x512x32_integrate --steps 256 --repeats 3 --integrate-steps-per-launch 16
run --workload miz_1024x512x32_integrate --steps 1024 --repeats 3 --integrate-steps-per-launch 256

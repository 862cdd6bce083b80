# Here (not on the GPU box): copy what is to be judged from gpurun_out/r3final (written by run_round3_final_profiles.sh) to profiles/r03_*
set -e
cd "$(dirname "$0")/../../.."
O=gpurun_out/r3final
ks() { ls -t $O/$1/*/*_kernel_stats.csv | head -1; }   # the newest: gpurun_out/ accumulates over calls
cp $O/bench_default.json profiles/r03_bench.json
cp $O/prof_headline.json profiles/r03_bench_under_rocprof.json
cp "$(ks prof_headline)" profiles/r03_kernel_stats.csv
cp "$(ks prof_step)" profiles/r03_step_diag_kernel_stats.csv
cp $O/prof_step.json profiles/r03_bench_step_diag.json
cp "$(ks prof_integrate)" profiles/r03_integrate_kernel_stats.csv
cp "$(ks prof_integrate_fused)" profiles/r03_integrate_fused_kernel_stats.csv
cp $O/docstring_example.log profiles/r03_docstring_example.log
cp "$(ks prof_fused180)" profiles/r03_fused_180_kernel_stats.csv
cp "$(ks prof_resident)" profiles/r03_resident_kernel_stats.csv
cp $O/prof_resident.json profiles/r03_bench_resident.json
cp "$(ks prof_imex)" profiles/r03_imex_kernel_stats.csv
cp "$(ks prof_zonal)" profiles/r03_zonal_kernel_stats.csv
cp "$(ks prof_zonal_single)" profiles/r03_zonal_single_grid_kernel_stats.csv
cp $O/pmc_summary.txt profiles/r03_pmc_summary.txt
cp $O/pmc_step_summary.txt profiles/r03_pmc_step_diag.txt
cp $O/pmc_zonal_summary.txt profiles/r03_pmc_zonal.txt
cp $O/other_workloads.jsonl profiles/r03_other_workloads.jsonl
cp $O/measured_errors.jsonl profiles/r03_measured_errors.jsonl
cp $O/bench_two_ranks_one_gpu.log profiles/r03_bench_two_ranks_one_gpu.log
cp $O/bench_gpus2_refused.log profiles/r03_bench_gpus2_refused.log
[ -f gpurun_out/two_rank_rehearsal.log ] && cp gpurun_out/two_rank_rehearsal.log profiles/r03_two_rank_rehearsal.log
python - <<'PY'
import json
d = json.loads(open("profiles/r03_bench.json").read())
print("headline", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for l in open("profiles/r03_other_workloads.jsonl"):
    e = json.loads(l)
    print(e["config"]["workload"].split(":")[0][:70].ljust(70), round(e["ms_per_step"], 5), e["roofline"]["kernel"], round(e["roofline"]["frac"], 3))
PY

# The round's final evidence batch (GPU box): the -m gpu suite with its measured errors, rocprofv3 kernel stats of the
# headline, step!-semantics, integrate, fused-K, extension and zonal workloads, PMC passes of the headline and diagnostic
# kernels, bench lines of the other workloads, the N > 1 rehearsals.  Copy what is to be judged from gpurun_out/r3final to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final
mkdir -p $O
rm -f gpurun_out/measured_errors.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
cp gpurun_out/measured_errors.jsonl $O/measured_errors.jsonl
set -e
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python3 bench.py --cpu-budget 0 > $O/prof_headline.json 2> $O/prof_headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -- python3 bench.py --workload miz_4096x2048_step --cpu-budget 0 > $O/prof_step.json 2> $O/prof_step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_integrate -- python3 bench.py --workload miz_1024x512x32_integrate --steps 200 --repeats 2 --cpu-budget 0 --integrate-steps-per-launch 1 > $O/prof_integrate.json 2> $O/prof_integrate.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_integrate_fused -- python3 bench.py --workload miz_1024x512x32_integrate --steps 256 --repeats 2 --cpu-budget 0 > $O/prof_integrate_fused.json 2> $O/prof_integrate_fused.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fused180 -- python3 bench.py --workload miz_180x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > $O/prof_fused180.json 2> $O/prof_fused180.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_resident -- python3 bench.py --workload miz_4096x2048 --steps 1024 --steps-per-launch 64 --cpu-budget 0 --repeats 2 > $O/prof_resident.json 2> $O/prof_resident.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_imex -- python3 bench.py --workload miz_imex_4096x2048 --cpu-budget 0 --repeats 2 > $O/prof_imex.json 2> $O/prof_imex.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zonal -- python3 tests/tools/zonal_profile.py 32 6 > $O/prof_zonal.log 2>&1
set +e
bash tests/tools/pmc_passes.sh $O/pmc
python tests/tools/pmc_summary.py $O/pmc "miz_step_kernel<4, 1, 0, 1024, false>" 3 > $O/pmc_summary.txt 2>&1
B="python3 bench.py --workload miz_4096x2048_step --steps 3 --warmup 0 --spinup 300 --cpu-budget 0 --preroll 0 --repeats 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_step_$C -- $B > $O/pmc_step_$C.log 2>&1 || echo "pass $C failed"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_zonal_$C -- python3 tests/tools/zonal_profile.py 32 3 > $O/pmc_zonal_$C.log 2>&1 || echo "zonal pass $C failed"
done
python tests/tools/pmc_summary.py $O/pmc_step_FETCH_SIZE "miz_step_kernel<4, 1, 1, 1024, false>" 3 > $O/pmc_step_summary.txt 2>&1
python tests/tools/pmc_summary.py $O/pmc_step_WRITE_SIZE "miz_step_kernel<4, 1, 1, 1024, false>" 3 >> $O/pmc_step_summary.txt 2>&1
: > $O/pmc_zonal_summary.txt
for K in zonal_seg_forward_kernel zonal_reduced_solve_kernel zonal_seg_backward_kernel; do
  echo "# $K" >> $O/pmc_zonal_summary.txt
  python tests/tools/pmc_summary.py $O/pmc_zonal_FETCH_SIZE "$K" 2 >> $O/pmc_zonal_summary.txt 2>&1
  python tests/tools/pmc_summary.py $O/pmc_zonal_WRITE_SIZE "$K" 2 >> $O/pmc_zonal_summary.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zonal_single -- python3 tests/tools/zonal_profile.py 1 4 4096 2048 > $O/prof_zonal_single.log 2>&1
: > $O/other_workloads.jsonl
for W in "miz_4096x2048 --launch-chains 2" "miz_4096x2048_step" "miz_4096x2048_step --launch-chains 2" \
         "miz_180x1 --steps 2000" "miz_180x1 --steps 2048 --steps-per-launch 64" "miz_180x1 --steps 2048 --steps-per-launch 1024" \
         "miz_1440x1 --steps 2000" "miz_1440x1 --steps 2048 --steps-per-launch 64" \
         "miz_180x8192 --steps 512" "miz_180x8192 --steps 512 --steps-per-launch 64" \
         "miz_1024x512x32" "miz_1024x512x32_integrate --steps 256 --repeats 3 --integrate-steps-per-launch 1" "miz_1024x512x32_integrate --steps 256 --repeats 3" \
         "miz_1024x512x32_integrate --steps 1024 --repeats 3 --integrate-steps-per-launch 256" "miz_2048x4096" \
         "classic_1024x512 --steps 2000" "classic_1024x512 --steps 2048 --steps-per-launch 64" "miz_imex_4096x2048" "miz_imex_4096x2048 --launch-chains 2" \
         "miz_4096x2048 --steps-per-launch 16" "miz_4096x2048 --steps 1024 --steps-per-launch 64" "miz_4096x2048 --steps 1024 --steps-per-launch 256" \
         "miz_2048x4096 --steps 1024 --steps-per-launch 64" "miz_2048x4096 --steps 1024 --steps-per-launch 64 --fused-state registers" \
         "miz_1024x512x32 --steps 512 --steps-per-launch 64" "miz_1024x512x32 --steps 512 --steps-per-launch 64 --fused-state registers" \
         "miz_180x8192 --steps 512 --steps-per-launch 64 --fused-state registers" \
         "miz_imex_4096x2048 --steps-per-launch 16" "miz_imex_4096x2048 --steps 1024 --steps-per-launch 64"; do
  python bench.py --workload $W --cpu-budget 0 >> $O/other_workloads.jsonl 2>> $O/other_workloads.err
done
EBM_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload miz_1024x512x32 --steps 50 --cpu-budget 0 > $O/bench_two_ranks_one_gpu.log 2>&1
python bench.py --gpus 2 --steps 20 --cpu-budget 0 > $O/bench_gpus2_refused.log 2>&1; echo "rc=$?" >> $O/bench_gpus2_refused.log
python tests/tools/docstring_example.py > $O/docstring_example.log 2>&1
tail -c 400 $O/bench_default.json

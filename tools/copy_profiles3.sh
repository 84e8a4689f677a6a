# copy the summaries of one tools/profile_round3.sh run (gpurun_out/<dir>) into profiles/r03_*
S=${1:?gpurun_out dir}; P=profiles
cp $S/bench.json $P/r03_bench.json
for NT in 1 16 100; do
  cp $S/bench_under_rocprof_nt$NT.json $P/r03_bench_under_rocprof_nt$NT.json
  cp $(ls $S/stats_nt$NT/*/*kernel_stats.csv | head -1) $P/r03_kernel_stats_nt$NT.csv
done
cp $S/pmc_insts.json $P/r03_pmc_insts.json
cp $S/pmc_traffic_nt100.json $P/r03_pmc_traffic_nt100.json
python3 tools/pmc_summary.py $S/pmc_cyc ais_half > /dev/null 2>&1 || true
cp $(ls $S/smc_stats/*/*kernel_stats.csv | head -1) $P/r03_smc_c4_loop_kernel_stats.csv
cp $(ls $S/smc_stats_kernels/*/*kernel_stats.csv | head -1) $P/r03_smc_c4_kernels_path_kernel_stats.csv
cp $S/smc_c4.txt $P/r03_smc_c4.txt
cp $S/config_sweep.jsonl $P/r03_config_sweep.jsonl
grep '^cycles' $S/pmc_collect.log | sed "s/^cycles //" > /tmp/_cyc.txt
python3 - <<PY
import ast, json
c = ast.literal_eval(open('/tmp/_cyc.txt').read().strip())
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- python3 bench.py --no-alt --no-smc --no-cpu-baseline --steps 30 --warmup 5 --min-seconds 0.1 (ntransitions = 100)",
           "kernel": "ais_half_kernel<8, rosenbrock, BOX, kernelized>", "unit": "per launch (mean)", **c},
          open('profiles/r03_pmc_cycles_nt100.json', 'w'), indent=1)
PY
git status --short profiles | head -20
cp $S/pmc_traffic_smc_loop.json $P/r03_pmc_traffic_smc_loop.json
cp $(ls $S/readme_stats/*/*kernel_stats.csv | head -1) $P/r03_readme_kernel_stats.csv
cp $S/readme_under_rocprof.json $P/r03_readme_under_rocprof.json

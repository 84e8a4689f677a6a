# copy the summaries of one tools/profile_round4.sh run (gpurun_out/<dir>) into profiles/r04_*
S=${1:?gpurun_out dir}; P=profiles
cp $S/bench.json $P/r04_bench.json
for NT in 1 16 100; do
  cp $S/bench_under_rocprof_nt$NT.json $P/r04_bench_under_rocprof_nt$NT.json
  cp $(ls -t $S/stats_nt$NT/*/*kernel_stats.csv | head -1) $P/r04_kernel_stats_nt$NT.csv
done
cp $S/pmc_insts.json $P/r04_pmc_insts.json
cp $S/pmc_traffic_nt100.json $P/r04_pmc_traffic_nt100.json
cp $S/pmc_traffic_smc_loop.json $P/r04_pmc_traffic_smc_loop.json
cp $(ls -t $S/smc_stats/*/*kernel_stats.csv | head -1) $P/r04_smc_c4_loop_kernel_stats.csv
cp $S/smc_c4.txt $P/r04_smc_c4.txt
cp $S/smc_c4_spec.txt $P/r04_smc_c4_spec.txt
cp $S/config_sweep.jsonl $P/r04_config_sweep.jsonl
cp $(ls -t $S/readme_stats/*/*kernel_stats.csv | head -1) $P/r04_readme_kernel_stats.csv
cp $S/readme_under_rocprof.json $P/r04_readme_under_rocprof.json
cp $(ls -t $S/readme_smc_stats/*/*kernel_stats.csv | head -1) $P/r04_readme_smc_kernel_stats.csv
cp $S/smc_small.json $P/r04_smc_small.json
cp $S/spec_probe.json $P/r04_spec_probe.json
cp $(ls -t $S/spec_stats/*/*kernel_stats.csv | head -1) $P/r04_spec_kernel_stats.csv
cp $S/smc_scaling.jsonl $P/r04_smc_scaling.jsonl
cp $S/abcde.json $P/r04_abcde.json
grep '^cycles' $S/pmc_collect.log | sed "s/^cycles //" > /tmp/_cyc.txt
python3 - <<PY
import ast, json
c = ast.literal_eval(open('/tmp/_cyc.txt').read().strip())
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- python3 bench.py --no-alt --no-smc --no-cpu-baseline --steps 30 --warmup 5 --min-seconds 0.1 --headline-seconds 0.1 (ntransitions = 100)",
           "kernel": "ais_half_kernel<8, rosenbrock, BOX, kernelized>", "unit": "per launch (mean)", **c},
          open('profiles/r04_pmc_cycles_nt100.json', 'w'), indent=1)
PY
git status --short profiles | head -30

# copy the summaries of one tools/profile_round6.sh run (gpurun_out/<dir>) into profiles/r06_*
S=${1:?gpurun_out dir}; P=profiles
cp $S/bench.json $P/r06_bench.json
for NT in 1 16 100; do
  cp $S/bench_under_rocprof_nt$NT.json $P/r06_bench_under_rocprof_nt$NT.json
  cp $(ls -t $S/stats_nt$NT/*/*kernel_stats.csv | head -1) $P/r06_kernel_stats_nt$NT.csv
done
cp $S/pmc_insts.json $P/r06_pmc_insts.json
cp $S/pmc_traffic_nt100.json $P/r06_pmc_traffic_nt100.json
cp $S/pmc_traffic_smc_loop.json $P/r06_pmc_traffic_smc_loop.json
cp $(ls -t $S/smc_stats/*/*kernel_stats.csv | head -1) $P/r06_smc_c4_loop_kernel_stats.csv
cp $S/smc_c4.txt $P/r06_smc_c4.txt
cp $S/xcd_barrier.json $P/r06_xcd_barrier.json
cp $S/reference_test_shapes.jsonl $P/r06_reference_test_shapes.jsonl
cp $(ls -t $S/small_stats/*/*kernel_stats.csv | head -1) $P/r06_reference_shapes_kernel_stats.csv
cp $S/dyn_teams.txt $P/r06_dyn_teams.txt
cp $S/config_sweep.jsonl $P/r06_config_sweep.jsonl
cp $(ls -t $S/readme_stats/*/*kernel_stats.csv | head -1) $P/r06_readme_kernel_stats.csv
cp $S/readme_under_rocprof.json $P/r06_readme_under_rocprof.json
cp $(ls -t $S/readme_smc_stats/*/*kernel_stats.csv | head -1) $P/r06_readme_smc_kernel_stats.csv
cp $S/smc_small.json $P/r06_smc_small.json
cp $S/spec_probe.json $P/r06_spec_probe.json
cp $S/smc_scaling.jsonl $P/r06_smc_scaling.jsonl
cp $S/abcde.json $P/r06_abcde.json
cp $S/pfilter.txt $P/r06_pfilter.txt
cp $S/small_defaults.txt $P/r06_small_defaults.txt
cp $S/smc_dist_modes.jsonl $P/r06_smc_dist_modes.jsonl
cp $S/bench_emulated8.json $P/r06_bench_emulated8_rehearsal.json
grep '^cycles' $S/pmc_collect.log | sed "s/^cycles //" > /tmp/_cyc.txt
python3 - <<PY
import ast, json
c = ast.literal_eval(open('/tmp/_cyc.txt').read().strip())
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- python3 bench.py --no-alt --no-smc --no-cpu-baseline --steps 30 --warmup 5 --min-seconds 0.1 --headline-seconds 0.1 (ntransitions = 100)",
           "kernel": "ais_half_kernel<8, rosenbrock, BOX, kernelized>", "unit": "per launch (mean)", **c},
          open('profiles/r06_pmc_cycles_nt100.json', 'w'), indent=1)
PY
git status --short profiles | head -40

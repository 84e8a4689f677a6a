# Round-6 evidence run (one GPU box): bench line, rocprofv3 kernel stats per ntransitions setting, PMC passes,
# smc C4 (stats + the PROBES build's phase stamps), the XCD barrier probe, the reference's own sample() shapes on
# both AIS drivers (+ kernel stats of the one-workgroup kernel), the run-time-dimension kernel's team sizes.
#   gpurun -- 'bash tools/profile_round6.sh gpurun_out/r06p'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r06p}; mkdir -p $O
(rocm-smi --showclocks --showpower 2>/dev/null || true) > $O/rocm_smi_before.txt
python3 bench.py > $O/bench.json 2> $O/bench.err
KABC_BENCH_EMULATE_RANKS=8 python3 bench.py --gpus 8 --no-cpu-baseline --steps 50 --warmup 5 > $O/bench_emulated8.json 2> $O/bench_emulated8.err
for NT in 1 16 100; do
  B="python3 bench.py --ntransitions $NT --no-alt --no-smc --no-cpu-baseline --steps 50 --warmup 5 --min-seconds 0.2 --headline-seconds 0.2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nt$NT -- $B > $O/bench_under_rocprof_nt$NT.json 2> $O/stats_nt$NT.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $O/pmc_inst_nt$NT -- $B > /dev/null 2>&1
done
B="python3 bench.py --no-alt --no-smc --no-cpu-baseline --steps 30 --warmup 5 --min-seconds 0.1 --headline-seconds 0.1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_cyc -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/smc_stats -- python3 tools/smc_c4_probe.py > $O/smc_c4_under_rocprof.json 2> /dev/null
python3 tools/smc_c4_probe.py --oracle > $O/smc_c4.txt 2>&1
KABC_SPECIALIZE=0 KABC_PROBES=1 KABC_SMC_STAMPS=1 python3 tools/smc_c4_probe.py >> $O/smc_c4.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_smc -- python3 tools/smc_c4_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_smc -- python3 tools/smc_c4_probe.py > /dev/null 2>&1
./tools/_bin/xcdbar > $O/xcd_barrier.json 2> $O/xcd_barrier.err
# the reference's own sample() shapes: both drivers, phases, the oracle's one-core wall; the small kernel under rocprofv3
python3 tools/reference_tests_probe.py --oracle > $O/reference_test_shapes.jsonl 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_stats -- python3 tools/reference_tests_probe.py > /dev/null 2>&1
python3 tools/dyn_prior_share_probe.py > $O/dyn_teams.txt 2>&1
for T in 4 8 16 32 64; do echo "== KABC_DYN_TEAM=$T" >> $O/dyn_teams.txt; KABC_DYN_TEAM=$T python3 tools/dyn_prior_share_probe.py 2>&1 | grep "box" >> $O/dyn_teams.txt; done
# the README workloads
rocprofv3 --kernel-trace --stats --output-format csv -d $O/readme_stats -- python3 tools/readme_probe.py > $O/readme_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/readme_smc_stats -- python3 tools/smc_small_probe.py > $O/readme_smc_under_rocprof.json 2> /dev/null
python3 tools/smc_small_probe.py > $O/smc_small.json 2>/dev/null
python3 tools/spec_probe.py > $O/spec_probe.json 2>/dev/null
python3 tools/smc_scaling_probe.py 131072 524288 2097152 > $O/smc_scaling.jsonl 2>/dev/null
python3 tools/abcde_probe.py > $O/abcde.json 2>/dev/null
python3 tools/pfilter_probe.py > $O/pfilter.txt 2>/dev/null
python3 tools/small_defaults_probe.py > $O/small_defaults.txt 2>/dev/null
KABC_NO_TORCH_PRELOAD=1 python3 tools/smc_dist_probe.py > $O/smc_dist_modes.jsonl 2>/dev/null
python3 tools/pmc_collect.py $O > $O/pmc_collect.log 2>&1
python3 tools/config_sweep.py > $O/config_sweep.jsonl 2>/dev/null
(rocm-smi --showclocks --showpower 2>/dev/null || true) > $O/rocm_smi_after.txt
ls $O; cat $O/pmc_collect.log | tail -20

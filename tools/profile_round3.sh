# Round-3 evidence run (one GPU box): bench line, rocprofv3 kernel stats per ntransitions
# setting, PMC passes (instruction counts per setting; HBM traffic), smc C4 stats.
#   gpurun -- 'bash tools/profile_round3.sh gpurun_out/r03p'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03p}; mkdir -p $O
(rocm-smi --showclocks --showpower 2>/dev/null || true) > $O/rocm_smi_before.txt
python3 bench.py > $O/bench.json 2> $O/bench.err
for NT in 1 16 100; do
  B="python3 bench.py --ntransitions $NT --no-alt --no-smc --no-cpu-baseline --steps 50 --warmup 5 --min-seconds 0.2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nt$NT -- $B > $O/bench_under_rocprof_nt$NT.json 2> $O/stats_nt$NT.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $O/pmc_inst_nt$NT -- $B > /dev/null 2>&1
done
B="python3 bench.py --no-alt --no-smc --no-cpu-baseline --steps 30 --warmup 5 --min-seconds 0.1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_cyc -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/smc_stats -- python3 tools/smc_c4_probe.py > $O/smc_c4_under_rocprof.json 2> /dev/null
KABC_SMC_LOOP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/smc_stats_kernels -- python3 tools/smc_c4_probe.py > $O/smc_c4_kernels_under_rocprof.json 2> /dev/null
KABC_SMC_STAMPS=1 python3 tools/smc_c4_probe.py --oracle > $O/smc_c4.txt 2>&1
# HBM traffic of the persistent smc loop kernel (one launch = the whole C4 run), separate passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_smc -- python3 tools/smc_c4_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_smc -- python3 tools/smc_c4_probe.py > /dev/null 2>&1
# the README workload: kernel stats of one sample() call (pre-pass + half-generation kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/readme_stats -- python3 tools/readme_probe.py > $O/readme_under_rocprof.json 2> /dev/null
python3 tools/pmc_collect.py $O > $O/pmc_collect.log 2>&1
python3 tools/config_sweep.py > $O/config_sweep.jsonl 2>/dev/null
(rocm-smi --showclocks --showpower 2>/dev/null || true) > $O/rocm_smi_after.txt
ls $O; cat $O/pmc_collect.log | tail -20

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r01m}; mkdir -p $O
(rocm-smi --showclocks --showpower 2>/dev/null || true) > $O/rocm_smi_before.txt
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-alt > $O/bench_under_rocprof.json 2> $O/stats.err
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt"   # default ntransitions = 100; add --ntransitions 16 for the secondary figure
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $O/pmc_inst -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_cyc -- $B > /dev/null 2>&1
for d in pmc_fetch pmc_write pmc_inst pmc_cyc; do python3 tools/pmc_summary.py "$O/$d/**/*counter_collection.csv"; done > $O/pmc_summary.txt
KABC_NT=16 python3 tools/barrier_probe.py > $O/barrier_nt16.txt 2>&1
KABC_NT=100 python3 tools/barrier_probe.py > $O/barrier_nt100.txt 2>&1
python3 tools/placement_probe.py > $O/placement.txt 2>&1
python3 tools/trace_probe.py --gens 128 > $O/trace_probe.json 2>/dev/null
python3 tools/pcie_probe.py > $O/pcie_probe.json 2>/dev/null
hipcc -O2 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate 8 > $O/valu_rate_w8.json && /tmp/valu_rate 1 > $O/valu_rate_w1.json
hipcc -O2 --offload-arch=gfx950 tools/valu_latency.hip -o /tmp/valu_latency 2>/dev/null && /tmp/valu_latency > $O/valu_latency.json
KABC_SMC_STAMPS=1 python3 tools/smc_c4_probe.py --oracle > $O/smc_c4.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/smc_stats -- python3 tools/smc_c4_probe.py > /dev/null 2>&1
find $O -name "*kernel_stats.csv" | head; tail -1 $O/bench.json | cut -c1-400; cat $O/pmc_summary.txt | head -40
(rocm-smi --showclocks --showpower 2>/dev/null || true) > $O/rocm_smi_after.txt
python3 tools/config_sweep.py > $O/config_sweep.jsonl 2>/dev/null

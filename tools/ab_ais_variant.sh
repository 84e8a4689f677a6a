# A/B of the default library against an experimental build (make VARIANT=<name> EXTRA=...) on ONE
# GPU box, AIS half-generation kernel: bash tools/ab_ais_variant.sh <name> [rounds]
# prints (kernel us, G evals/s) at ntransitions = 1, 16, 100, alternating, then runs the AIS parity
# tests on the experimental build.
V=$1; R=${2:-3}
run() {
  python bench.py --no-cpu-baseline --no-smc --no-cold-spec --min-seconds 0.5 --headline-seconds 1.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({nt:(round(v['kernel_avg_us'],2), round(v['value']/1e9,2)) for nt,v in d['by_ntransitions'].items()})"
}
for i in $(seq $R); do
  echo -n "default  "; run
  echo -n "$V  "; KABC_LIB=$PWD/kissabc.jl_amd/lib/libkabc_hip_$V.so run
done
KABC_LIB=$PWD/kissabc.jl_amd/lib/libkabc_hip_$V.so timeout 900 python -m pytest tests/test_gpu_ais_parity.py tests/test_gpu_random_sweep.py -x -q 2>&1 | tail -2

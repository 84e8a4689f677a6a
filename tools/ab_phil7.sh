run() {
  python bench.py --no-cpu-baseline --no-smc --no-alt --min-seconds 0.5 --ntransitions $1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({nt:(round(v['kernel_avg_us'],2), round(v['roofline_frac'],3)) for nt,v in d['by_ntransitions'].items()})"
}
for i in 1 2; do
  for nt in 100 16 1; do
  echo -n "default nt$nt "; run $nt
  echo -n "phil7 nt$nt  "; KABC_LIB=$PWD/kissabc.jl_amd/lib/libkabc_hip_phil7.so run $nt
  done
done

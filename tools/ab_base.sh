# A/B of the AIS half-generation kernel between library builds on one box:
#   tools/ab_base.sh [lib ...]     (default: tools/_base.so) -- each against the in-tree library
run() {
  python bench.py --no-cpu-baseline --no-smc --no-alt --min-seconds 0.5 --ntransitions $1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({nt:(round(v['kernel_avg_us'],2), round(v['roofline_frac'],3)) for nt,v in d['by_ntransitions'].items()})"
}
LIBS=${@:-tools/_base.so}
for i in 1 2; do
  for nt in 100 16 1; do
  echo -n "in-tree nt$nt "; run $nt
  for l in $LIBS; do echo -n "$l nt$nt  "; KABC_LIB=$PWD/$l run $nt; done
  done
done

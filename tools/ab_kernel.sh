# A/B of two builds of libkabc_hip.so on ONE GPU box (boxes differ by ~1 %, so both builds must run
# in the same call): the working tree's build against tools/_base.so, a copy of an earlier build.
#   cp kissabc.jl_amd/lib/libkabc_hip.so tools/_base.so      # before changing the kernel
#   ... edit, make -C kissabc.jl_amd/csrc ...
#   gpurun -- 'bash tools/ab_kernel.sh'
# prints (kernel us, G evals/s) of the AIS half-generation kernel at ntransitions = 1, 16, 100,
# twice per build, alternating.  tools/_base.so matches *.so in .gitignore and is not committed.
run() {
  python bench.py --no-cpu-baseline --no-smc --min-seconds 0.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({nt:(round(v['kernel_avg_us'],2), round(v['value']/1e9,2)) for nt,v in d['by_ntransitions'].items()})"
}
cp kissabc.jl_amd/lib/libkabc_hip.so /tmp/_new.so
for i in 1 2; do
  echo -n "new  "; run
  cp tools/_base.so kissabc.jl_amd/lib/libkabc_hip.so
  echo -n "base "; run
  cp /tmp/_new.so kissabc.jl_amd/lib/libkabc_hip.so
done

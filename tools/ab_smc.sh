# A/B of the working tree's library against tools/_base.so on the C4 smc wall time (one box)
cp kissabc.jl_amd/lib/libkabc_hip.so /tmp/_new.so
for i in 1 2 3; do
  echo -n "new  "; python tools/smc_wall_probe.py $1 2>/dev/null | tail -1
  echo -n "base "; KABC_LIB=$PWD/tools/_base.so python tools/smc_wall_probe.py 2>/dev/null | tail -1
done

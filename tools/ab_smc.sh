# A/B of the working tree's library against tools/_base.so on the C4 smc wall time (one box);
# every argument goes to BOTH legs.  (A library loaded from tools/ finds the kernel headers for
# run-time compilation through KABC_RTC_INCLUDE.)
export KABC_RTC_INCLUDE=$PWD/kissabc.jl_amd/csrc:$PWD/include
for i in 1 2 3; do
  echo -n "new  "; python tools/smc_wall_probe.py "$@" 2>/dev/null | tail -1
  echo -n "base "; KABC_LIB=$PWD/tools/_base.so python tools/smc_wall_probe.py "$@" 2>/dev/null | tail -1
done

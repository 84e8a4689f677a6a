# A/B of the default library build against an experimental one (make VARIANT=<name> EXTRA=...)
# on ONE GPU box: bash tools/ab_variant.sh <name> [probe command]
# default probe: C4 smc wall time (median of 9 after warm-up) + bit-exactness against the oracle
V=$1; shift
PROBE=${*:-python tools/smc_wall_probe.py}
for i in 1 2; do
  echo -n "default  "; $PROBE
  echo -n "$V  "; KABC_LIB=$PWD/kissabc.jl_amd/lib/libkabc_hip_$V.so $PROBE
done

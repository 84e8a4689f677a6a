# A/B of the working tree's library against tools/_base.so with any probe: bash tools/ab_probe.sh <probe.py> [args]
for i in 1 2; do
  echo -n "new  "; python "$@" 2>/dev/null | tail -1
  echo -n "base "; KABC_LIB=$PWD/tools/_base.so python "$@" 2>/dev/null | tail -1
done

# A/B builds of the run-time-dimension AIS kernel alone: bash tools/ab_dyn.sh <name> "<extra flags>"
# compiles csrc/ais_dyn.hip with the flags and links it with the other objects of the default build into
# kissabc.jl_amd/lib/libkabc_hip_<name>.so (select with KABC_LIB=<path>).
set -e
cd "$(dirname "$0")/../kissabc.jl_amd/csrc"
V=$1; EXTRA=$2
mkdir -p build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. $EXTRA -c ais_dyn.hip -o build_ab/ais_dyn_$V.o
OBJS=$(ls build/*.o | grep -v "build/ais_dyn.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libkabc_hip_$V.so $OBJS build_ab/ais_dyn_$V.o -ldl
echo built ../lib/libkabc_hip_$V.so

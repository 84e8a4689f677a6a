# A/B build of the persistent smc loop kernel's workgroup size: bash tools/ab_smc_block.sh <name> "<extra flags>"
# compiles capi_smc.hip and smc_inst.hip for cost 3 (hier_gauss_sim, C4's) with the flags and links them with the
# other objects of the default build into kissabc.jl_amd/lib/libkabc_hip_<name>.so (KABC_LIB=<path>).  Only
# cost 3's smc kernels are consistent with the host code in such a library.
set -e
cd "$(dirname "$0")/../kissabc.jl_amd/csrc"
V=$1; EXTRA=$2
mkdir -p build_ab
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. $EXTRA"
/opt/rocm/bin/hipcc $F -c capi_smc.hip -o build_ab/capi_smc_$V.o &
/opt/rocm/bin/hipcc $F -DKABC_INST_COST=3 -c smc_inst.hip -o build_ab/smc_inst_3_$V.o &
wait
OBJS=$(ls build/*.o | grep -v "build/capi_smc.o" | grep -v "build/smc_inst_3.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libkabc_hip_$V.so $OBJS build_ab/capi_smc_$V.o build_ab/smc_inst_3_$V.o -ldl
echo built ../lib/libkabc_hip_$V.so

#!/bin/bash
# A/B builds of the HIP library that differ in proj.hip only: scripts/build_proj_variant.sh <name> [-D...]  ->  build/libhyena_<name>.so
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/var_$NAME
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/proj.hip -o $R/build/var_$NAME/proj.o
O=$R/hyena_dna_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $O/fftconv.hip.o $O/onchip.hip.o $O/onchip_dk.hip.o $O/cm.hip.o $R/build/var_$NAME/proj.o $O/filter16.hip.o -o $R/build/libhyena_$NAME.so
echo built build/libhyena_$NAME.so

#!/bin/bash
# A/B builds of the HIP library: scripts/build_variant.sh <name> [-D...]  ->  build/libhyena_<name>.so
# (fftconv.hip object is reused from the regular build; only onchip.hip, onchip_dk.hip and filter16.hip are recompiled unless FULL=1)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/var_$NAME
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/onchip.hip -o $R/build/var_$NAME/onchip.o &
/opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/onchip_dk.hip -o $R/build/var_$NAME/onchip_dk.o &
/opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/filter16.hip -o $R/build/var_$NAME/filter16.o &
if [ "$FULL" = "1" ]; then /opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/fftconv.hip -o $R/build/var_$NAME/fftconv.o & else cp $R/hyena_dna_amd/csrc/_obj/fftconv.hip.o $R/build/var_$NAME/fftconv.o; fi
if [ "$FULL" = "1" ]; then /opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/cm.hip -o $R/build/var_$NAME/cm.o & else cp $R/hyena_dna_amd/csrc/_obj/cm.hip.o $R/build/var_$NAME/cm.o; fi
if [ "$FULL" = "1" ]; then /opt/rocm/bin/hipcc $FL "$@" -c $R/hyena_dna_amd/csrc/proj.hip -o $R/build/var_$NAME/proj.o & else cp $R/hyena_dna_amd/csrc/_obj/proj.hip.o $R/build/var_$NAME/proj.o; fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $R/build/var_$NAME/fftconv.o $R/build/var_$NAME/onchip.o $R/build/var_$NAME/onchip_dk.o $R/build/var_$NAME/cm.o $R/build/var_$NAME/proj.o $R/build/var_$NAME/filter16.o -o $R/build/libhyena_$NAME.so
echo built build/libhyena_$NAME.so

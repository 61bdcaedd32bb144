#!/bin/bash
# The CPU suite under AddressSanitizer + UBSan (round 6).  GPU sanitizers are not available on this pool; the kernels' index arithmetic is the same code
# under -DHIPEMU (tests/hipemu), so a sanitized build of the emulator library checks every global / LDS access the emulated kernels make against the
# redzones of the tensors PyTorch allocated (libasan is preloaded: torch's posix_memalign goes through it).  ~10 min to build, ~15 min to run.
#   bash scripts/asan_emulator.sh            -> /tmp/asan_suite.log
set -e
R=$(cd $(dirname $0)/.. && pwd); H=$R/tests/hipemu; C=$R/hyena_dna_amd/csrc
mkdir -p $R/build/asan
g++ -x c++ -DHIPEMU -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize=vptr,alignment -fopenmp -fPIC -shared \
  -Wno-unknown-pragmas -Wno-psabi -I $H $C/fftconv.hip $C/onchip.hip $C/onchip_dk.hip $C/cm.hip $C/proj.hip $C/filter16.hip $H/hipemu.cpp -o $R/build/asan/emu_asan.so
cp $H/_emu_fftconv_test_only.so $R/build/asan/emu_plain_backup.so
cp $R/build/asan/emu_asan.so $H/_emu_fftconv_test_only.so
trap "cp $R/build/asan/emu_plain_backup.so $H/_emu_fftconv_test_only.so; touch $H/_emu_fftconv_test_only.so" EXIT
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=0:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python -m pytest tests/ -q -m "not gpu" -x > /tmp/asan_suite.log 2>&1 || true
tail -5 /tmp/asan_suite.log

#!/bin/bash
# Host-side sanitizer run (CPU build only): AddressSanitizer + UndefinedBehaviorSanitizer builds of
#   * the serial test driver tests/emul (the product's rule headers, automaton.cpp, energy_tables.cpp) and
#   * libelemdp.so with its HOST code instrumented (-fno-gpu-sanitize: device code as usual; GPU ASan does not exist on this pool),
# both with the ROCm clang so that they share one sanitizer runtime, then the CPU tests that reach them: the emulation against
# the oracle and the host-only entry points of the C ABI (automaton description, initial parameters, shuffles, permutations,
# error paths, the gloo all-reduce path).
#   bash tools/sanitize_cpu.sh [pytest args]        # exit code of pytest; reports on stderr
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$($CL -print-file-name=libclang_rt.asan-x86_64.so)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -shared-libsan"
mkdir -p $R/build/san
$CL -std=c++14 -O1 $SAN -fPIC -shared -Wno-unknown-pragmas -o $R/build/san/libelemdp_emul.so \
  $R/tests/emul/emul.cpp $R/rnaelem_amd/csrc/automaton.cpp $R/rnaelem_amd/csrc/energy_tables.cpp
cd $R/rnaelem_amd/csrc
for s in kernels.hip train_kernels.hip lin_kernels.hip bpp_kernels.hip; do   # (device code: no host logic worth instrumenting)
  [ $R/build/san/$s.o -nt $s ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -munsafe-fp-atomics -c $s -o $R/build/san/$s.o 2>/dev/null &
done
for s in engine.cpp automaton.cpp energy_tables.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC $SAN -fno-gpu-sanitize -c $s -o $R/build/san/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -fno-gpu-sanitize $R/build/san/*.o -o $R/build/san/libelemdp.so -ldl
cd $R
export ELEMDP_LIBRARY=$R/build/san/libelemdp.so ELEMDP_EMUL_LIBRARY=$R/build/san/libelemdp_emul.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ $# -eq 0 ]; then set -- tests/test_emul_vs_oracle.py tests/test_round4_cpu.py tests/test_host_abi.py tests/test_round3_cpu.py; fi
LD_PRELOAD=$RT python -m pytest -x -q -m "not gpu" -p no:cacheprovider "$@"

#!/bin/bash
# builds library variants build/var/lib_<name>.so from "name:flags" arguments (timed on the GPU by tools/try_variants.sh);
# a flag -O<x> replaces the default -O3; the compiler's messages of every variant are kept in build/var/<name>.log
cd "$(dirname "$0")/../rnaelem_amd/csrc"
mkdir -p ../../build/var && rm -f ../../build/var/lib_*.so ../../build/var/*.log
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  opt=-O3
  case "$f" in *-O2*) opt=-O2; f=${f/-O2/};; *-Os*) opt=-Os; f=${f/-Os/};; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $opt -std=c++17 -fPIC -shared -munsafe-fp-atomics $f kernels.hip train_kernels.hip lin_kernels.hip bpp_kernels.hip engine.cpp automaton.cpp energy_tables.cpp -o ../../build/var/lib_$n.so -ldl > ../../build/var/$n.log 2>&1 &
done
wait
for v in "$@"; do n=${v%%:*}; [ -f ../../build/var/lib_$n.so ] || { echo "variant $n did not build:"; tail -5 ../../build/var/$n.log; }; done
ls ../../build/var

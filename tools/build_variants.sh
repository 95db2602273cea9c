#!/bin/bash
# builds library variants build/var/lib_<name>.so from "name:flags" arguments (timed on the GPU by a lease script through
# ELEMDP_LIBRARY); per-file flags as rnaelem_amd/build.py; the compiler's messages of every variant are kept in build/var/<name>.log
cd "$(dirname "$0")/../rnaelem_amd/csrc"
mkdir -p ../../build/var && rm -f ../../build/var/lib_*.so ../../build/var/*.log
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  (
    mkdir -p ../../build/var/o_$n
    for src in kernels.hip train_kernels.hip lin_kernels.hip bpp_kernels.hip engine.cpp automaton.cpp energy_tables.cpp; do
      x=""; [ $src = lin_kernels.hip ] && x="-mllvm -disable-machine-licm"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $x $f -c $src -o ../../build/var/o_$n/$src.o >> ../../build/var/$n.log 2>&1 &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../build/var/o_$n/*.o -o ../../build/var/lib_$n.so -ldl >> ../../build/var/$n.log 2>&1
    rm -rf ../../build/var/o_$n
  ) &
done
wait
for v in "$@"; do n=${v%%:*}; [ -f ../../build/var/lib_$n.so ] || { echo "variant $n did not build:"; tail -5 ../../build/var/$n.log; }; done
ls ../../build/var

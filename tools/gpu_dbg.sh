#!/bin/bash
# ceilings: train pipeline time with phases switched off (option "dbg": 1 = no split sums, 2 = no item sums, 4 = no unary phase)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
for T in 0 1; do for D in 0 1 2 3 4 7; do
echo "== tile=$T dbg=$D"; timeout -k 10 120 python tools/run_eval.py 4096 200 2 4 0 tile=$T dbg=$D 2>&1 | tail -1 || exit 1
done; done > gpurun_out/dbg/ceil.log 2>&1
cat gpurun_out/dbg/ceil.log

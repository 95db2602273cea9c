#!/bin/bash
# two processes sharing the GPU, 2048 sequences each, against one process with 4096 (is the GPU left idle by one stream?)
# usage: two_proc.sh [option=value ...]   (options go to tools/run_eval.py, e.g. tile=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tp
timeout -k 10 200 python tools/run_eval.py 2048 200 24 4 0 "$@" > gpurun_out/tp/a.log 2>&1 &
P1=$!
timeout -k 10 200 python tools/run_eval.py 2048 200 24 4 0 "$@" > gpurun_out/tp/b.log 2>&1 &
P2=$!
wait $P1; wait $P2
echo "== two processes x 2048 $@"; grep pipeline gpurun_out/tp/a.log | sed -n '8,12p'; grep pipeline gpurun_out/tp/b.log | sed -n '8,12p'
timeout -k 10 200 python tools/run_eval.py 4096 200 3 4 0 "$@" > gpurun_out/tp/c.log 2>&1
echo "== one process x 4096 $@"; grep pipeline gpurun_out/tp/c.log | tail -2

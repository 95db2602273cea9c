#!/bin/bash
# round 4: how much of a scan (and of a train evaluation) runs with only the exterior chains / the traceback on the GPU
O=$GRAFT_REPO_ROOT/gpurun_out/r4ag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/skt -o run -- python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 10000 300 "(.....)" > $O/skt.log 2>&1 || { tail -3 $O/skt.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python3 $GRAFT_REPO_ROOT/tools/run_eval.py 10000 200 3 4 > $O/kt.log 2>&1 || { tail -3 $O/kt.log; exit 1; }
cd $GRAFT_REPO_ROOT
tail -1 $O/skt.log
echo "scan (load + 3 scans):"; python3 tools/kexposed.py $O/skt k5_cyk_ext k4_in_ext k4_out_ext k5_pick k4_r7; python3 tools/kexposed.py $O/skt k5_cyk_ext
echo "train (load + 3 evaluations):"; python3 tools/kexposed.py $O/kt k4_in_ext k4_out_ext k4_r7 k4_combine k_reduce k4_weights
rm -rf $O/skt $O/kt

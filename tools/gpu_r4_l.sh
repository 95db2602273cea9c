#!/bin/bash
# round 4: scan of 10 000 x L=300 with 2 (default), 3 and 4 groups at a time
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
cd $GRAFT_REPO_ROOT
for ns in 2 3 4 2 3; do
  timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" 0 0 $ns > $O/scan_$ns.txt 2>&1; echo "streams $ns: $(tail -1 $O/scan_$ns.txt)"
done

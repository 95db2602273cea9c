#!/bin/bash
# Round-4 evidence: full bench line, kernel trace of the bench command (one stream), HBM traffic counters ON THE BENCH CONFIGURATION
# (10 000 x L=200, default groups; separate --pmc passes, no tracing domains), the same for the scan (10 000 x L=300, (.....)).
# tools/pmc_traffic.py reads the number of evaluations / scans / loads of every pass from its own dispatch counts and refuses a
# per-kernel sum that exceeds the bench line's step.  Outputs under gpurun_out/r4ev/ -- the summaries are copied to profiles/.
O=$GRAFT_REPO_ROOT/gpurun_out/r4ev3
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- $B --serial-passes > $O/kt.log 2>&1; tail -1 $O/kt.log | cut -c1-200
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- $B > $O/fetch.log 2>&1; tail -1 $O/fetch.log | cut -c1-120
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- $B > $O/write.log 2>&1; tail -1 $O/write.log | cut -c1-120
S="python3 $GRAFT_REPO_ROOT/tools/scan_bench.py 10000 300 (.....)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/skt -o run -- $S 0 0 1 > $O/skt.log 2>&1; tail -1 $O/skt.log | cut -c1-200
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/sfetch -o run -- $S > $O/sfetch.log 2>&1; tail -1 $O/sfetch.log | cut -c1-200
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/swrite -o run -- $S > $O/swrite.log 2>&1; tail -1 $O/swrite.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/kt $O/kstats_bench10k.csv && head -8 $O/kstats_bench10k.csv
python tools/kstats.py $O/skt $O/kstats_scan.csv && head -10 $O/kstats_scan.csv
python tools/pmc_traffic.py $O/fetch $O/write 10000 $O/traffic.json --kt $O/kt --bench $O/bench.json --seq-len 200 --S 22 \
  --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (10 000 x L=200, default groups: the bench configuration); bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md), k4_* kernels + k_reduce; ms from the kernel trace of the same command with --serial-passes; evaluations counted from the k4_weights dispatches of each pass" | tail -12
python tools/pmc_traffic.py $O/sfetch $O/swrite 10000 $O/traffic.json --kt $O/skt --seq-len 300 --S 29 --pattern "(.....)" --scan --merge $O/traffic.json | tail -14
for d in kt fetch write skt sfetch swrite; do rm -rf $O/$d; done

#!/bin/bash
# round 4: A/B on ONE box at the bench size (10 000 x L=200): one block per workgroup against the default, equal against skewed groups
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  for cfg in "0 0" "1 0" "0 20" "1 20" "0 40" "1 40"; do
    set -- $cfg
    e="ELEMDP_NBLK=$1"; [ $1 = 0 ] && e="ELEMDP_DUMMY=1"
    s="ELEMDP_GROUP_SKEW=$2"; [ $2 = 0 ] && s="ELEMDP_DUMMY2=1"
    env $e $s timeout -k 10 200 $B > $O/b_$1_$2_$rep.json 2> $O/b_$1_$2_$rep.err || { echo "failed $cfg"; tail -3 $O/b_$1_$2_$rep.err; exit 1; }
    echo "nblk $1 skew $2 rep $rep: $(python -c "import json,sys; d=json.loads(open('$O/b_$1_$2_$rep.json').read().strip().split(chr(10))[-1]); print('%.1f ms %.0f seq/s fn %.9g' % (d['ms_per_step'], d['value'], d['config']['fn']))")"
  done
done

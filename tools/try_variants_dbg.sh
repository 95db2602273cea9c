#!/bin/bash
# phase knock-outs (tools/dbg_phases.py) on prebuilt library variants; args: dbg values
cd $GRAFT_REPO_ROOT
cp rnaelem_amd/libelemdp.so /tmp/keep.so
trap 'cp /tmp/keep.so rnaelem_amd/libelemdp.so' EXIT
for f in build/var/lib_*.so; do
  cp $f rnaelem_amd/libelemdp.so
  echo "== $f"
  timeout -k 10 200 python tools/dbg_phases.py 4096 200 "$@" 2>&1 | grep dbg || exit 1
done

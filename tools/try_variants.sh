#!/bin/bash
# timing of prebuilt library variants (build/var/lib_*.so): each is copied over rnaelem_amd/libelemdp.so on the GPU box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for f in build/var/lib_*.so; do
  cp $f rnaelem_amd/libelemdp.so
  echo "== $f" >> gpurun_out/var/log.txt
  timeout -k 10 120 python tools/sweep_group.py 4096 200 4 4096 >> gpurun_out/var/log.txt 2>&1 || exit 1
done
cat gpurun_out/var/log.txt

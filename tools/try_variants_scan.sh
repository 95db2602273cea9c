#!/bin/bash
# timing of prebuilt library variants (build/var/lib_*.so): each is copied over rnaelem_amd/libelemdp.so on the GPU box; the
# shipped library is put back whatever happens (trap).  VARGS: extra "option=value" arguments for tools/run_eval.py.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
rm -f gpurun_out/var/log.txt
cp rnaelem_amd/libelemdp.so /tmp/keep.so
trap 'cp /tmp/keep.so rnaelem_amd/libelemdp.so' EXIT
for f in build/var/lib_*.so; do
  cp $f rnaelem_amd/libelemdp.so
  echo "== $f" >> gpurun_out/var/log.txt
  timeout -k 10 120 python tools/scan_bench.py ${VN:-10000} 300 "(.....)" >> gpurun_out/var/log.txt 2>&1 || { echo "variant $f failed"; tail -5 gpurun_out/var/log.txt; exit 1; }
done
grep -v "^load" gpurun_out/var/log.txt | awk '/^==/{name=$2} /seq\/s/{last=$0} /^==/{if(prev)print prev; prev=""} {if($0 ~ /seq\/s/) prev=name" "$0} END{print prev}'

#!/bin/bash
# round 3, step g: scan parity on the GPU (traceback by re-derivation), then the scan rate
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "scan or Scan or parse or viterbi" > $O/pytest_scan.log 2>&1; rc=$?; echo "pytest scan rc $rc"; tail -5 $O/pytest_scan.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/scan_bench.py 10000 300 "(.....)" > $O/scan.log 2>&1; echo "scan rc $?"; tail -6 $O/scan.log

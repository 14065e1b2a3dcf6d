#!/bin/bash
# Dev helper (GPU box): bvh_check (pixels against the reference-order kernels and the oracle) and the default frame times of the five configurations.
set -e
timeout -k 10 200 python tools/bvh_check.py > gpurun_out/chk.log 2>&1
grep -c "differing pixels: 0" gpurun_out/chk.log
for s in hw14 hw11 hw12 hw08 hw07; do timeout -k 10 200 python tools/bvh_sweep.py $s "" "$1"; done

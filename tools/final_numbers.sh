#!/bin/bash
# Dev helper (GPU box): every number DESIGN.md quotes for the round, into gpurun_out/final_<tag>/ (one tool after the other; a failure stops the chain).
tag=${1:-r04}
out=gpurun_out/final_$tag
mkdir -p $out
export TMPDIR=/tmp
set -e
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 700 python tools/collect_profiles.py $tag > $out/collect.log 2>&1
timeout -k 10 200 python tools/trace_levels.py hw14 $tag > $out/trace_hw14.txt 2>&1
for s in hw14 hw11 hw12 hw08 hw07; do timeout -k 10 200 python tools/bvh_sweep.py $s "" >> $out/frames.txt 2>&1; done
timeout -k 10 300 python tools/rank_time.py hw14 1,2,4,8 > $out/rank_time.txt 2>&1
timeout -k 10 200 python tools/regrow_time.py > $out/regrow_time.txt 2>&1
timeout -k 10 200 python tools/bvh_exec.py hw14 > $out/bvh_exec.txt 2>&1
timeout -k 10 300 python tools/many_meshes.py 200 > $out/many_meshes.txt 2>&1
timeout -k 10 300 python tools/anim_time.py hw14 > $out/anim_time.txt 2>&1
echo done > $out/DONE

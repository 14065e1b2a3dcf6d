#!/bin/bash
# Dev helper (GPU box): kernel trace of a short bench run, summarised per launch of one frame.  usage: tools/trace_frame.sh <tag> [tuning] [scene]
R=$PWD; tag=$1; tuning="$2"; scene="${3:-hw14}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tr_$tag --output-format csv -- python3 $R/bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-alone --in-flight 0 --settle 0 --scene "$scene" --tuning "$tuning" > $R/gpurun_out/tr_$tag.log 2>&1 || exit 1
python3 $R/tools/frame_timeline.py $R/gpurun_out/tr_$tag > $R/gpurun_out/tl_$tag.txt

#!/bin/bash
# Kernel-trace timeline of one graph-replayed training step (scripts/trace_timeline.py) on the GPU box.
# usage: scripts/make_timeline.sh <tag> [bench args...]   -> gpurun_out/timeline_<tag>.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python3 bench.py --steps 20 --warmup 5 --no-secondary --no-roofline --no-cpu-baseline "$@" > gpurun_out/trace_$tag.log 2>&1
python3 scripts/trace_timeline.py gpurun_out/trace_$tag/*/*_kernel_trace.csv full > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/trace_$tag
head -8 gpurun_out/timeline_$tag.txt

#!/bin/bash
# Runs the default bench under rocprofv3 (kernel trace + stats) on the GPU box and leaves
#   gpurun_out/<tag>_kernel_stats.csv   per-kernel summary
#   gpurun_out/<tag>.json               the bench's JSON line of the same run
# usage: scripts/make_profile.sh <tag> [bench args...]      (copy the two files into profiles/ afterwards)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
grep '^{"' gpurun_out/prof_$tag.log | tail -1 > gpurun_out/$tag.json
cp gpurun_out/prof_$tag/*/*_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
head -c 300 gpurun_out/$tag.json; echo

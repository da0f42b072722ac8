#!/bin/bash
# A/B of this tree against another built tree (e.g. ab_old/, an earlier commit built in place) on one box:
#   scripts/ab_tree.sh <other tree> [bench args...]     each run twice, interleaved
other=$1; shift
run() { (cd $1 && timeout -k 10 200 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"); }
for rep in 1 2; do run . "$@" || exit 1; run $other "$@" || exit 1; done

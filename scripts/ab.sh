#!/bin/bash
# A/B on one box: committed tree in _ab/ (git worktree of HEAD, built) vs the working tree.
run() { (cd "$1" && timeout -k 10 200 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"); }
run _ab base || exit 1
run . new || exit 1
run _ab base || exit 1
run . new || exit 1

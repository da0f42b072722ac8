#!/bin/bash
# A/B of environment knobs on one box: scripts/ab_env.sh "VAR=a" "VAR=b" ...   (each setting run twice, interleaved)
run() { (env $1 timeout -k 10 200 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"); }
for rep in 1 2; do for s in "$@"; do run "$s" || exit 1; done; done

#!/bin/bash
# A/B of environment knobs on another workload: scripts/ab_env_wl.sh <workload> <steps> "VAR=a" "VAR=b" ...
wl=$1; steps=$2; shift 2
run() { (env $1 timeout -k 10 200 python -u bench.py --workload $wl --steps $steps --warmup 5 --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $1', d['value'], d['ms_per_step'])"); }
for rep in 1 2; do for s in "$@"; do run "$s" || exit 1; done; done

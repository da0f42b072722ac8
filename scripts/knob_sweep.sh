#!/bin/bash
# One box: step time under a few launch-shape knobs (tuning aid).
run() { env "$@" timeout -k 10 120 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])" || exit 1; }
run A=0
for b in 96 128 192 384 512 768; do run STMI_WG_BLOCKS=$b; done
run STMI_PC_CFG=1
run STMI_PC_CFG=2
run A=0

#!/bin/bash
# working tree under several environment settings, interleaved with the committed tree in _ab/
run() { (cd "$1" && env $3 timeout -k 10 200 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"); }
run _ab base X=1 || exit 1
run . new6 STMI_GRAD_BUFS=6 || exit 1
run . new8 STMI_GRAD_BUFS=8 || exit 1
run _ab base X=1 || exit 1
run . new6 STMI_GRAD_BUFS=6 || exit 1
run . new8 STMI_GRAD_BUFS=8 || exit 1

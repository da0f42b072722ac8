#!/bin/bash
# Step time with each side-stream branch folded back into the chain (STMI_SERIAL letters), one box.
for s in "" g w c gw gc wc gwc; do
  STMI_SERIAL=$s timeout -k 10 120 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial=[$s]', d['value'], d['ms_per_step'])" || exit 1
done

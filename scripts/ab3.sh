#!/bin/bash
# working tree under several environment settings (each "name:VAR=value"), interleaved, two rounds
run() { (cd "$1" && env $3 timeout -k 10 200 python -u bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"); }
for round in 1 2; do
  for cfg in "$@"; do
    name=${cfg%%:*}; var=${cfg#*:}
    dir=.; [ "$name" = base ] && dir=_ab
    run $dir $name $var || exit 1
  done
done

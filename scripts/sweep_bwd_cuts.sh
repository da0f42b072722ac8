#!/bin/bash
# Data-parallel bucket cuts: times the training bench on N GPUs for several STMI_BWD_CUTS settings (first block of
# every backward range but the last; one all-reduce bucket per range) and prints value, ms/step and the host-side
# figures of config.allreduce.  On a box with fewer GPUs than ranks, STMI_BENCH_REHEARSE=1 puts all ranks on GPU 0
# over gloo (the numbers then only show the host side).
#   scripts/sweep_bwd_cuts.sh <gpus> [cuts ...]        default cuts: "" 3 7 11,7 11,7,3
n=${1:-2}; shift
cuts=("$@"); [ ${#cuts[@]} -eq 0 ] && cuts=("" "3" "7" "11,7" "11,7,3")
for c in "${cuts[@]}"; do
  STMI_BWD_CUTS="$c" timeout -k 10 300 python -u bench.py --gpus $n --steps 50 --warmup 10 --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cuts=%-8r %8.1f img/s  %.4f ms/step  %s' % ('$c', d['value'], d['ms_per_step'], json.dumps(d['config'].get('allreduce'))))" || exit 1
done

#!/bin/bash
# HBM-side traffic per kernel launch, as rocprofv3 counters see it: two separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), a short bench run each.  Leaves
# gpurun_out/pmc_fetch/ and gpurun_out/pmc_write/; scripts/pmc_summary.py turns them into
# profiles/<tag>_pmc_traffic.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  rm -rf $d
  timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $d.log 2>&1
done
echo done

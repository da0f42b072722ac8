#!/bin/bash
# HBM-side traffic per kernel launch, as rocprofv3 counters see it: two separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), a short bench run each, and a third pass for MFMA activity.  Leaves
# gpurun_out/pmc_fetch/ and gpurun_out/pmc_write/; scripts/pmc_summary.py turns them into
# profiles/<tag>_pmc_traffic.json.
# usage: scripts/make_pmc.sh [bench args, e.g. --workload infer1080p]   (then scripts/pmc_summary.py <out.json>)
set -e
# The counters are chip-wide: with side-stream branches two kernels run at once and each is charged the other's
# traffic and MFMA cycles.  The counter passes therefore run the step with every branch folded into the chain.
export STMI_SERIAL=gwc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  rm -rf $d
  timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary "$@" > $d.log 2>&1
done
# MFMA activity of the same run: matrix-core busy cycles, MFMA operations (x512 = FLOP), chip-active cycles
d=gpurun_out/pmc_mfma
rm -rf $d
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary "$@" > $d.log 2>&1
echo done

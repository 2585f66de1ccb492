#!/bin/bash
# The two SQ counter passes of tools/profile_pmc.sh on their own, for comparing two builds of one kernel.
# Usage (GPU box, repo root): bash tools/exp/pmc_sq.sh <tag> <images>
set -u
TAG=$1; IMAGES=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
ARGS="--images $IMAGES --steps 1 --warmup 1 --no-cpu-baseline --no-sensitivity --h2h-sample 0"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU || exit 1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" && echo "summary written"

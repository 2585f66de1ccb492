#!/bin/bash
# Collects rocprofv3 evidence for the decode kernels on the GPU box: one kernel-trace pass and
# separate counter passes (counters are never combined with tracing, and every pass fits the
# per-block slot limits: 8 SQ, 4 TCC with FETCH_SIZE = 3 and WRITE_SIZE = 2).
# Usage (from the repo root, on the GPU box):  bash tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-run}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
IMAGES=${IMAGES:-256}
# ARGS_OVERRIDE replaces the default c3 arguments (e.g. "--config c5 --steps 1 --warmup 1 --no-cpu-baseline")
ARGS=${ARGS_OVERRIDE:-"--images $IMAGES --steps 1 --warmup 1 --no-cpu-baseline --no-sensitivity --h2h-sample 0 $*"}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -3 "$OUT/trace.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU || exit 1
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" && echo "summary written: $OUT/summary.json"
# TRAFFIC_KEY names the workload in the traffic table (bench.py looks its own workload up under the same key)
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/summary.json" "${TRAFFIC_KEY:-c3:$IMAGES}" "$ROOT/gpurun_out/traffic.json"

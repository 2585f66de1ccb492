#!/bin/bash
# rocprofv3 evidence for every benchmark configuration, on the GPU box: the full counter set for the headline
# batch (c3, 4096 images) and kernel traces (with counters where cheap) for c2, c4, c5 and c6.
# Usage (repo root, GPU box): bash tools/profile_all.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
IMAGES=4096 bash tools/profile_pmc.sh ${TAG}_c3 || exit 1
TRAFFIC_KEY=c2 ARGS_OVERRIDE="--config c2 --steps 3 --warmup 1 --no-cpu-baseline" bash tools/profile_pmc.sh ${TAG}_c2 || exit 1
TRAFFIC_KEY=c4:512/128 ARGS_OVERRIDE="--config c4 --images 512 --sub-batch 128 --steps 1 --warmup 1 --no-cpu-baseline" bash tools/profile_pmc.sh ${TAG}_c4 || exit 1
TRAFFIC_KEY=c5 ARGS_OVERRIDE="--config c5 --steps 1 --warmup 1 --no-cpu-baseline" bash tools/profile_pmc.sh ${TAG}_c5 || exit 1
TRAFFIC_KEY=c6 ARGS_OVERRIDE="--config c6 --steps 3 --warmup 1 --no-cpu-baseline" bash tools/profile_pmc.sh ${TAG}_c6 || exit 1
echo "all profiles done; copy gpurun_out/traffic.json to profiles/ and run tools/stamp_traffic.py in the build container"

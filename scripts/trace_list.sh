#!/bin/bash
# Per-LAUNCH table of the last step of a bench.py command (one stream), in launch order:
#   gpurun -- 'bash scripts/trace_list.sh <tag> [bench.py args...]'   ->  gpurun_out/<tag>_launches.txt
# rocprofv3 --kernel-trace only (no counters); the program sits directly after `--`.
set -euo pipefail
TAG=${1:?usage: trace_list.sh <tag> [bench args]}; shift
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/trace_$TAG"
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$TAG" -o t -- \
  python3 "$R/bench.py" "$@" --steps 2 --warmup 2 --no-cpu-baseline --no-profile --no-also --serial-streams > "$O/trace_$TAG.log" 2>&1
python3 "$R/scripts/trace_list.py" "$O/trace_$TAG" > "$O/${TAG}_launches.txt"
rm -rf "$O/trace_$TAG"
tail -3 "$O/${TAG}_launches.txt"

#!/bin/bash
# Per-round evidence for bench.py (run ON the GPU box: `gpurun -- 'bash scripts/round_profile.sh r02a'`):
#   gpurun_out/<tag>_bench.json               the bench line as the driver sees it (compact, `summary` last); <tag>_bench_detail.json = the complete record
#   gpurun_out/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc_traffic{,_bf16_b8,_infer_f16}.json
#                                             FETCH_SIZE / WRITE_SIZE / fabric read-request per-launch means of every kernel of the fp32
#                                             step, the bf16 batch-8 step and the fp16 forward (separate --pmc passes: the TCC slots
#                                             cannot hold them together; kernel-trace only, as the MI355X guide prescribes)
# Every pass writes into a directory removed beforehand, keeps its log next to the outputs, and the script stops at the
# first pass that fails or leaves no CSV -- nothing stale can be published.  The program sits directly after `--`.
set -euo pipefail
TAG=${1:?usage: round_profile.sh <tag>}
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp

python3 "$R/bench.py" --steps 10 --warmup 3 > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"
test -s "$O/${TAG}_bench.json" || { echo "bench.py printed nothing"; tail -5 "$O/${TAG}_bench.err"; exit 1; }
cut -c1-400 "$O/${TAG}_bench.json"
cp "$O/bench_detail.json" "$O/${TAG}_bench_detail.json"     # the complete record behind the compact line (every kernel, counter read-outs)

need() {  # need <dir> <file name>: the one CSV a pass must have produced, non-empty
  local f
  f=$(find "$1" -name "$2" -size +0 | head -1)
  test -n "$f" || { echo "missing or empty $2 under $1 (log: $1.log)"; tail -20 "$1.log"; exit 1; }
  echo "$f"
}

# kernel stats on ONE stream (--serial-streams): per-kernel launch durations that mean something; `roofline` in the bench line is
# measured the same way.  The default (multi-stream) run is captured too: its kernels overlap, so their durations add up to more
# than the step -- it is the evidence that the streams do overlap, not a per-kernel table.
rm -rf "$O/prof_$TAG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o bench -- \
  python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-also --serial-streams > "$O/prof_$TAG.log" 2>&1
cp "$(need "$O/prof_$TAG" bench_kernel_stats.csv)" "$O/${TAG}_bench_kernel_stats.csv"
rm -rf "$O/prof_${TAG}_ms"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${TAG}_ms" -o bench -- \
  python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-also > "$O/prof_${TAG}_ms.log" 2>&1
cp "$(need "$O/prof_${TAG}_ms" bench_kernel_stats.csv)" "$O/${TAG}_bench_multistream_kernel_stats.csv"

# the 16-bit configurations (BASELINE configs[2]: bf16 storage, batch 8; configs[4]: fp16 full-volume inference): their own bench
# lines and one-stream kernel tables
python3 "$R/bench.py" --dtype bf16 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-also > "$O/${TAG}_train_bf16_b8.json" 2> "$O/${TAG}_train_bf16_b8.err"
cut -c1-300 "$O/${TAG}_train_bf16_b8.json"
python3 "$R/bench.py" --infer --dtype f16 --steps 10 --warmup 3 --no-cpu-baseline --no-also > "$O/${TAG}_infer_f16.json" 2> "$O/${TAG}_infer_f16.err"
cut -c1-300 "$O/${TAG}_infer_f16.json"
rm -rf "$O/prof_${TAG}_bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${TAG}_bf16" -o bench -- \
  python3 "$R/bench.py" --dtype bf16 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-also --serial-streams > "$O/prof_${TAG}_bf16.log" 2>&1
cp "$(need "$O/prof_${TAG}_bf16" bench_kernel_stats.csv)" "$O/${TAG}_train_bf16_b8_kernel_stats.csv"
rm -rf "$O/prof_${TAG}_inf"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${TAG}_inf" -o bench -- \
  python3 "$R/bench.py" --infer --dtype f16 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-also --serial-streams > "$O/prof_${TAG}_inf.log" 2>&1
cp "$(need "$O/prof_${TAG}_inf" bench_kernel_stats.csv)" "$O/${TAG}_infer_f16_kernel_stats.csv"

# HBM-side counters, one rocprofv3 pass per counter group (the TCC block has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), for each of the
# three configurations of the default bench line.  Third pass: the raw fabric request counters FETCH_SIZE is derived from
# (TCC_EA0_RDREQ: all read requests; _32B / _64B / _128B: by request size) -- bench.py takes the read bytes of every kernel as
# 32 x R32 + 64 x R64 + 128 x R128 instead of doubling FETCH_SIZE (= RDREQ x 64 B on gfx950) across the board.
REQ="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"   # (all four exist on gfx950: rocprofv3 -L, round 4)
traffic_passes() {   # traffic_passes <suffix> <bench args...>
  local sfx=$1; shift
  # (GRBM_GUI_ACTIVE rides on the WRITE_SIZE pass -- the GRBM block has its own slots: busy cycles summed over the 8 XCDs, which over
  # the launch's duration in the same pass is the clock the part held under that kernel)
  # Fourth pass (round 6, review item 6): the matrix pipe's own busy counter next to the launch cycles of the SAME pass --
  # mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the counter-side check of bench.py's FLOP / time fraction
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE GRBM_GUI_ACTIVE" "req:$REQ" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    local name=${pass%%:*} ctr=${pass#*:}
    rm -rf "$O/pmc_${TAG}${sfx}_$name"
    # shellcheck disable=SC2086
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$O/pmc_${TAG}${sfx}_$name" -o "$name" -- \
      python3 "$R/bench.py" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-also --serial-streams > "$O/pmc_${TAG}${sfx}_$name.log" 2>&1
    need "$O/pmc_${TAG}${sfx}_$name" "${name}_counter_collection.csv" > /dev/null
  done
}
traffic_passes ""
traffic_passes "_bf16_b8" --dtype bf16 --batch 8
traffic_passes "_infer_f16" --infer --dtype f16

python3 "$R/scripts/pmc_aggregate.py" "$O" "$TAG"
python3 - "$O" "$TAG" <<'PY'
import csv, sys
O, TAG = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open('%s/%s_bench_kernel_stats.csv' % (O, TAG))))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step (7 steps in the trace): %.2f ms' % (tot / 7e6))
for r in rows[:28]:
    print('%-66s %5s x %8.1f us  %6.2f ms/step %5.1f%%' % (r['Name'][:66], r['Calls'], float(r['AverageNs']) / 1e3,
          float(r['TotalDurationNs']) / 7e6, 100 * float(r['TotalDurationNs']) / tot))
PY

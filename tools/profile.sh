#!/bin/bash
# Collect the rocprofv3 evidence bench.py / DESIGN.md quote, on the GPU box:
#   tools/profile.sh TAG        -> gpurun_out/prof_TAG/{stats,fetch,write}/..., summaries in gpurun_out/prof_TAG/
# Three runs of the same command, as gpurun requires: --kernel-trace --stats alone, then one --pmc pass per
# counter (FETCH_SIZE, WRITE_SIZE).  Copy the summaries you want to keep into profiles/ (tracked).
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu > "$OUT/stats.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o run -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu > "$OUT/fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o run -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu > "$OUT/write.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG"
tail -1 "$OUT/stats.log" | cut -c1-400

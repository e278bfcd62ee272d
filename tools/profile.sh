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
# rocprofv3's kernel trace serialises the copy stream's blit kernel with the compute stream and charges the copy's
# 2.4 ms to the K3 kernel queued behind each flush; with synchronous flushes every kernel is timed alone.
export BCE_HIP_SYNC_FLUSH=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --no-decode > "$OUT/stats.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o run -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-decode > "$OUT/fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o run -- \
  python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-decode > "$OUT/write.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG"
tail -1 "$OUT/stats.log" | cut -c1-400

#!/bin/bash
# Collect the rocprofv3 evidence bench.py / DESIGN.md quote, on the GPU box:
#   tools/profile.sh TAG   ->  gpurun_out/prof_TAG/: per-kernel stats of the four workloads (synth-text, natural corpus,
#   binary corpus, synth-rand), FETCH_SIZE / WRITE_SIZE sums of the synth-text run, K3's traffic summary.
# Separate runs, as gpurun requires: --kernel-trace --stats alone, then one --pmc pass per counter.
# Copy the summaries you want to keep into profiles/ (tracked).
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
python3 "$ROOT/tools/make_corpus.py" --out /tmp/bce_natural_100000000.bin --size 100000000 2> "$OUT/natural.corpus.log"
python3 "$ROOT/tools/make_binary_corpus.py" --out /tmp/bce_binary_100000000.bin --size 100000000 2> "$OUT/binary.corpus.log"
cd /tmp && export TMPDIR=/tmp
# rocprofv3's kernel trace serialises the copy stream's blit kernel with the compute stream and charges the copy's
# 2.4 ms to the K3 kernel queued behind each flush; with synchronous flushes every kernel is timed alone.
export BCE_HIP_SYNC_FLUSH=1
B="--steps 2 --warmup 1 --no-cpu --no-decode --no-workloads --no-e2e --no-stream --no-cli"   # 3 compressions per run
run() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o run -- \
    python3 "$ROOT/bench.py" $B "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats_${name}.csv" \;
  grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/${TAG}_bench_${name}.json"
  rm -rf "$OUT/$name"
  echo "$name done"
}
run synthtext_1e8
run natural_1e8 --file /tmp/bce_natural_100000000.bin
run binary_1e8 --file /tmp/bce_binary_100000000.bin
run synthrand_32Mi --workload synth-rand --size 33554432
# BASELINE configs[2] (enwik9-sized): 2 compressions of 10^9 bytes (--steps 1 --warmup 1 replaces $B's counts)
B="--steps 1 --warmup 1 --no-cpu --no-decode --no-workloads --no-e2e --no-stream --no-cli"
run synthtext_1e9 --size 1000000000
P="--steps 1 --warmup 0 --no-cpu --no-decode --no-workloads --no-e2e --no-stream --no-cli"   # ONE compression
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o run -- \
  python3 "$ROOT/bench.py" $P > "$OUT/fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o run -- \
  python3 "$ROOT/bench.py" $P > "$OUT/write.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG"
rm -rf "$OUT/fetch" "$OUT/write"
# the same two passes at 10^9 bytes (config 3 asks for the HBM figure at that size)
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_1e9" -o run -- \
  python3 "$ROOT/bench.py" $P --size 1000000000 > "$OUT/fetch_1e9.log" 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_1e9" -o run -- \
  python3 "$ROOT/bench.py" $P --size 1000000000 > "$OUT/write_1e9.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG" _1e9
rm -rf "$OUT/fetch_1e9" "$OUT/write_1e9"
# the GPU-assisted decoder on the three 10^8-byte workloads: 3 decodes of one archive each (kernel statistics only)
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
sys.path.insert(0, "$ROOT/tools")
import bce_amd
from scan_time import load
for kind in ("text", "natural", "binary"):
    open("/tmp/bce_prof_%s.bce" % kind, "wb").write(bce_amd.compress(load(kind)))
PY
for kind in text natural binary; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dec_$kind" -o run -- \
    python3 "$ROOT/tools/decode_archive_timing.py" /tmp/bce_prof_$kind.bce 3 > "$OUT/decode_$kind.log" 2>&1
  find "$OUT/dec_$kind" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats_decode_${kind}_1e8.csv" \;
  rm -rf "$OUT/dec_$kind"
  echo "decode $kind done"
done

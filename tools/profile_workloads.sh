#!/bin/bash
# Per-kernel rocprofv3 stats of bench.py on the four workloads (synth-text, natural corpus, binary corpus, synth-rand),
# on the GPU box:   tools/profile_workloads.sh TAG   ->  gpurun_out/prof_TAG/<workload>_kernel_stats.csv + bench lines
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
python3 "$ROOT/tools/make_corpus.py" --out /tmp/bce_natural_100000000.bin --size 100000000 2> "$OUT/natural.corpus.log"
python3 "$ROOT/tools/make_binary_corpus.py" --out /tmp/bce_binary_100000000.bin --size 100000000 2> "$OUT/binary.corpus.log"
cd /tmp && export TMPDIR=/tmp
export BCE_HIP_SYNC_FLUSH=1
run() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o run -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --no-decode "$@" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.json"
  rm -rf "$OUT/$name"
  echo "$name done"
}
run synthtext
run natural --file /tmp/bce_natural_100000000.bin
run binary --file /tmp/bce_binary_100000000.bin
run synthrand --workload synth-rand --size 33554432

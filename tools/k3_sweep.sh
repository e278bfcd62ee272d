#!/bin/bash
# K3 time of the binary and the natural corpus under the tail's environment knobs: tools/k3_sweep.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_binary_corpus.py --out /tmp/bce_binary_100000000.bin --size 100000000 >/dev/null 2>&1
python3 $ROOT/tools/make_corpus.py --out /tmp/bce_natural_100000000.bin --size 100000000 >/dev/null 2>&1
run() {  # label, env assignments...
  local label=$1; shift
  for w in binary natural; do
    r=$(env "$@" timeout -k 10 120 python3 $ROOT/tools/encode_file_timing.py /tmp/bce_${w}_100000000.bin 2>/dev/null | tail -1 | sed 's/.*k3 \([0-9.]*\) ms sha \([0-9a-f]*\).*/\1 \2/')
    printf "%-40s %-8s k3 %s\n" "$label" $w "$r"
  done
}
run base X=1
run uni_max=8192 BCE_HIP_UNI_MAX=8192
run uni_max=16384 BCE_HIP_UNI_MAX=16384
run uni_max=2048 BCE_HIP_UNI_MAX=2048
run lane_pass=128 BCE_HIP_DFS_LANE_PASS=128
run lane_pass=512 BCE_HIP_DFS_LANE_PASS=512
run pass=16384 BCE_HIP_DFS_PASS=16384
run pass=1024 BCE_HIP_DFS_PASS=1024
run local_from=8192 BCE_HIP_LOCAL_FROM=8192
run local_from=32768 BCE_HIP_LOCAL_FROM=32768
run local_from=65536 BCE_HIP_LOCAL_FROM=65536
run local_budget=128 BCE_HIP_LOCAL_BUDGET=128
run local_budget=256 BCE_HIP_LOCAL_BUDGET=256
run dfs_enter=2M BCE_HIP_DFS_ENTER=2000000
run dfs_enter=512K BCE_HIP_DFS_ENTER=524288
run no_help BCE_HIP_DFS_NO_HELP=1

#!/bin/bash
# SQ / TA / TCP counters of the K3 kernels (one compression), separate --pmc passes.  tools/pmc_k3.sh TAG [bench args]
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp BCE_HIP_SYNC_FLUSH=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-decode --no-workloads --no-e2e --no-stream --no-cli "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'P'
import csv,glob,sys,re,collections,json
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    seen=set()
    for row in csv.DictReader(open(f)):
        k=re.sub(r'\(.*','',row['Kernel_Name']).replace('bce::','').replace('void ','').strip()
        acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
json.dump(acc,open(out+'/summary.json','w'),indent=1)
keep={k:dict(v) for k,v in acc.items() if 'k3_' in k}
json.dump({'collected':'tools/pmc_k3.sh: rocprofv3 --pmc <set> --kernel-trace, five separate passes, ONE compression each (bench.py --steps 1 --warmup 0), BCE_HIP_SYNC_FLUSH=1; sums over all dispatches of the kernel','kernels':keep},open(out+'/k3_counters.json','w'),indent=1)
names=sorted({c for k in acc for c in acc[k]})
for k in sorted(acc,key=lambda k:-acc[k].get('SQ_WAVE_CYCLES',0))[:12]:
    print(k); print('   '+'  '.join('%s=%.4g'%(c,acc[k][c]) for c in names if c in acc[k]))
P
rm -rf $OUT/p[0-9]

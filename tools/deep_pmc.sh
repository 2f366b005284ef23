#!/bin/bash
# instruction-cache behaviour of the persistent deep-level kernel (rocprofv3 PMC pass, counters in their own run)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_ic
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY} -d /tmp/pmc_ic -o ic --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-graph > /tmp/pmc_ic.log 2>&1
f=$(find /tmp/pmc_ic -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.Counter())
n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k,c in sorted(agg.items(), key=lambda kv:-kv[1]['SQ_WAVE_CYCLES'])[:6]:
    print(k, n[k], {a:int(b/max(n[k],1)) for a,b in c.items()})
PY

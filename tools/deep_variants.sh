#!/bin/bash
# headline bench with variant libraries prebuilt into tools/microbench/variants/ (timing experiments on the persistent kernel)
for v in "$@"; do
  r=$(JEN1_LIB=$PWD/tools/microbench/variants/libjen1_$v.so timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(d['value'], d['ms_per_step'], 'deep launch us', r.get('avg_launch_us'), 'us/phase', r.get('us_per_phase'))")
  echo "$v: steps/s, ms/step = $r"
done

#!/bin/bash
# timing experiments on the persistent deep-level kernel: builds variants of the library (timing-only switches, results of some
# are garbage) and runs the headline bench with each.  Usage (on the GPU box): bash tools/deep_experiments.sh "<name>:<flags>" ...
cd "$(dirname "$0")/.."
CS=jen-1-pytorch_amd/csrc
SRCS=$(python -c "import sys; sys.path.insert(0,'jen-1-pytorch_amd'); from jen1_amd import lib; print(' '.join('$CS/'+s for s in lib.SOURCES))")
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  out=gpurun_out/libjen1_exp_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -Iinclude -I$CS $SRCS -o $out || { echo "$name: build failed"; continue; }
  r=$(JEN1_LIB=$PWD/$out timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get(\"roofline\",{}); print(d[\"value\"], d[\"ms_per_step\"], \"deep launch us\", r.get(\"avg_launch_us\"), \"us/phase\", r.get(\"us_per_phase\"))")
  echo "$name [$flags]: steps/s, ms/step = $r"
done

# tuning sweep of the persistent launch (GPU box): variant builds via -D flags (JEN1_LIB) and engine knobs via the environment
# usage: bash tools/deep_sweep.sh "<-D flags A>" "<-D flags B>" ...   (an entry starting with ENV: sets environment variables instead)
cd $GRAFT_REPO_ROOT
CS=jen-1-pytorch_amd/csrc
SRC="$CS/conv_gemm.hip $CS/stream_gemm.hip $CS/tile_gemm.hip $CS/norm_apply.hip $CS/attention.hip $CS/deep_kernel.hip $CS/elementwise.hip $CS/optimizer.hip $CS/train_gemm.hip $CS/train_ops.hip $CS/train_attn.hip $CS/encodec.hip"
i=0
for v in "$@"; do
  i=$((i+1))
  if [[ "$v" == ENV:* ]]; then
    out=$(env ${v#ENV:} python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1)
  else
    lib=gpurun_out/libsweep_$i.so
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -Iinclude -I$CS $SRC -o $lib || { echo "build failed: $v"; continue; }
    out=$(JEN1_LIB=$PWD/$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1)
    rm -f $lib
  fi
  echo "$v => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], "steps/s; deep", r.get("avg_launch_us"), "us,", r.get("us_per_phase"), "us/phase; long", r.get("long_levels",{}).get("conv_ms_per_step"))' 2>&1 | tail -1)"
done

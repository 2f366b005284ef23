#!/bin/bash
# Tuning tool: headline bench (configs[1]) for a sweep of the tile-kernel workgroup target (engine.tile_target_wgs).
#   bash tools/tile_sweep.sh > gpurun_out/tile_sweep.txt
cd "$(dirname "$0")/.."
for tgt in 256 384 512 768 1024; do
  for one in 0 1; do
    out=$(JEN1_TILE_TARGET_WGS=$tgt JEN1_TILE_ONE_ROUND=$one python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | tail -1)
    echo "target=$tgt one_round=$one $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], "deep_us", r["avg_launch_us"], "long_us_per_launch", r["long_levels"]["avg_launch_us"], "long_ms", r["long_levels"]["conv_ms_per_step"])')"
  done
done

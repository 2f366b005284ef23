run() { out=$(env "$@" python bench.py --steps 100 --warmup 10 --repeats 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1); echo "$* $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; l=r["long_levels"]; print(d["value"], d["ms_per_step"], "deep_us", r["avg_launch_us"], "long_ms", l["conv_ms_per_step"], l["avg_launch_us"], [x[1] for x in l["slowest_launches_us"]])')"; }
for t in 96 128 160 192 224 256; do run JEN1_TILE_TARGET_WGS=$t; done
for t in 128 192; do run JEN1_TILE_TARGET_WGS=$t JEN1_TILE_ONE_ROUND=1; done

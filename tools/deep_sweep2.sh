#!/bin/bash
# Tuning tool: headline bench for a sweep of the persistent launch's unit target / weight re-read cap (engine.deep_unit_target, deep_reread_cap)
cd "$(dirname "$0")/.."
run() { out=$(env "$@" python bench.py --steps 100 --warmup 10 --repeats 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1); echo "$* $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], "deep_us", r["avg_launch_us"], "phases", r["phases"])')"; }
run JEN1_DEEP_UNIT_TARGET=128
run JEN1_DEEP_UNIT_TARGET=64
run JEN1_DEEP_UNIT_TARGET=96
run JEN1_DEEP_UNIT_TARGET=160
run JEN1_DEEP_UNIT_TARGET=192
run JEN1_DEEP_UNIT_TARGET=256
run JEN1_DEEP_REREAD_MB=8
run JEN1_DEEP_REREAD_MB=32
run JEN1_DEEP_REREAD_MB=64
run JEN1_DEEP_NB_MAX=4
run JEN1_DEEP_NB_MAX=2

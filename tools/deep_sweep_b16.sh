# unit target / re-read cap / batch split of the persistent launch at 16 rows per step (the CFG pair of configs[2])
run() { out=$(env "$@" python bench.py --batch 16 --steps 100 --warmup 10 --repeats 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1); echo "$* $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], "deep_us", r["avg_launch_us"], "phases", r["phases"])')"; }
run A=0
run JEN1_DEEP_UNIT_TARGET=64
run JEN1_DEEP_UNIT_TARGET=96
run JEN1_DEEP_UNIT_TARGET=192
run JEN1_DEEP_UNIT_TARGET=256
run JEN1_DEEP_UNIT_TARGET=64 JEN1_DEEP_REREAD_MB=8
run JEN1_DEEP_UNIT_TARGET=128 JEN1_DEEP_REREAD_MB=8
run JEN1_DEEP_UNIT_TARGET=128 JEN1_DEEP_REREAD_MB=32

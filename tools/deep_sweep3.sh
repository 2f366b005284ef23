run() { out=$(env "$@" python bench.py --steps 100 --warmup 10 --repeats 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1); echo "$* $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], "deep_us", r["avg_launch_us"], "phases", r["phases"])')"; }
run A=0
run JEN1_DEEP_CAP_BEFORE_ATTN=128
run JEN1_DEEP_CAP_BEFORE_ATTN=192
run JEN1_DEEP_UNIT_CAP=192

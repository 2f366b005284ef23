run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])"; }
echo "default:"; run
echo "cap 128:"; JEN1_DEEP_UNIT_CAP=128 run
echo "cap 192:"; JEN1_DEEP_UNIT_CAP=192 run
echo "cap 64:"; JEN1_DEEP_UNIT_CAP=64 run

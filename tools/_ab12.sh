run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
echo "shards:"; run
export JEN1_LIB=$PWD/.alt/libjen1_flags.so
timeout 600 python -m pytest tests/test_gpu_deep.py -x -q 2>&1 | tail -2
echo "flags:"; run
echo "flags:"; run

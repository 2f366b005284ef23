run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['long_levels']['conv_ms_per_step'])"; }
timeout 600 python -m pytest tests/test_gpu_deep.py -x -q 2>&1 | tail -2
echo "new:"; run
echo "new:"; run

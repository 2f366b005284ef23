run() { python bench.py --no-extra --no-cpu-baseline $@ 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['launches_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('phases'))"; }
timeout 900 python -m pytest tests/test_gpu_deep.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
python tools/deep_units.py --batch 1 --length 9000 --cfg 2>/dev/null | head -2 | cut -c1-600
echo "B8:"; run
echo "T9000 B1 (no cfg):"; run --batch 1 --length 9000

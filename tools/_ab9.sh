run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])"; }
timeout 600 python -m pytest tests/test_gpu_deep.py -x -q 2>&1 | tail -1
echo "sleep 8:"; run
for sl in 3 16; do echo "sleep $sl:"; JEN1_LIB=$PWD/.alt/libjen1_sl$sl.so run; done
echo "sleep 8:"; run

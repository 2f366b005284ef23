run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])"; }
timeout 600 python -m pytest tests/test_gpu_deep.py -x -q 2>&1 | tail -2
echo "NPOLL=3:"; run
echo "NPOLL=1:"; JEN1_LIB=$PWD/.alt/libjen1_np1.so run
echo "NPOLL=2:"; JEN1_LIB=$PWD/.alt/libjen1_np2.so run
echo "NPOLL=4:"; JEN1_LIB=$PWD/.alt/libjen1_np4.so run
echo "NPOLL=3:"; run

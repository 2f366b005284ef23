run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])"; }
echo "sleep 1:"; run
for sl in 0 3 8 20; do echo "sleep $sl:"; JEN1_LIB=$PWD/.alt/libjen1_sl$sl.so run; done
echo "sleep 1:"; run

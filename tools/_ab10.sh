run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
echo "shards 8:"; run
for sh in 16 32 48; do echo "shards $sh:"; JEN1_LIB=$PWD/.alt/libjen1_sh$sh.so run; done
echo "shards 8:"; run

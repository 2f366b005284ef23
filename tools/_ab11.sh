run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
echo "32 shards, 256 B apart:"; JEN1_LIB=$PWD/.alt/libjen1_sh32.so run
echo "32 shards, 128 B apart:"; JEN1_LIB=$PWD/.alt/libjen1_sw32.so run
echo "32 shards, 64 B apart:"; JEN1_LIB=$PWD/.alt/libjen1_sw16.so run

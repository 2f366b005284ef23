run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['long_levels']['conv_ms_per_step'], d['roofline']['long_levels']['launches_per_step'])"; }
echo "new fused:"; run
echo "new unfused:"; JEN1_FUSE_SHORTCUT_TILES=0 run
echo "HEAD lib unfused:"; JEN1_FUSE_SHORTCUT_TILES=0 JEN1_LIB=$PWD/.alt/libjen1_head.so run
echo "new fused:"; run

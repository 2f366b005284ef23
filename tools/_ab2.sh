run() { python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['long_levels']['conv_ms_per_step'])"; }
echo "default:"; run
echo "TARGET_WGS=192:"; JEN1_TILE_TARGET_WGS=192 run
echo "ONE_ROUND:"; JEN1_TILE_ONE_ROUND=1 run
echo "TARGET_WGS=128:"; JEN1_TILE_TARGET_WGS=128 run
echo "default:"; run

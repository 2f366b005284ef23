run() { python bench.py --no-extra --no-cpu-baseline --batch 1 --length 9000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['roofline']['long_levels']; print(d['value'], d['roofline']['avg_launch_us'], l['conv_ms_per_step'], l['launches_per_step'], l['slowest_launches_us'][0])"; }
echo "default:"; run
echo "ONE_ROUND:"; JEN1_TILE_ONE_ROUND=1 run
echo "TARGET 512:"; JEN1_TILE_TARGET_WGS=512 run
echo "TARGET 384:"; JEN1_TILE_TARGET_WGS=384 run
echo "TARGET 192:"; JEN1_TILE_TARGET_WGS=192 run

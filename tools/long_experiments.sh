# removal experiments on the sample-resident long-level launches (tuning builds: results of those builds are garbage, the clock is not).
# build here (container):  for v in NOWAIT NOMMA NOSILU; do JEN1_LIB=$PWD/jen-1-pytorch_amd/jen1_amd/libvar_$v.so JEN1_HIPCC_FLAGS=-DJEN1_LONG_EXP_$v \
#     PYTHONPATH=jen-1-pytorch_amd python -c "from jen1_amd import lib; lib.build()"; done
# run on the GPU box:      bash tools/long_experiments.sh > gpurun_out/long_experiments.txt
cd $GRAFT_REPO_ROOT
run() {
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
l = d['roofline']['long_levels']
print('%-34s %8.1f steps/s  %7.4f ms/step   long levels: %6.4f ms per step (%d phases, %5.2f us per phase incl. the poisoning node)' % (sys.argv[1], d['value'], d['ms_per_step'], l['conv_ms_per_step'], l['phases'], l['conv_ms_per_step'] * 1e3 / max(1, l['phases'])))" "$1"
}
echo "# B = 8, T = 1500, bf16, hipGraph-replayed step; product build first, then builds with one stage removed (their results are garbage)"
run "product build"
for v in NOWAIT NOMMA NOSILU NOW; do
  L=$PWD/jen-1-pytorch_amd/jen1_amd/libvar_$v.so
  [ -f $L ] && JEN1_LIB=$L run "-DJEN1_LONG_EXP_$v"
done
JEN1_LONG_LOCAL=1 run "product, XCD-local stores"
JEN1_PACK_FIXED_ORDER=0 run "product, pack sums by atomics"
JEN1_LONG=0 run "one launch per layer (r05 path)"

# round profile: kernel stats + per-step breakdown + PMC passes (HBM traffic, MFMA busy) of the default bench command.
# run on the GPU box:  bash tools/profile_round.sh [tag]     outputs under gpurun_out/profile (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/profile; mkdir -p $O
B="python bench.py --no-extra --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format rocpd -- $B --steps 20 --warmup 5 > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --top 24 > $O/${TAG}_kernel_stats_bf16_B8_T1500.txt 2>&1
python tools/rocpd_step.py $(find $O/trace -name "*.db" | head -1) 12 40 > $O/${TAG}_step_breakdown_bf16_B8_T1500.txt 2>&1
ARGS=""
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format rocpd -- $B --steps 8 --warmup 3 > $d.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  for n in $c; do ARGS="$ARGS $n=$db"; done
done
python tools/pmc_summary.py $O/${TAG}_pmc.json $ARGS > $O/${TAG}_pmc.txt 2>&1
cat $O/${TAG}_pmc.txt | head -12
head -30 $O/${TAG}_kernel_stats_bf16_B8_T1500.txt
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES

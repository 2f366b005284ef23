cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/profile; mkdir -p $O
JEN1_BENCH_NO_CPU=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format rocpd -- python bench.py --steps 20 --warmup 5 --no-extra > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) --top 24 > $O/kernel_stats.txt 2>&1
python tools/rocpd_step.py $(find $O/trace -name "*.db" | head -1) 12 40 > $O/step_breakdown.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  JEN1_BENCH_NO_CPU=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p --output-format rocpd -- python bench.py --steps 8 --warmup 3 --no-extra > $O/pmc_$c.log 2>&1
  python tools/rocpd_pmc.py $(find $O/pmc_$c -name "*.db" | head -1) $c 3 >> $O/pmc.txt 2>&1
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400
cat $O/pmc.txt | head -30
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE

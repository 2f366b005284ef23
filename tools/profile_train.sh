# training profile: kernel trace of tools/train_step.py (replayed pass + optimiser step), summarised over the steady-state iterations
# run on the GPU box:  bash tools/profile_train.sh [tag]     outputs under gpurun_out/profile_train (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/profile_train; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t --output-format rocpd -- python tools/train_step.py --iters 6 > $O/run.log 2>&1
DB=$(find $O/tr -name "*.db" | head -1)
python tools/rocpd_stats.py $DB --top 60 --between adamw_kernel > $O/${TAG}_train_step_kernel_stats.txt 2>&1
python tools/rocpd_sequence.py $DB adamw_kernel --list > $O/${TAG}_train_step_sequence.txt 2>&1
head -12 $O/${TAG}_train_step_kernel_stats.txt; head -4 $O/${TAG}_train_step_sequence.txt | cut -c1-400
rm -rf $O/tr

mkdir -p gpurun_out/r2f
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest.log 2>&1; tail -3 gpurun_out/r2f/pytest.log
timeout 900 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; tail -c 1200 gpurun_out/r2f/bench.json
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r2f/train.json 2> gpurun_out/r2f/train.err; cat gpurun_out/r2f/train.json | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2f/smoke.log 2>&1; tail -2 gpurun_out/r2f/smoke.log
bash tools/profile_round.sh r02 > gpurun_out/r2f/profile.log 2>&1; tail -20 gpurun_out/r2f/profile.log

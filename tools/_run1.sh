mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 900 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -c 3000 gpurun_out/r2a/bench.json; tail -5 gpurun_out/r2a/bench.err
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r2a/train.json 2> gpurun_out/r2a/train.err; cat gpurun_out/r2a/train.json; tail -5 gpurun_out/r2a/train.err

#!/bin/bash
# soak variants of the GPU suite (one fresh lease): every tensor its own hipMalloc (an out-of-range access faults instead of landing in a
# neighbour of the caching allocator's block), serialised kernels, reversed file order.  Logs under gpurun_out/soak_*.log
mkdir -p gpurun_out
export JEN1_TEST_BREADCRUMB_STDERR=0
run() { name=$1; shift; echo "== $name"; ( "$@" ) > gpurun_out/soak_$name.log 2>&1; echo "rc=$? ($name)"; grep -E "passed|failed|error" gpurun_out/soak_$name.log | tail -3; }
run reversed timeout 1500 python3 -m pytest $(ls -r tests/test_*.py) -q -m gpu -p no:cacheprovider
run nocache env PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 2400 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider
run serialize env AMD_SERIALIZE_KERNEL=3 timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider

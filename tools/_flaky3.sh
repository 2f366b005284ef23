cd .alt/r1
n=0; for i in $(seq 1 12); do timeout 120 python -m pytest tests/test_gpu_model.py -q -k "independent_batches" 2>&1 | grep -q "1 passed" || n=$((n+1)); done; echo "r1 independent_batches: $n failures of 12"
for i in 1 2 3; do timeout 700 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed"; done

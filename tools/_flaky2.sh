t() { n=0; for i in $(seq 1 12); do timeout 120 python -m pytest tests/test_gpu_model.py -q -k "independent_batches" 2>&1 | grep -q "1 passed" || n=$((n+1)); done; echo "$1: $n failures of 12"; }
t default
JEN1_FUSE_SHORTCUT_TILES=0 t nofuse
JEN1_LIB=$PWD/.alt/libjen1_head.so JEN1_FUSE_SHORTCUT_TILES=0 t headlib
JEN1_DEEP=0 t nodeep

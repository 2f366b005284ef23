#!/bin/bash
# timing experiments on big_gemm_nt256_kernel (tuning tool): variant libraries built by hand into tools/microbench/variants/
for v in "$@"; do
  echo "== $v"
  JEN1_LIB=tools/microbench/variants/libjen1_$v.so python tools/big_gemm_bench.py 2>&1 | grep -E "4096\^3|set_context: 13|8192" | sed 's/hipBLASLt.*//'
done

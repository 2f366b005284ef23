#!/bin/bash
# configs[4]-shaped plans (B = 1, T = 9000) with and without column chunks (JEN1_DEEP_CHUNKS build, JEN1_DEEP_MAX_LEN)
run() { python - "$@" <<'PY'
import json, os, sys, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "jen-1-pytorch_amd")]
import bench
from jen1_amd.config import full_model_config
from jen1_amd.model import UNetCFG1d
m = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")
for cfgp in (False, True):
    st = bench.build_stepper(m, 1, 9000, "cuda", cfg_pair=cfgp, use_graph=True)
    dt = bench.timed_steps(st, 40, 10, lambda: None)
    st.check()
    print(f"   cfg_pair={cfgp}: {40 / dt:7.1f} steps/s, {st.plan.n_launch + 1} launches, deep phases {len(st.plan.deep) if st.plan.deep is not None else 0}, deep_level {st.plan.deep_level}, errors {st.plan.deep_errors[:2]}")
PY
}
echo "== default library, max_len 64"; run
echo "== chunks library, max_len 64"; JEN1_LIB=$PWD/tools/microbench/variants/libjen1_CHUNKS.so run
echo "== chunks library, max_len 80"; JEN1_LIB=$PWD/tools/microbench/variants/libjen1_CHUNKS.so JEN1_DEEP_MAX_LEN=80 run
echo "== chunks library, max_len 144"; JEN1_LIB=$PWD/tools/microbench/variants/libjen1_CHUNKS.so JEN1_DEEP_MAX_LEN=144 run

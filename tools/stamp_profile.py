#!/usr/bin/env python3
"""stamp a profile JSON (profiles/rNN_pmc.json, rNN_train_pmc.json) with the commit and the bench.py hash it was collected at
(the GPU box has no .git: run here, after copying the summaries from gpurun_out/, BEFORE committing them)

    python tools/stamp_profile.py profiles/r03_pmc.json "bash tools/profile_round.sh r03"
"""
import hashlib
import json
import subprocess
import sys

path, cmd = sys.argv[1], sys.argv[2]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "jen-1-pytorch_amd", "bench.py", "include"], capture_output=True, text=True).stdout.strip()
d = json.load(open(path))
d["collected_at"] = {"repo_head": head + ("+dirty" if dirty else ""), "bench_py_sha16": hashlib.sha256(open("bench.py", "rb").read()).hexdigest()[:16],
                     "command": cmd}
json.dump(d, open(path, "w"), indent=1)
print(path, d["collected_at"])

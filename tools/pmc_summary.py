#!/usr/bin/env python3
"""Per-kernel-family PMC totals of one replayed denoiser step -> profiles/r02_pmc.json (read by bench.py's roofline block).

usage: pmc_summary.py OUT.json  COUNTER=results.db [COUNTER=results.db ...]  [--step 3]
Each results.db is one rocprofv3 --pmc pass (rocpd format) of `python bench.py --steps 8 --warmup 3 --no-extra
--no-cpu-baseline`; counters are collected in separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do
not fit one pass).  Corrections applied here and recorded in the output:
  FETCH_SIZE (KiB)  x 2   -- gfx950 reports half of the bytes of wide coalesced streaming reads (the guide's HBM section);
  WRITE_SIZE (KiB)  x 1   -- uncalibrated, taken as is;
  MFMA busy = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4)   (rocprofiler-sdk's MfmaUtil expression;
  SQ_VALU_MFMA_BUSY_CYCLES summed over the chip, GRBM_GUI_ACTIVE averaged over its per-XCD instances).
"""
import collections
import json
import sqlite3
import os
import sys


def family(n):
    """kernel family of a (demangled or mangled) kernel name"""
    if "deep_kernel" in n:
        return "deep_kernel"
    if "long_kernel" in n:
        return "long_kernel"
    if "conv_gemm" in n or "stream_gemm" in n or "tile_gemm" in n or "TileArgs" in n:
        return "conv_family"
    if "norm_apply" in n:
        return "norm_apply"
    if "attention" in n:
        return "attention"
    if n.startswith("_Z"):                  # a mangled name: the length-prefixed identifier that names the kernel
        ids, i = [], 2
        while i < len(n):                   # <length><identifier> pieces, whatever stands between them
            if n[i].isdigit():
                j = i
                while j < len(n) and n[j].isdigit():
                    j += 1
                k = int(n[i:j])
                ids.append(n[j:j + k])
                i = j + k
            else:
                i += 1
        ids = [x for x in ids if "kernel" in x] or ids
        if ids:
            return ids[0]
    # the function's own name: what stands in front of the argument list, without return type, namespaces and template arguments
    # ("void (anonymous namespace)::poison_kernel(...)" -> "poison_kernel"; "void at::native::foo<float>(...)" -> "foo")
    head = n
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):              # the first "(" outside template brackets that is not "(anonymous namespace)"
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not n.startswith("(anonymous namespace)", i):
            cut = i
            break
    head = n[:cut].strip()
    out, depth = [], 0
    for ch in head:                         # drop template argument lists
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif depth == 0:
            out.append(ch)
    head = "".join(out).strip()
    head = head.split(" ")[-1] if " " in head else head
    head = head.split("::")[-1]
    return head or n.strip()[:40] or "unnamed"


def per_step(db, counter, which):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    ipcols = [r[1] for r in c.execute(f"pragma table_info({ip})")]
    namecol = "name" if "name" in ipcols else ipcols[8]
    ids = [r[0] for r in c.execute(f"select id from {ip} where {namecol}=?", (counter,))]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    # GRBM_GUI_ACTIVE is reported once per XCD: the cycle count of a dispatch is the mean over its instances, not their sum
    agg_fn = "avg" if counter == "GRBM_GUI_ACTIVE" else "sum"
    q = f"""select s.{name_col}, d.start, {agg_fn}(p.value) from {kd} d join {ks} s on d.kernel_id = s.id
            join {pe} p on p.event_id = d.event_id where p.pmc_id in ({','.join(str(i) for i in ids)})
            group by d.id order by d.start"""
    rows = list(c.execute(q))
    # a replayed step is delimited by its tail launch (round 6: jen1_step_tail), else by pack_input at its head
    tails = [i + 1 for i, r in enumerate(rows) if "step_tail" in r[0]]
    idx = tails if len(tails) > which + 1 else [i for i, r in enumerate(rows) if "pack_input" in r[0]]
    a = idx[which]
    b = idx[which + 1] if which + 1 < len(idx) else len(rows)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, _, v in rows[a:b]:
        f = family(n)
        agg[f][0] += 1
        agg[f][1] += v
    return agg, b - a


def main():
    out = sys.argv[1]
    which = 3
    passes = {}
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--step":
            which = int(args.pop(0))
        else:
            k, v = a.split("=", 1)
            passes[k] = v
    data = {k: per_step(v, k, which) for k, v in passes.items()}
    fams = sorted({f for agg, _ in data.values() for f in agg})
    import hashlib
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("JEN1_HEAD", "")
    except OSError:
        head = os.environ.get("JEN1_HEAD", "")
    bench_sha = hashlib.sha256(open(os.path.join(root, "bench.py"), "rb").read()).hexdigest()[:16]
    res = {"source": "rocprofv3 --kernel-trace --pmc <one counter set per pass>, `python bench.py --steps 8 --warmup 3 --no-extra "
                     "--no-cpu-baseline`, one replayed step (tools/profile_round.sh, tools/pmc_summary.py)",
           "collected_at": {"utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "head": head, "bench_py_sha16": bench_sha},
           "corrections": {"FETCH_SIZE": "KiB x 2 (gfx950 wide streaming reads are tallied at half)", "WRITE_SIZE": "KiB as reported"},
           "dispatches_per_step": {k: n for k, (_, n) in data.items()}, "kernels": {}}
    for f in fams:
        e = {}
        n = max((data[k][0][f][0] for k in data if f in data[k][0]), default=0)
        e["launches_per_step"] = n
        if "FETCH_SIZE" in data and f in data["FETCH_SIZE"][0]:
            e["fetch_kib_raw"] = round(data["FETCH_SIZE"][0][f][1], 1)
        if "WRITE_SIZE" in data and f in data["WRITE_SIZE"][0]:
            e["write_kib_raw"] = round(data["WRITE_SIZE"][0][f][1], 1)
        if "fetch_kib_raw" in e or "write_kib_raw" in e:
            e["hbm_bytes_per_launch"] = int((2.0 * e.get("fetch_kib_raw", 0.0) + e.get("write_kib_raw", 0.0)) * 1024 / max(n, 1))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in data and "GRBM_GUI_ACTIVE" in data and f in data["GRBM_GUI_ACTIVE"][0]:
            busy = data["SQ_VALU_MFMA_BUSY_CYCLES"][0][f][1]
            act = data["GRBM_GUI_ACTIVE"][0][f][1]
            e["mfma_busy_cycles"] = int(busy)
            e["gui_active_cycles"] = int(act)
            e["mfma_busy_pct"] = round(100.0 * busy / (act * 256 * 4), 3) if act else None
        e["source"] = "profiles/" + os.path.basename(out).replace(".json", ".txt")
        res["kernels"][f] = e
    json.dump(res, open(out, "w"), indent=1)
    for f, e in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) * kv[1]["launches_per_step"]):
        print(f"{f:42s} n={e['launches_per_step']:4d}  fetch {e.get('fetch_kib_raw', 0) / 1024:9.1f} MiB raw  write {e.get('write_kib_raw', 0) / 1024:8.1f} MiB"
              f"  hbm/launch {e.get('hbm_bytes_per_launch', 0) / 1e6:8.3f} MB  mfma_busy {e.get('mfma_busy_pct')}")


if __name__ == "__main__":
    main()

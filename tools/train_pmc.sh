# MFMA-busy of the training step's kernels: rocprofv3 PMC pass over tools/train_step.py (counters in their own run)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/trainpmc; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o p --output-format rocpd -- python tools/train_step.py --iters 3 > $O/pmc.log 2>&1
python - $(find $O/pmc -name "*.db" | head -1) > $O/${TAG}_train_pmc.txt <<'PY'
import sqlite3, sys, collections, json
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
ipcols = [r[1] for r in c.execute(f"pragma table_info({ip})")]
namecol = "name" if "name" in ipcols else ipcols[8]
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scols else "kernel_name"
res = {}
for counter, fn in (("SQ_VALU_MFMA_BUSY_CYCLES", "sum"), ("GRBM_GUI_ACTIVE", "avg")):
    ids = [r[0] for r in c.execute(f"select id from {ip} where {namecol}=?", (counter,))]
    q = f"""select s.{name_col}, {fn}(p.value) from {kd} d join {ks} s on d.kernel_id = s.id join {pe} p on p.event_id = d.event_id
            where p.pmc_id in ({','.join(str(i) for i in ids)}) group by d.id"""
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, v in c.execute(q):
        k = "train_gemm" if ("train_gemm" in n or "big_gemm" in n) else ("gn/ln/act kernels" if any(x in n for x in ("gn_", "ln_", "act_", "softmax", "colsum")) else "other")
        agg[k][0] += 1; agg[k][1] += v
    res[counter] = agg
out = {}
for k in res["GRBM_GUI_ACTIVE"]:
    busy = res["SQ_VALU_MFMA_BUSY_CYCLES"][k][1]; act = res["GRBM_GUI_ACTIVE"][k][1]
    out[k] = {"dispatches": res["GRBM_GUI_ACTIVE"][k][0], "mfma_busy_cycles": int(busy), "gui_active_cycles": int(act),
              "mfma_busy_pct": round(100.0 * busy / (act * 256 * 4), 3) if act else None}
print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/train_step.py --iters 3 (tools/train_pmc.sh); "
                            "MFMA busy = 100 * busy / (GUI_ACTIVE per XCD * 256 CUs * 4)", "kernels": out}, indent=1))
PY
cat $O/${TAG}_train_pmc.txt; tail -2 $O/pmc.log
rm -rf $O/pmc

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/trainprof; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format rocpd -- python tools/train_step.py --iters 4 > $O/trace.log 2>&1
python - <<'PY' $(find $O/trace -name "*.db" | head -1) > $O/train_kernels_full.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scols else "kernel_name"
rows = list(c.execute(f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.{name_col} order by 3 desc"))
tot = sum(r[2] for r in rows)
for r in rows[:40]:
    print(f"{r[1]:6d} {r[2]/1e6:9.3f}ms {r[3]/1e3:8.2f}us {100*r[2]/tot:5.1f}%  {r[0][:600]}")
PY
tail -3 $O/trace.log
rm -rf $O/trace

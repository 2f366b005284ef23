#!/usr/bin/env python3
"""Sum one PMC counter (FETCH_SIZE / WRITE_SIZE, in KiB) per kernel family over one replayed denoiser step.
usage: rocpd_pmc.py results.db COUNTER [step_index]"""
import collections
import sqlite3
import sys

db, counter = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
ipcols = [r[1] for r in c.execute(f"pragma table_info({ip})")]
namecol = "name" if "name" in ipcols else ipcols[8]
pmc_ids = [r[0] for r in c.execute(f"select id from {ip} where {namecol}=?", (counter,))]
scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scols else "kernel_name"
q = f"""select s.{name_col}, d.start, sum(p.value) from {kd} d join {ks} s on d.kernel_id = s.id
        join {pe} p on p.event_id = d.event_id where p.pmc_id in ({','.join(str(i) for i in pmc_ids)})
        group by d.id order by d.start"""
rows = list(c.execute(q))
idx = [i for i, r in enumerate(rows) if 'pack_input' in r[0]]
a, b = idx[which], idx[which + 1] if which + 1 < len(idx) else len(rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for n, _, v in rows[a:b]:
    key = ('conv_gemm' if ('conv_gemm' in n or 'stream_gemm' in n or 'tile_gemm' in n or 'TileArgs' in n) else 'norm_apply' if 'norm_apply' in n else 'attention' if 'attention' in n else n.split('(')[0][-36:])
    agg[key][0] += 1
    agg[key][1] += v
tot = sum(v[1] for v in agg.values())
print(f"{counter} over one step ({b - a} dispatches): {tot / 1024:.1f} MiB raw")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {k:40s} n={v[0]:4d}  {v[1] / 1024:9.1f} MiB raw  ({v[1] / 1024 / v[0]:.3f} MiB / launch)")

"""Summarise a rocprofv3 --kernel-trace results.db: per (kernel, grid) calls / average / total.  python scripts/trace_summary.py results.db [filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "mmf::"
rows = list(c.execute("select name, grid_x / workgroup_x, grid_y, count(*), avg(end - start) / 1e3, sum(end - start) / 1e6 from kernels "
                      "where name like ? group by name, grid_x, grid_y order by 6 desc", (f"%{flt}%",)))
tot = sum(r[5] for r in rows)
for r in rows:
    print(f"{r[0].split('(')[0][:44]:44s} grid {r[1]:6d} x {r[2]:4d} calls {r[3]:6d} avg {r[4]:9.2f} us total {r[5]:9.3f} ms ({100 * r[5] / tot:5.1f} %)")

"""Per-kernel summary of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db):
python tools/rocpd_stats.py file.db [top]  ->  name, calls, avg us, total ms, grid"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name = "kernel_name" if "kernel_name" in cols else ("display_name" if "display_name" in cols else cols[-1])
q = f"select s.{name}, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.grid_size_x), max(d.grid_size_y) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name}, d.grid_size_x, d.grid_size_y order by 4 desc limit {top}"
for n, c, a, t, gx, gy in db.execute(q):
    print(f"{n[:100]:100s} calls {c:5d}  avg {a / 1e3:9.1f} us  total {t / 1e6:9.3f} ms  grid {gx}x{gy}")

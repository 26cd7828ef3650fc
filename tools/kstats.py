"""Per-kernel table from a rocprofv3 rocpd database:  python tools/kstats.py gpurun_out/prof_x/name_results.db [rows]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print(f"{'kernel':<100} {'calls':>5} {'total_ms':>10} {'avg_ms':>9} {'pct':>6}")
for n, c, t, a, p in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{n[:100]:<100} {c:>5} {t / 1e3:>10.3f} {a / 1e3:>9.4f} {p:>6.2f}")

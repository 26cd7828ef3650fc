"""Turn rocprofv3's rocpd sqlite outputs (gpurun_out/prof/*/..._results.db) into the small text summaries kept
under profiles/:  per-kernel stats (calls, total, average duration) and per-kernel PMC counter averages.

    python tools/summarize_rocprof.py gpurun_out/prof profiles/r01
"""
import glob
import json
import os
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    # rocpd's top_kernels view reports durations in microseconds
    lines = [f"{'kernel':<110} {'calls':>6} {'total_ms':>14} {'avg_ms':>12} {'pct':>7}"]
    for name, calls, total, avg, pct in rows:
        lines.append(f"{name[:110]:<110} {calls:>6} {total / 1e3:>14.3f} {avg / 1e3:>12.4f} {pct:>7.3f}")
    occ = list(cur.execute("select distinct name, workgroup_x, grid_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size "
                           "from kernels where name like '%mlp_fwd%' limit 4"))
    lines.append("")
    lines.append("dispatch geometry of the dominant kernel (name, workgroup, grid, lds, vgpr, agpr, sgpr, scratch):")
    for r in occ:
        lines.append("  " + str(r))
    return "\n".join(lines), rows


def pmc_summary(db):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by kernel_name, counter_name")
    out = {}
    for k, c, n, s, a, d in cur.execute(q):
        out.setdefault(k, {})[c] = {"dispatches": n, "sum": s, "avg_per_dispatch": a, "avg_duration_ns": d}
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats_db = glob.glob(os.path.join(src, "stats", "*_results.db"))
    if stats_db:
        text, _ = kernel_stats(stats_db[0])
        open(dst + "_kernel_stats.txt", "w").write(text + "\n")
        print(text)
    pmc = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for db in glob.glob(os.path.join(d, "*_results.db")):
            for k, v in pmc_summary(db).items():
                pmc.setdefault(k, {}).update(v)
    if pmc:
        json.dump(pmc, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
        for k, v in pmc.items():
            if "aon::" in k:
                print(k[:80])
                for c, x in v.items():
                    print(f"   {c:<28} n={x['dispatches']:<4} avg/dispatch={x['avg_per_dispatch']:.4g}  avg_dur_us={x['avg_duration_ns'] / 1e3:.1f}")


if __name__ == "__main__":
    main()

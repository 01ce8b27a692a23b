"""Turn a rocprofv3 results.db (rocpd sqlite) into a short per-kernel stats table.

    python scripts/prof_summary.py gpurun_out/<dir> > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def main(path, note=""):
    dbs = glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True) if os.path.isdir(path) else [path]
    for db in dbs:
        c = sqlite3.connect(db)
        print("# rocprofv3 --kernel-trace --stats summary of %s" % os.path.relpath(db))
        if note:
            print("# " + note)
        print("%-72s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        rows = c.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
            "group by name order by sum(duration) desc"
        ).fetchall()
        tot = sum(r[2] for r in rows)
        for n, k, s, a, mn, mx in rows[:12]:
            short = n if len(n) <= 72 else n[:69] + "..."
            print("%-72s %8d %14.1f %12.2f %12.2f %12.2f %6.2f%%" % (short, k, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100 * s / tot))
        r = c.execute(
            "select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, lds_size "
            "from kernels where name like '%k_step%' limit 1"
        ).fetchone()
        if r:
            print("# k_step launch: grid=%d wg=%d vgpr=%d agpr=%d sgpr=%d scratch=%d lds=%d" % r[1:])


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))

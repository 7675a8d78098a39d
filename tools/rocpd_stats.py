#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns, VGPRs, LDS) from a rocprofv3 rocpd
SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db).
Usage: tools/rocpd_stats.py file.db [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    dur = "(end - start)" if "duration" not in cols else "duration"
    extra = [x for x in ("arch_vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_block_size", "scratch_size",
                         "grid_size", "workgroup_size") if x in cols]
    q = (f"select {name_col}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur})"
         + "".join(f", max({e})" for e in extra) + f" from kernels group by {name_col} order by sum({dur}) desc")
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print("# name | calls | total_ns | avg_ns | min_ns | max_ns | pct | " + " | ".join(extra))
    for r in rows:
        print(f"{r[0][:110]} | {r[1]} | {int(r[2])} | {r[3]:.0f} | {int(r[4])} | {int(r[5])} | {100.0 * r[2] / tot:.1f}% | "
              + " | ".join(str(x) for x in r[6:]))


if __name__ == "__main__":
    main(sys.argv[1])

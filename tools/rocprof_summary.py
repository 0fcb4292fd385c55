#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory (rocpd SQLite .db) into small CSV files and delete the .db
(gpurun copies back at most 64 MiB).   python tools/rocprof_summary.py <dir> <out_prefix> [--keep]"""
import glob
import os
import sqlite3
import sys


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    keep = "--keep" in sys.argv
    for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(db)
        cur = con.cursor()
        tot = cur.execute("select sum(end-start)/1e6 from kernels").fetchone()[0] or 0.0
        rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                           "from kernels group by name order by 3 desc").fetchall()
        with open(prefix + "_kernel_stats.csv", "w") as f:
            f.write(f"# total kernel time {tot:.3f} ms\nname,calls,total_ms,percent,avg_us,min_us,max_us\n")
            for r in rows:
                f.write(f"\"{r[0]}\",{r[1]},{r[2]:.4f},{100 * r[2] / max(tot, 1e-9):.2f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f}\n")
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        pmc = [t for t in tabs if t.lower() in ("counters_collection", "pmc_events", "rocpd_pmc_event")]
        try:
            if "counters_collection" in tabs:
                cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
                with open(prefix + "_counters.csv", "w") as f:
                    f.write("# columns: " + ",".join(cols) + "\n")
                    kn = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
                    cn = "counter_name" if "counter_name" in cols else None
                    vn = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
                    if cn and vn:
                        f.write("kernel,counter,dispatches,sum,avg\n")
                        for r in cur.execute(f"select {kn}, {cn}, count(*), sum({vn}), avg({vn}) from counters_collection group by {kn}, {cn} order by 4 desc"):
                            f.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]},{r[4]}\n")
            else:
                with open(prefix + "_counters.csv", "w") as f:
                    f.write("# no counters_collection view; tables: " + " ".join(tabs) + "\n")
        except Exception as e:       # keep going: the kernel stats are the important part
            with open(prefix + "_counters.csv", "a") as f:
                f.write(f"# error {e!r}; tables: {' '.join(tabs)}\n")
        con.close()
        if not keep:
            os.remove(db)


if __name__ == "__main__":
    main()

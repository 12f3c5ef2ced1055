#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite database (kernel trace, optionally PMC counters) into small text summaries that can be
committed under profiles/.

  python tools/rocprof_summary.py <results.db> --stats out.csv          # per-kernel calls / total / average (us)
  python tools/rocprof_summary.py <results.db> --pmc out.csv            # per-kernel mean counter values
  python tools/rocprof_summary.py <results.db> --pmc-rows rows.csv      # one row per (dispatch, counter), in dispatch order
"""
import argparse
import csv
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--stats")
    ap.add_argument("--pmc")
    ap.add_argument("--pmc-rows", dest="pmc_rows")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cur = db.cursor()
    if a.stats:
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        unit = 1.0
        with open(a.stats, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
            for name, calls, tot, avg, pct in rows[: a.top]:
                w.writerow([name[:160], calls, round(tot * unit, 1), round(avg * unit, 3), round(pct, 3)])
        print(f"wrote {a.stats} ({min(len(rows), a.top)} kernels)")
    if a.pmc:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        rows = list(cur.execute("select * from counters_collection"))
        ix = {c: i for i, c in enumerate(cols)}
        name_c = "kernel_name" if "kernel_name" in ix else "name"
        agg = {}
        for r in rows:
            key = (r[ix[name_c]], r[ix["counter_name"]])
            agg.setdefault(key, []).append(float(r[ix["value"]]))
        with open(a.pmc, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
            for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k[:160], c, len(v), sum(v) / len(v), min(v), max(v)])
        print(f"wrote {a.pmc} ({len(agg)} kernel/counter pairs); columns={cols}")
    if a.pmc_rows:
        pmc_rows(db, a.pmc_rows)


def pmc_rows(db, path):
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "kernel_name" if "kernel_name" in ix else "name"
    order_c = next((c for c in ("dispatch_id", "start", "start_timestamp", "id") if c in ix), None)
    q = "select * from counters_collection" + (f" order by {order_c}" if order_c else "")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["order", "kernel", "counter", "value"])
        for r in cur.execute(q):
            w.writerow([r[ix[order_c]] if order_c else "", r[ix[name_c]][:160], r[ix["counter_name"]], float(r[ix["value"]])])
    print(f"wrote {path}; ordered by {order_c}; columns={cols}")


if __name__ == "__main__":
    sys.exit(main())

"""Turn a rocprofv3 --kernel-trace output (rocpd .db or kernel_trace.csv) into the small per-kernel
summary committed under profiles/ (name, calls, total us, mean us, % of GPU kernel time)."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in rows]   # durations in us


def from_csv(path):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += d
    tot = sum(v[1] for v in acc.values()) or 1.0
    return sorted(((k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot) for k, v in acc.items()), key=lambda x: -x[2])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
    csvs = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = from_db(dbs[0]) if dbs else from_csv(csvs[0])
    with open(dst, "w") as f:
        f.write("| kernel | calls | total us | mean us | % |\n|---|---:|---:|---:|---:|\n")
        for n, c, t, a, p in rows[:40]:
            n = n.replace("f3dgs::(anonymous namespace)::", "").replace("void ", "")
            f.write(f"| `{n[:90]}` | {c} | {t:.1f} | {a:.2f} | {p:.2f} |\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()

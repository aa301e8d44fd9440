"""Aggregate rocprofv3 --pmc counter_collection.csv files into the per-kernel JSON committed under profiles/.

usage: pmc_summary.py <out.json> <dir-or-csv> [<dir-or-csv> ...]
       pmc_summary.py --traffic <pmc_traffic.json> <raw.json>

The first form averages every counter over the launches of each kernel (summing the per-dispatch rows rocprofv3
emits per counter).  The second form derives HBM bytes per launch for the four heavy kernels as
(2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE count KiB, and gfx950 reports half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM/rocprofv3 section), hence the factor 2 on the read side.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

HEAVY = {"render_bwd": "render_backward", "render_fwd": "render_forward", "preprocess": "preprocess_kernel",
         "preprocess_bwd": "preprocess_backward_kernel"}


def short(name):
    return name.replace("f3dgs::(anonymous namespace)::", "").replace("void ", "")[:60]


def aggregate(paths):
    files = []
    for p in paths:
        files += [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"),
                                                           recursive=True)
    per_dispatch = defaultdict(float)          # (file, dispatch, kernel, counter) -> value
    for fi, f in enumerate(files):
        for r in csv.DictReader(open(f)):
            per_dispatch[(fi, r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for (_, _, k, c), v in per_dispatch.items():
        a = acc[k][c]
        a[0] += 1
        a[1] += v
    return {k: {c: a[1] / a[0] for c, a in sorted(cs.items())} | {"launches": max(a[0] for a in cs.values())}
            for k, cs in acc.items()}


def main():
    if sys.argv[1] == "--traffic":
        raw = json.load(open(sys.argv[3]))
        out = {}
        for key, pat in HEAVY.items():
            for k, cs in raw.items():
                if k.startswith(pat) and "FETCH_SIZE" in cs:
                    out[key] = int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)
        out["_note"] = ("HBM bytes per launch at config c3 from rocprofv3 PMC, separate passes: "
                        "(2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 "
                        "reports half the bytes of wide coalesced reads; other widths and WRITE_SIZE uncalibrated). "
                        f"Raw counters: profiles/{os.path.basename(sys.argv[3])}; tool: tools/pmc_summary.py")
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    else:
        json.dump(aggregate(sys.argv[2:]), open(sys.argv[1], "w"), indent=1)
    print("wrote", sys.argv[2] if sys.argv[1] == "--traffic" else sys.argv[1])


if __name__ == "__main__":
    main()

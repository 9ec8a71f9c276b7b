#!/usr/bin/env python3
"""per counter: mean over the fast (even) and slow (odd) launches of the LAST 8 scs_spmv_tlc dispatches of every pass under <dir>/g*/"""
import collections, csv, glob, sys
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "scs_spmv_tlc"
for d in sorted(glob.glob(sys.argv[1] + "/g*")):
    rows = collections.defaultdict(list)
    for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    for c, v in sorted(rows.items()):
        v = sorted(v)[-8:]
        fast = [x[1] for x in v[0::2]]; slow = [x[1] for x in v[1::2]]
        tf = [x[2] for x in v[0::2]]; ts = [x[2] for x in v[1::2]]
        mf, ms = sum(fast) / len(fast), sum(slow) / len(slow)
        print(f"{c:44s} fast {mf:14.6g}  slow {ms:14.6g}  slow/fast {ms / mf if mf else float('nan'):7.4f}   (kernel ms {sum(tf) / len(tf):.4f} / {sum(ts) / len(ts):.4f})")

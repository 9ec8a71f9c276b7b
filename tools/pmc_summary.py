#!/usr/bin/env python3
"""Summarise the counter_collection.csv files written by tools/pmc.sh: mean per dispatch, per kernel."""
import collections, csv, glob, re, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "stream_" in k or "at::native" in k or "rocclr" in k:
            continue
        m = re.search(r"(\w+<[^>]*>)", k)
        k = m.group(1) if m else k[:60]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in out:
    d = sorted(dur[k])
    print(f"== {k}   dispatches={len(d)//max(1,len(out[k]))} median_ms={d[len(d)//2]:.4f}")
    for c, v in sorted(out[k].items()):
        v = sorted(v)
        print(f"   {c:45s} {v[len(v)//2]:.6g}")

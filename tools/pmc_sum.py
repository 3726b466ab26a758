#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per (kernel, counter): python tools/pmc_sum.py DIR [out.csv]"""
import collections, csv, glob, os, sys
tot, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        # 'void svtme::me_search_kernel<svtme::OnePicture>(svtme::OnePicture, int, int, int)' -> 'svtme::me_search_kernel'
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        k = (name[:60], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
rows = ["kernel,counter,dispatches,sum,per_dispatch"]
for k in sorted(tot):
    rows.append("%s,%s,%d,%.0f,%.1f" % (k[0], k[1], len(n[k]), tot[k], tot[k] / max(1, len(n[k]))))
text = "\n".join(rows)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")

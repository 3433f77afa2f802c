#!/usr/bin/env python
"""Summarise rocprofv3 PMC csv files: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections
def short(n):
    n = n.replace("void ", "").replace("(W2xcConvDesc, int, int)", "")
    return n[:46]
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("#", path)
    for k in acc:
        if not k.startswith("conv3x3"): continue
        d = sum(dur[k]) / len(dur[k])
        print("%-46s n=%d avg_ns=%d " % (k, len(dur[k]) // max(len(acc[k]), 1), d) + " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))

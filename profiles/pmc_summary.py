#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv rows per kernel and counter.
usage: pmc_summary.py <dir with passN/*/..._counter_collection.csv> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "k_tick"
acc = collections.defaultdict(float); calls = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/pass*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            calls[row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(acc):
    n = max(1, len(calls[k]))
    print("%-24s %18.0f  per launch %18.0f  (%d launches)" % (k, acc[k], acc[k] / n, n))

"""Summarise rocprofv3 counter_collection.csv files: per kernel, mean of each counter."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:48]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if not any(s in k for s in ("shrink", "oklab", "qoi", "expand", "tree", "pack", "pixlzr")): continue
            print(d.split("/")[-1], k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))

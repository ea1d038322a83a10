"""HBM bytes per step from rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE collected in
separate runs, MI355X_MICROARCH.md HBM section): per kernel the mean over launches, summed over the
kernels of one step.  FETCH_SIZE is doubled (gfx950 counts 128-byte requests in 64-byte units);
WRITE_SIZE is exact (calibrated on synth_kernel: 1036800 KB for 8 frames).  Units: KB = 1024 B.

  python3 tools/pmc_traffic.py <dir with pmc_{fetch,write}_{dir,by}/...> > profiles/pmc_traffic.json
"""
import csv, glob, json, sys, collections
root = sys.argv[1]
STEP_KERNELS = ("shrink32_kernel", "shrink_kernel", "oklab32_kernel", "finish_kernel")
out = {"_how": __doc__.strip(), "raw_KB": {}}
for mode, tag in (("shrink_directionally", "dir"), ("shrink_by", "by")):
    raw = collections.defaultdict(dict)
    for counter, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{root}/pmc_{name}_{tag}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] != counter:
                    continue
                for k in STEP_KERNELS:
                    if k in row["Kernel_Name"]:
                        acc[k].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            raw[k][counter] = round(sum(v) / len(v), 2)
            raw[k]["launches"] = len(v)
    out["raw_KB"][mode] = raw
    total = 0.0
    for k, c in raw.items():
        total += 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0 + c.get("WRITE_SIZE", 0.0) * 1024.0
    out[mode] = int(total)
print(json.dumps(out, indent=1))

"""HBM bytes per step of every flow, from rocprofv3 PMC passes of tools/pmc_run.py (one flow per run; FETCH_SIZE and
WRITE_SIZE in separate runs, as MI355X_MICROARCH.md's HBM section prescribes).  Every kernel of the library that ran in
the flow is counted (names starting with pxz::), per kernel: the sum over its launches / the steps of the run; the
kernels of a flow's set-up (tools/pmc_run.py names them, e.g. the shrink in front of the writer) are listed apart.
FETCH_SIZE is doubled (gfx950 counts the 128-byte requests of wide streaming reads in 64-byte units); WRITE_SIZE is exact
(calibrated on synth_kernel: 1036800 KB for 8 frames).  Counter units: KB = 1024 B.

  python3 tools/pmc_traffic.py <dir made by tools/pmc_traffic.sh> > profiles/rNN_pmc_traffic.json
"""
import csv, glob, json, os, re, sys, collections

root = sys.argv[1]
out = {"_how": __doc__.strip(), "flows": {}}


def short(name):
    m = re.search(r"pxz::(\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


for info_path in sorted(glob.glob(f"{root}/*.json")):
    info = json.load(open(info_path))
    tag = os.path.basename(info_path)[:-5]
    steps = info["steps"]
    step_pref = [p for p in info.get("step_kernels", "").split(",") if p]
    setup_pref = [p for p in info.get("setup_kernels", "").split(",") if p]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(int)
    for counter, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for f in glob.glob(f"{root}/{tag}_{name}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if row["Counter_Name"] != counter or k is None or k.startswith("synth_kernel"):
                    continue
                per[k][counter] += float(row["Counter_Value"])
                if counter == "FETCH_SIZE":
                    calls[k] += 1
    kernels, setup = {}, {}
    total = 0.0
    for k, c in sorted(per.items()):
        is_setup = any(k.startswith(p) for p in setup_pref) or (step_pref and not any(k.startswith(p) for p in step_pref))
        b = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        row = {"launches": calls[k], "FETCH_SIZE_KB_sum": round(c.get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KB_sum": round(c.get("WRITE_SIZE", 0.0), 1)}
        if is_setup:
            setup[k] = row
            continue
        row["hbm_bytes_per_step"] = int(b / steps)
        kernels[k] = row
        total += b / steps
    algo = info.get("algo_bytes")
    out["flows"][tag] = {"run": info, "kernels": kernels, "setup_kernels_not_counted": setup, "hbm_bytes_per_step": int(total),
                         "algorithmic_bytes_per_step": algo, "traffic_over_algorithmic": round(total / algo, 3) if algo else None}
# the two keys bench.py reads (8 x 8K, 32x32 tiles)
for key, tag in (("shrink_directionally", "dir32"), ("shrink_by", "by32"), ("shrink_directionally+encode_to_vec", "enc32")):
    if tag in out["flows"]:
        out[key] = out["flows"][tag]["hbm_bytes_per_step"]
if "dir32" in out["flows"] and "enc32" in out["flows"]:
    out["shrink_directionally+encode_to_vec"] = out["flows"]["dir32"]["hbm_bytes_per_step"] + out["flows"]["enc32"]["hbm_bytes_per_step"]
print(json.dumps(out, indent=1))

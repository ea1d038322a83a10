"""SQ counter passes (tools/sq_counters.sh -> gpurun_out/<tag>_sq.txt, one line per kernel and counter group) as one JSON
with the ratios bench.py quotes:  python3 tools/sq_summary.py gpurun_out/a_sq.txt gpurun_out/b_sq.txt ... > profiles/sq_summary.json
All SQ_* cycle counters are sums over the waves (or SIMDs) of a launch, in the counters' own units:
  valu_active_share_of_wave_cycles   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: the share of a wave's resident time in which one of
                                     ITS vector instructions executes; times the waves per SIMD (valu_busy_at_4_waves_per_simd
                                     for the 16-wave blocks of the detector and of shrink32_kernel) it is the share of SIMD
                                     cycles with a vector instruction executing
  wait_share                         SQ_WAIT_ANY / SQ_WAVE_CYCLES: parked at s_waitcnt / s_barrier
  issue_stall_share                  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: ready but not issued
  valu_instructions_per_wave         SQ_INSTS_VALU / SQ_WAVES
"""
import ast, json, re, sys, collections
kern = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r"^(g\d+) (.*?) (\{.*\}) n=(\d+)$", line.strip())
        if not m:
            continue
        name = re.sub(r"^void ", "", m.group(2)).replace("pxz::", "")
        name = re.sub(r"\(.*$", "", name)
        kern[name].update(ast.literal_eval(m.group(3)))
out = {"_how": __doc__.strip(), "kernels": {}}
for name, c in kern.items():
    rec = {k: v for k, v in c.items()}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        if "SQ_ACTIVE_INST_VALU" in c:
            rec["valu_active_share_of_wave_cycles"] = round(c["SQ_ACTIVE_INST_VALU"] / wc, 4)
        if "SQ_WAIT_ANY" in c:
            rec["wait_share"] = round(c["SQ_WAIT_ANY"] / wc, 4)
        if "SQ_WAIT_INST_ANY" in c:
            rec["issue_stall_share"] = round(c["SQ_WAIT_INST_ANY"] / wc, 4)
        if "SQ_ACTIVE_INST_VALU" in c:
            rec["valu_busy_at_4_waves_per_simd"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / wc, 4)
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        rec["valu_instructions_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
    out["kernels"][name] = rec
print(json.dumps(out, indent=1))

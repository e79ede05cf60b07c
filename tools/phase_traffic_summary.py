#!/usr/bin/env python3
"""Bytes per call of each phase from the counters of tools/phase_traffic.sh (FETCH_SIZE / WRITE_SIZE in KB per dispatch, summed over the
XCDs; fetch corrected by the factor tools/pmc_calib measured on gfx950 for 8-byte-per-lane buffer loads, default and nt cache policy alike:
the counter reports HALF the bytes).  What the counters see is what crosses L2 towards the fabric: Infinity-Cache hits as well as HBM."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
FETCH_FACTOR = 2.0
disp = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(src, "pmc_" + c, "*", "*counter_collection.csv"))
    if not f:
        sys.exit("no counters for " + c)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        e = per.setdefault(int(r["Dispatch_Id"]), {"kernel": k, "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "v": 0.0})
        if r["Counter_Name"] == c:
            e["v"] += float(r["Counter_Value"])
    for n, (d, e) in enumerate(per.items()):
        x = disp.setdefault(n, {"kernel": e["kernel"]})
        x[c + "_KB"] = e["v"]; x["ms_" + c] = e["ms"]
rows = list(disp.values())
ncell = 3072
for x in rows:
    x["read_bytes_per_cell"] = x["FETCH_SIZE_KB"] * 1024 * FETCH_FACTOR / ncell
    x["write_bytes_per_cell"] = x["WRITE_SIZE_KB"] * 1024 / ncell
names = [x["kernel"] for x in rows]
out = {"cells": ncell, "dispatches": rows}
newt = [x for x in rows if x["kernel"] == "k_newton"]
if len(newt) == 3:
    base, lu9, so9 = newt
    out["per_call"] = {
        "lu": {"read": (lu9["read_bytes_per_cell"] - base["read_bytes_per_cell"]) / 8, "write": (lu9["write_bytes_per_cell"] - base["write_bytes_per_cell"]) / 8,
               "ms_per_call_3072_cells": (lu9["ms_FETCH_SIZE"] - base["ms_FETCH_SIZE"]) / 8},
        "solve": {"read": (so9["read_bytes_per_cell"] - base["read_bytes_per_cell"]) / 8, "write": (so9["write_bytes_per_cell"] - base["write_bytes_per_cell"]) / 8,
                  "ms_per_call_3072_cells": (so9["ms_FETCH_SIZE"] - base["ms_FETCH_SIZE"]) / 8}}
    k = {x["kernel"]: x for x in rows if x["kernel"] != "k_newton"}
    if "k_rates" in k:
        for nm in ("k_rhs", "k_jac"):
            if nm in k:
                out["per_call"][nm[2:]] = {"read": k[nm]["read_bytes_per_cell"] - k["k_rates"]["read_bytes_per_cell"],
                                           "write": k[nm]["write_bytes_per_cell"] - k["k_rates"]["write_bytes_per_cell"],
                                           "note": "k_%s minus k_rates (both compute and write the rate vector first)" % nm[2:]}
        out["per_call"]["rates"] = {"read": k["k_rates"]["read_bytes_per_cell"], "write": k["k_rates"]["write_bytes_per_cell"]}
json.dump(out, open(os.path.join(src, "phase_traffic.json"), "w"), indent=1)
print(json.dumps(out.get("per_call"), indent=1))

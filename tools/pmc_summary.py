#!/usr/bin/env python3
"""Condenses the rocprofv3 CSV output of tools/profile_pmc.sh into one JSON: per kernel the average
duration, every counter averaged per launch, and derived figures (VALU lane utilisation, LDS
bank-conflict share, HBM traffic per launch with the gfx950 FETCH_SIZE correction)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        per_dispatch = defaultdict(dict)
        for r in rows:
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for (_, k), cs in per_dispatch.items():
            for c, v in cs.items():
                out[k][c].append(v)
    return out


def main():
    root = sys.argv[1]
    res = defaultdict(dict)
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            res[r["Name"]]["avg_ns"] = float(r["AverageNs"])
            res[r["Name"]]["calls"] = int(r["Calls"])
    for p in ("sq1", "sq2", "fetch", "write"):
        for k, cs in read_counters(os.path.join(root, p)).items():
            for c, vals in cs.items():
                res[k][c] = sum(vals) / len(vals)
    for k, r in res.items():
        if "SQ_THREAD_CYCLES_VALU" in r and r.get("SQ_ACTIVE_INST_VALU"):
            # thread-cycles / (instruction-cycles * 64 lanes): share of lanes doing work per VALU instruction
            r["valu_lane_utilisation"] = r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_ACTIVE_INST_VALU"] * 64.0) if r["SQ_ACTIVE_INST_VALU"] else None
        if r.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_bank_conflict_share"] = r.get("SQ_LDS_BANK_CONFLICT", 0.0) / r["SQ_LDS_IDX_ACTIVE"]
        if r.get("SQ_WAVE_CYCLES"):
            r["wait_any_share"] = r.get("SQ_WAIT_ANY", 0.0) / r["SQ_WAVE_CYCLES"]
            r["active_inst_share"] = r.get("SQ_ACTIVE_INST_ANY", 0.0) / r["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in r or "WRITE_SIZE" in r:
            # rocprofv3 reports KB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
            r["hbm_read_bytes_raw"] = r.get("FETCH_SIZE", 0.0) * 1024.0
            r["hbm_read_bytes_corrected"] = 2.0 * r.get("FETCH_SIZE", 0.0) * 1024.0
            r["hbm_write_bytes"] = r.get("WRITE_SIZE", 0.0) * 1024.0
            r["hbm_traffic_bytes"] = r["hbm_read_bytes_corrected"] + r["hbm_write_bytes"]
        if r.get("TCC_HIT_sum") is not None and (r.get("TCC_HIT_sum", 0) + r.get("TCC_MISS_sum", 0)) > 0:
            r["l2_hit_rate"] = r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"])
    print(json.dumps({k: v for k, v in sorted(res.items()) if "zj_" in k}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

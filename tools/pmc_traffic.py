#!/usr/bin/env python3
"""Adds the per-launch HBM traffic of one profiled workload (a summary written by tools/pmc_summary.py) to the table
that bench.py attaches to its roofline object:
  {"kernels_sha256": hash of the kernel sources the counters were taken on, "commit": set by tools/stamp_traffic.py,
   "workloads": {key: {short kernel name: bytes per launch}}}
A table whose hash differs from the tree's is started afresh: counters of other kernels say nothing about these.
Usage: pmc_traffic.py <summary.json> <workload key> <traffic.json to update>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUP = {"k1c_dcdiff": "k1d_dcscan", "k1d_tilesum": "k1d_dcscan", "k1d_tilescan": "k1d_dcscan", "k1d_dcscan_small": "k1d_dcscan",
         "k1d_dcscan_seg": "k1d_dcscan", "k1s_scan_few": "k1s_scan", "k1a_sync_packed": "k1a_sync"}  # timed together by the library
KERNEL_SOURCES = ["zj_kernels.hip", "zj_core.h", "zj_dev.h"]


def kernels_sha256():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "jpeg-decoder_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def main():
    summary = json.load(open(sys.argv[1]))
    key, path = sys.argv[2], sys.argv[3]
    out = {}
    for name, r in summary.items():
        m = re.search(r"zj_(k[0-9a-z]+_[a-z_]+)", name)
        if not m or "hbm_traffic_bytes" not in r:
            continue
        short = GROUP.get(m.group(1), m.group(1))
        out[short] = out.get(short, 0.0) + r["hbm_traffic_bytes"]
    sha = kernels_sha256()
    table = {}
    if os.path.exists(path):
        try:
            table = json.load(open(path))
        except Exception:
            table = {}
    if table.get("kernels_sha256") != sha:
        table = {"kernels_sha256": sha, "commit": None,
                 "collected": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each (tools/profile_pmc.sh), bytes per launch "
                              "with the gfx950 read correction of tools/pmc_summary.py",
                 "workloads": {}}
    table["workloads"][key] = {k: round(v) for k, v in sorted(out.items())}
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(f"traffic of {key}: {len(out)} kernels -> {path}")


if __name__ == "__main__":
    main()

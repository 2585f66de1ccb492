#!/usr/bin/env python3
"""Turns a summary written by tools/pmc_summary.py into the per-launch HBM traffic table that bench.py
attaches to its roofline object: {"images": N, "kernels": {short kernel name: bytes per launch}}.
Usage: pmc_traffic.py <summary.json> <images per launch> > profiles/traffic.json"""
import json
import re
import sys

GROUP = {"k1c_dcdiff": "k1d_dcscan"}  # timed together by the library


def main():
    summary = json.load(open(sys.argv[1]))
    images = int(sys.argv[2])
    out = {}
    for name, r in summary.items():
        m = re.search(r"zj_(k[0-9a-z]+_[a-z_]+)", name)
        if not m or "hbm_traffic_bytes" not in r:
            continue
        short = GROUP.get(m.group(1), m.group(1))
        out[short] = out.get(short, 0.0) + r["hbm_traffic_bytes"]
    print(json.dumps({"images": images, "kernels": {k: round(v) for k, v in sorted(out.items())}}, indent=1))


if __name__ == "__main__":
    main()

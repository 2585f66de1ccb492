#!/usr/bin/env python3
"""Build container only (the GPU box has no git): records in profiles/traffic.json the commit whose kernel sources the
counters were taken on -- HEAD, when the hash in the table is that of the working tree's kernel sources.
Usage: python tools/stamp_traffic.py"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import ROOT, kernels_sha256

path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(path))
if t.get("kernels_sha256") != kernels_sha256():
    sys.exit("profiles/traffic.json was taken on other kernel sources than the tree holds: profile again")
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "jpeg-decoder_amd/csrc"], capture_output=True, text=True).stdout.strip()
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
t["commit"] = head + ("+uncommitted kernel changes" if dirty else "")
json.dump(t, open(path, "w"), indent=1, sort_keys=True)
print("stamped", t["commit"])

"""Scratch: the randomised GPU-against-oracle sweep of tests/sweep_parity.py with larger images (sizes up to 2800 and 5600)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sweep_parity as s
r = s.run(30, 777001, scale=4, per_batch=32)
print("RESULT", r)
r2 = s.run(6, 777002, scale=8, per_batch=16)
print("RESULT8", r2)

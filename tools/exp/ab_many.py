"""Scratch: kernel times of several builds of the library on one box, in turn, twice: ab_many.py <images> <a.so> <b.so> ... (uses the
child mode of ab_kernel_ms.py)."""
import subprocess, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs=sys.argv[2:]
for rnd in range(2):
    for lib in libs:
        out=subprocess.run([sys.executable, os.path.join(ROOT,"tools/exp/ab_kernel_ms.py"), "--child", sys.argv[1], lib], capture_output=True, text=True, timeout=300)
        line=out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:]
        try:
            d=json.loads(line); print(os.path.basename(lib), d.get("k2_idct_rgb"), d.get("total"), flush=True)
        except Exception: print(os.path.basename(lib), line, flush=True)

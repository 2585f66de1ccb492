"""Kernel times of the benchmark batch for several builds of the library on one GPU box (each in a child process).
Usage: multi_kernel_ms.py <images> <a.so> <b.so> ..."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for lib in sys.argv[2:]:
    out = subprocess.run([sys.executable, os.path.join(HERE, "ab_kernel_ms.py"), "--child", sys.argv[1], lib], capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)

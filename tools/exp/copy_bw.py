"""Practical HBM ceiling on this GPU: device-to-device copy (half read, half write) and fill (write only)."""
import torch, time
n = 8 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1); b.fill_(2); torch.cuda.synchronize()
for name, fn, bytes_ in [("copy", lambda: b.copy_(a), 2 * n), ("fill", lambda: b.fill_(3), n)]:
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {bytes_ / ms / 1e6:.0f} GB/s ({ms:.2f} ms)", flush=True)
# strided rows: 192-byte fragments at a 5760-byte pitch, like one wavefront's share of 16 pixel rows
x = torch.empty((1 << 20, 5760), dtype=torch.uint8, device="cuda")
src = torch.empty((1 << 20, 192), dtype=torch.uint8, device="cuda")
for w in (192, 768, 5760):
    s2 = torch.empty((1 << 20, w), dtype=torch.uint8, device="cuda")
    x[:, :w].copy_(s2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): x[:, :w].copy_(s2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"rows of {w} B at pitch 5760: {2 * (1 << 20) * w / ms / 1e6:.0f} GB/s", flush=True)

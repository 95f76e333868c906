"""Developer probe: do a pinned H2D and a pinned D2H copy on two streams overlap
on this box (PCIe is full duplex; the question is the DMA engines)?"""
import torch, time
up = torch.empty(531 * 1000 * 1000 // 4, dtype=torch.float32).pin_memory()
dn = torch.empty(143 * 1000 * 1000 // 4, dtype=torch.float32).pin_memory()
d_up = torch.empty_like(up, device="cuda"); d_dn = torch.empty_like(dn, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def h2d():
    with torch.cuda.stream(s1): d_up.copy_(up, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): dn.copy_(d_dn, non_blocking=True)
def both():
    h2d(); d2h()
a = t(h2d); b = t(d2h); c = t(both)
print("H2D 531 MB %.2f ms (%.1f GB/s)  D2H 143 MB %.2f ms (%.1f GB/s)  both %.2f ms" % (a, 0.531 / a * 1e3, b, 0.143 / b * 1e3, c))
# chunked H2D (64 frames of 8.3 MB), as the staging path issues it
chunks = up.view(64, -1); dch = d_up.view(64, -1)
def h2d_chunked():
    with torch.cuda.stream(s1):
        for i in range(64): dch[i].copy_(chunks[i], non_blocking=True)
print("H2D in 64 chunks %.2f ms" % t(h2d_chunked))

"""Developer probe: achievable HBM bandwidth of plain streaming patterns on this
GPU (torch kernels): fill (write only), copy (1 read : 1 write), sum (read
only), and a 1 read : 2 write pattern like the polar-gradient kernel's."""
import torch, time
n = 1 << 29  # 2 GiB of float32
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda"); c2 = torch.empty(2 * n, device="cuda")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
dt = t(lambda: a.fill_(1.0)); print("fill   write %.2f TB/s" % (4 * n / dt / 1e12))
dt = t(lambda: b.copy_(a)); print("copy   r+w   %.2f TB/s (write %.2f)" % (8 * n / dt / 1e12, 4 * n / dt / 1e12))
dt = t(lambda: a.sum()); print("sum    read  %.2f TB/s" % (4 * n / dt / 1e12))
v = c2.view(n, 2)
dt = t(lambda: torch.stack((a, a), dim=1, out=v)); print("1r:2w  total %.2f TB/s (write %.2f)" % (12 * n / dt / 1e12, 8 * n / dt / 1e12))

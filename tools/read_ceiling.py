"""How fast can a read-only streaming kernel go on this box?  torch's int64 sum over 32 GiB as the yardstick."""
import torch, time
n = 32 * (1 << 30)
buf = torch.empty(n // 8, dtype=torch.int64, device="cuda").random_()
for _ in range(2):
    buf.sum()
torch.cuda.synchronize()
best = 1e9
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s = buf.sum(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"torch int64 sum over 32 GiB: {best:.3f} ms = {n / best / 1e6:.0f} GB/s")
x = torch.empty(n // 16, dtype=torch.int64, device="cuda"); y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize(); best = 1e9
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"torch copy 16 GiB -> 16 GiB: {best:.3f} ms = {n / best / 1e6:.0f} GB/s (read+write)")

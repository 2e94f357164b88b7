"""What a plain read + write stream reaches on this box (the describe stage's kernels are half reads, half writes):
torch copy / sign over 1 GiB, and a pure read (sum)."""
import torch

x = torch.randn(1 << 28, device="cuda")   # 1 GiB of fp32
y = torch.empty_like(x)


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gib = float(1 << 30)
ms = t(lambda: y.copy_(x))
print(f"copy 1 GiB -> 1 GiB: {ms:.3f} ms = {2 * gib / ms / 1e9:.2f} TB/s (read + write)")
ms = t(lambda: torch.sign(x, out=y))
print(f"sign 1 GiB -> 1 GiB: {ms:.3f} ms = {2 * gib / ms / 1e9:.2f} TB/s (read + write)")
ms = t(lambda: x.sum())
print(f"sum of 1 GiB: {ms:.3f} ms = {gib / ms / 1e9:.2f} TB/s (read)")
ms = t(lambda: y.fill_(1.0))
print(f"fill 1 GiB: {ms:.3f} ms = {gib / ms / 1e9:.2f} TB/s (write)")

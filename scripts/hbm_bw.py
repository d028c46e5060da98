#!/usr/bin/env python
"""What the box's HBM delivers to plain streaming kernels (torch elementwise ops): write-only, read-only, copy, and the
conv1-forward mix (1 byte read : 1.83 bytes written)."""
import torch
torch.cuda.set_device(0)
n = 1 << 31                      # 8 GiB of fp32
x = torch.empty(n // 4 * 4 // 4, dtype=torch.float32, device='cuda')
y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
nb = x.numel() * 4
print('write-only (fill)  %.2f TB/s' % (nb / t(lambda: x.fill_(1.0)) / 1e12))
print('write-only (zero_) %.2f TB/s' % (nb / t(lambda: x.zero_()) / 1e12))
print('read-only (sum)    %.2f TB/s' % (nb / t(lambda: x.sum()) / 1e12))
print('copy (r+w)         %.2f TB/s' % (2 * nb / t(lambda: y.copy_(x)) / 1e12))
u = torch.empty(x.numel(), dtype=torch.uint8, device='cuda')
print('u8 -> f32 convert  %.2f TB/s (1 B read : 4 B written)' % (5 * u.numel() / t(lambda: torch.ops.aten.copy_(x, u)) / 1e12))

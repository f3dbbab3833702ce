"""Development aid: pinned host -> device copy time of one 256-window batch and of its poses back."""
import torch, time
x = torch.empty((256, 243, 17, 3), dtype=torch.float32).pin_memory()
d = torch.empty_like(x, device="cuda")
for _ in range(5): d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): d.copy_(x, non_blocking=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("H2D of one 256-window batch (%.1f MB, pinned): %.3f ms = %.1f GB/s" % (x.numel() * 4 / 1e6, ms, x.numel() * 4 / ms / 1e6))
o = torch.empty((256, 1, 17, 3), dtype=torch.float32, device="cuda"); oh = torch.empty_like(o, device="cpu").pin_memory()
e0.record()
for _ in range(50): oh.copy_(o, non_blocking=True)
e1.record(); torch.cuda.synchronize()
print("D2H of its poses (%.0f KB): %.3f ms" % (o.numel() * 4 / 1e3, e0.elapsed_time(e1) / 50))

"""What this box's HBM gives a plain kernel: fill (write-only), copy (read+write), sum (read-only) of the
K value array size (1.208 GB).  Context for the assembly roofline (its traffic is 78 % writes)."""
import torch
n = 151044100
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print(f"fill   {ms*1e3:7.1f} us  {8*n/ms/1e9:6.2f} TB/s written")
ms = t(lambda: a.zero_()); print(f"zero_  {ms*1e3:7.1f} us  {8*n/ms/1e9:6.2f} TB/s written")
ms = t(lambda: b.copy_(a)); print(f"copy   {ms*1e3:7.1f} us  {16*n/ms/1e9:6.2f} TB/s read+written")
ms = t(lambda: a.sum()); print(f"sum    {ms*1e3:7.1f} us  {8*n/ms/1e9:6.2f} TB/s read")

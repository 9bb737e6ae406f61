"""Practical HBM ceiling probe on the GPU box: in-place read-modify-write and copy streams (torch elementwise
kernels), at the footprint of the block-lower covariance (about 200 MB) and larger.  Prints GB/s of
read+write traffic.  Used to put the P-GEMM's achieved bandwidth in context."""
import json
import torch

def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

out = {}
for mb in (200, 400, 1600):
    n = mb * 1000 * 1000 // 4
    x = torch.zeros(n, device="cuda", dtype=torch.float32)
    y = torch.empty_like(x)
    t = timed(lambda: x.sub_(1.0))
    out[f"rmw_inplace_{mb}MB"] = round(2 * n * 4 / t / 1e9, 1)
    t = timed(lambda: y.copy_(x))
    out[f"copy_{mb}MB"] = round(2 * n * 4 / t / 1e9, 1)
    t = timed(lambda: x.sum())
    out[f"read_{mb}MB"] = round(n * 4 / t / 1e9, 1)
    t = timed(lambda: x.fill_(1.0))
    out[f"write_{mb}MB"] = round(n * 4 / t / 1e9, 1)
    del x, y
print(json.dumps(out))

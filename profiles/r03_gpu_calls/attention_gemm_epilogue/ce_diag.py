import sys, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from olmoasr_amd import ops
torch.manual_seed(0)
V, ld, rows = 51865, 51968, 896
for scale in (0.05, 1.0, 4.0):
    x = (torch.randn(rows, ld, device="cuda") * scale).bfloat16()
    t = torch.randint(0, V, (rows,), device="cuda")
    t[::4] = 51864
    xf = x[:, :V].float().requires_grad_(True)
    loss = F.cross_entropy(xf, t, ignore_index=51864) * 1024.0
    loss.backward()
    ref = xf.grad.bfloat16()
    y = x.clone()
    l, _ = ops.cross_entropy_(y, V, t, 51864, gscale=1024.0)
    got = y[:, :V]
    neq = (got != ref)
    rel = ((got.float() - ref.float()).abs() / (ref.float().abs() + 1e-30))
    print(f"scale {scale}: loss {float(l):.6f} vs {float(loss)/1024:.6f}; elements differing from bf16(torch fp32 grad): {int(neq.sum())} of {neq.numel()} ({float(neq.float().mean()):.2e}); max rel {float(rel[neq].max()) if neq.any() else 0:.3g}; rel-L2 {float((got.float()-ref.float()).norm()/ref.float().norm()):.3g}")

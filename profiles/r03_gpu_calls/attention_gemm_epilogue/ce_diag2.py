import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.nn.functional as F
from oracle import mel_oracle as me, model_oracle as mo
from olmoasr_amd.model import OLMoASR
from olmoasr_amd import ops
import test_gpu_autograd as T
dims = mo.VARIANTS["tiny"]
sd = mo.init_state_dict(dims, seed=0)
pcm, ti, ty, tl = mo.synthetic_batch([0, 1])
mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
DEV = "cuda"
net = OLMoASR(T._dims(dims), device=DEV, seed=0, compute_dtype="bfloat16"); net.load_state_dict(sd)
logits = net(mel.to(DEV), ti.to(DEV), T._mask(tl).to(DEV)).detach()
V = logits.shape[-1]; PAD = V - 1
xf = logits.view(-1, V).clone().requires_grad_(True)
tg = ty.to(DEV).view(-1)
loss = F.cross_entropy(xf, tg, ignore_index=PAD)
(loss * 1024.0).backward()
ref = xf.grad.bfloat16()
y = torch.zeros(xf.shape[0], 51968, device=DEV, dtype=torch.bfloat16)
y[:, :V] = logits.view(-1, V).bfloat16()
assert torch.equal(y[:, :V].float(), logits.view(-1, V))
l, _ = ops.cross_entropy_(y, V, tg, PAD, gscale=1024.0)
got = y[:, :V]
valid = (tg != PAD)
print("valid rows", int(valid.sum()), "loss", float(l), float(loss))
neq = got != ref
print("differing elements", int(neq.sum()), "of", int(valid.sum()) * V, "in valid rows;", "rel-L2", float((got.float() - ref.float()).norm() / ref.float().norm()))
r = neq.nonzero()
print("rows with diffs", torch.unique(r[:, 0]).numel(), "example", got[neq][:5].float().tolist(), ref[neq][:5].float().tolist())
print("pad cols of y nonzero:", int((y[:, V:] != 0).sum()))
dd = (got.float() - ref.float()).abs()
top = torch.topk(dd.flatten(), 8)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    r_, c_ = divmod(i, V)
    print("diff", v, "row", r_, "col", c_, "got", float(got[r_, c_]), "ref", float(ref[r_, c_]), "is target", int(tg[r_]) == c_, "fp32 ref", float(xf.grad[r_, c_]))

import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.nn.functional as F
from oracle import mel_oracle as me, model_oracle as mo
from olmoasr_amd.model import OLMoASR
from olmoasr_amd import _native as N
from olmoasr_amd.config.model_dims import ModelDimensions
dims = mo.VARIANTS["tiny"]
sd = mo.init_state_dict(dims, seed=0)
pcm, ti, ty, tl = mo.synthetic_batch([0, 1])
mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
import test_gpu_autograd as T
DEV = "cuda"
def fused(pp):
    N.lib().oasr_attention_set_pingpong(pp)
    net = OLMoASR(T._dims(dims), device=DEV, seed=0, compute_dtype="bfloat16"); net.load_state_dict(sd); net.zero_grad()
    net.loss_and_backward(mel.to(DEV), ti.to(DEV), ty.to(DEV), tl.to(DEV), loss_scale=1024.0)
    return {n: p.grad.clone() for n, p in net.named_parameters()}
def autog(pp):
    N.lib().oasr_attention_set_pingpong(pp)
    net = OLMoASR(T._dims(dims), device=DEV, seed=0, compute_dtype="bfloat16"); net.load_state_dict(sd); net.zero_grad()
    c = dict(mel=mel, tokens=ti, targets=ty, text_len=tl)
    loss, _ = T._ref_loss(net, c)
    (loss * 1024.0).backward()
    return {n: p.grad.clone() for n, p in net.named_parameters()}
def cmp(a, b, tag):
    rs = sorted(((float((a[n] - b[n]).norm() / (b[n].norm() + 1e-20)), n) for n in a), reverse=True)
    print(tag, "worst rel-L2", rs[:3], "median", rs[len(rs) // 2][0], "min", rs[-1])
f0, a0 = fused(0), autog(0)
cmp(a0, f0, "autograd vs fused, pp=0")
sys.exit(0)
cmp(f1, f1b, "fused pp=1 twice       ")
cmp(f1, f0, "fused pp=1 vs pp=0     ")
cmp(a1, f1, "autograd vs fused, pp=1")
cmp(a0, f0, "autograd vs fused, pp=0")
cmp(a1, a0, "autograd pp=1 vs pp=0  ")

import os, sys, torch
sys.path.insert(0, "/root/repo")
from olmoasr_amd import ops
out = sys.argv[1]
res = {}
for name, B, H, Tq, Tk in (("enc", 2, 3, 1500, 1500), ("cross", 2, 2, 448, 1500), ("odd", 1, 2, 130, 77), ("tiny", 2, 2, 5, 200), ("big", 1, 1, 300, 1000)):
    d = H * 64
    g = torch.Generator(device="cuda").manual_seed(5)
    qb = torch.randn(B, Tq, d, device="cuda", generator=g).bfloat16()
    kvb = torch.randn(B, Tk, 2 * d, device="cuda", generator=g).bfloat16()
    q = qb.unflatten(2, (H, 64))
    k, v = (kvb[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(2))
    d_o = torch.randn(B, Tq, d, device="cuda", generator=g).bfloat16()
    o, lse, o_lo = ops.attention_fwd(q, k, v, None, False, want_o_lo=True)
    cs = torch.zeros(d, device="cuda"); cv = torch.zeros(d, device="cuda")
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, d_o, None, False, o_lo=o_lo, dq_colsum=cs, dv_colsum=cv)
    res[name] = [t.cpu() for t in (dq, dk, dv, cs, cv)]
torch.cuda.synchronize()
if os.path.exists(out):
    ref = torch.load(out)
    for n in res:
        for i, (x, y) in enumerate(zip(res[n], ref[n])):
            print(n, "dq dk dv cs cv".split()[i], "equal" if torch.equal(x, y) else f"DIFF max {float((x.float()-y.float()).abs().max())} nan {bool(torch.isnan(x.float()).any())}")
else:
    torch.save(res, out)
    print("saved")

import sys, os, torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.getcwd())
from olmoasr_amd import _native as N, ops
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
torch.manual_seed(0)
BF=torch.bfloat16
ok=True
for (M,Nn,K,ta,tb) in [(3000,1152,384,False,False),(4096,1024,1024,False,True),(2560,512,2048,False,False),(1000,384,1536,False,True)]:
    A=(torch.randn(K,M) if ta else torch.randn(M,K)).to(BF).cuda()
    B=((torch.randn(K,Nn) if tb else torch.randn(Nn,K))*0.05).to(BF).cuda()
    outs=[]
    for v in (0,1,2,3,6,7):
        N.lib().oasr_gemm_set_variant(v)
        N.lib().oasr_gemm_force_general(4)   # force the ping-pong kernel
        out=torch.full((M,Nn),float('nan'),device='cuda',dtype=BF)
        ops.gemm(A,B,M,Nn,K,ta=ta,tb=tb,out=out)
        outs.append(out)
    N.lib().oasr_gemm_force_general(0); N.lib().oasr_gemm_set_variant(-1)
    ref=(A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
    e0=float((outs[0].float()-ref).abs().max()); e1=float((outs[-1].float()-ref).abs().max())
    same=all(torch.equal(outs[0],o) for o in outs[1:])
    print(M,Nn,K,ta,tb,'err',e0,e1,'bit-identical',same)
    ok = ok and same
print('VARIANT_OK' if ok else 'VARIANT_MISMATCH')

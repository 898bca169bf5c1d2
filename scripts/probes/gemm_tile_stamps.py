"""Reads the per-workgroup cycle records of a library built from scripts/probes/gemm_tile_stamps_patch.py (OASR_LIB=...): prologue /
K loop / epilogue cycles per 256 x 256 tile, effective clock, gap between consecutive workgroups of a CU.  usage: OASR_LIB=<lib> python scripts/probes/gemm_tile_stamps.py [M]"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
from olmoasr_amd import ops
lib = ctypes.CDLL(os.environ["OASR_LIB"])
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
def run(name, N, K, resid=False, act=0):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(BF)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=BF)
    r = torch.randn(M, N, device="cuda").to(BF) if resid else None
    for _ in range(6):
        ops.gemm(A, B, M, N, K, bias=bias, act=act, resid=r, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.gemm(A, B, M, N, K, bias=bias, act=act, resid=r, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    buf = np.zeros(8 * 16384, dtype=np.uint64)
    lib.oasr_gemm_dbg_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    x = buf.reshape(16384, 8)[:min(ntile, 16384)]
    r0, r1 = x[:, 0].astype(np.float64), x[:, 1].astype(np.float64)
    pro, loop, epi, tot = (x[:, i].astype(np.float64) for i in (2, 3, 4, 6))
    key = x[:, 5]
    cu = {}
    for i in range(len(x)):
        k_ = int(key[i])
        cu.setdefault((k_ >> 32, (k_ >> 8) & 0xF, (k_ >> 12) & 0x1, (k_ >> 13) & 0x7), []).append(i)
    gaps, life = [], []
    for k_, idx in cu.items():
        idx = sorted(idx, key=lambda i: r0[i])
        for a_, b_ in zip(idx[:-1], idx[1:]):
            gaps.append((r0[b_] - r1[a_]) * 10.0)  # ns (100 MHz ticks)
        life += [(r1[i] - r0[i]) * 10.0 for i in idx]
    gaps = np.array(gaps); life = np.array(life)
    clk = (tot / np.maximum(1, (r1 - r0))).mean() * 0.1
    print(f"{name:28s} {ms:7.3f} ms {2.0*M*N*K/ms/1e9:7.0f} TF/s | tiles {ntile} on {len(cu)} CUs | per tile (cycles): prologue {pro.mean():7.0f} loop {loop.mean():7.0f} epilogue {epi.mean():7.0f} total {tot.mean():7.0f} | clock {clk:5.2f} GHz | tile lifetime {life.mean()/1e3:6.2f} us, gap to next WG on the CU {gaps.mean()/1e3:6.2f} us (p10 {np.percentile(gaps,10)/1e3:5.2f}, p90 {np.percentile(gaps,90)/1e3:5.2f})")
run("N=3072 K=1024 bias only", 3072, 1024)
run("attn.out N=1024 K=1024 +res", 1024, 1024, resid=True)
run("mlp.0 N=4096 K=1024 gelu", 4096, 1024, act=1)
run("mlp.2 N=1024 K=4096 +res", 1024, 4096, resid=True)

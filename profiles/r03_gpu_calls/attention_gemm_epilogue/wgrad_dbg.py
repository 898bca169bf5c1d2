import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
from olmoasr_amd import ops
lib = ctypes.CDLL(os.environ["OASR_LIB"])
from olmoasr_amd import _native as N
N.lib().oasr_gemm_force_general(4)  # 256x256 ping-pong
BF = torch.bfloat16
T = 192000
def run(name, M, N, split):
    # wgrad: dW[M=d_out][N=d_in] = sum_t dY[t][M] * X[t][N]  (TN: both operands token-major)
    A = (torch.randn(T, M, device="cuda") * 0.5).to(BF)
    B = (torch.randn(T, N, device="cuda") * 0.5).to(BF)
    o = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(A, B, M, N, T, ta=True, tb=True, out_f32=o, atomic=True, split_k=split)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.gemm(A, B, M, N, T, ta=True, tb=True, out_f32=o, atomic=True, split_k=split)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    buf = np.zeros(8 * 16384, dtype=np.uint64)
    lib.oasr_gemm_dbg_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    x = buf.reshape(16384, 8)[:ntile * split].astype(np.float64)
    x = x[x[:, 6] > 0]
    print(f"{name:30s} split {split:3d} {ms:7.3f} ms {2.0*M*N*T/ms/1e9:7.0f} TF/s | records {len(x)} | cycles: prologue {x[:,2].mean():7.0f} loop {x[:,3].mean():8.0f} epilogue {x[:,4].mean():7.0f} ({100*x[:,4].mean()/x[:,6].mean():4.1f} %)")
run("wgrad [1024 x 1024]", 1024, 1024, 16)
run("wgrad [4096 x 1024]", 4096, 1024, 8)
run("wgrad [1024 x 4096]", 1024, 4096, 4)

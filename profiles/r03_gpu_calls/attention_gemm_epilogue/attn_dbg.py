import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
from olmoasr_amd import ops, _native
B, H, d, Tq, Tk = 32, 16, 1024, 1500, 1500
qkv = torch.randn(B, Tq, 3 * d, device="cuda").bfloat16()
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
d_o = torch.randn(B, Tq, d, device="cuda").bfloat16()
o, lse, o_lo = ops.attention_fwd(q, k, v, None, False, want_o_lo=True)
for _ in range(3):
    ops.attention_bwd(q, k, v, o, lse, d_o, None, False, o_lo=o_lo)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["OASR_LIB"])
buf = np.zeros(131072, dtype=np.uint64)
rc = lib.oasr_attn_dbg_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
x = buf[:65536 // 1].copy()[:1024 * 64].reshape(1024, 8, 8).astype(np.float64)
z = buf[65536:65536 + 1024 * 16].reshape(1024, 8, 2).astype(np.float64)
print("s_sleep 100 (6400 clocks):", z[:, :, 0].mean(), "ticks; 256 dependent v_add_f32:", z[:, :, 1].mean(), "ticks")
n = x[:, :, 6]
print("rc", rc, "segments per kind", n[0, 0])
for name, i in (("compute", 0), ("barrier after compute", 1), ("load", 2), ("barrier after load", 3)):
    per = x[:, :, i] / n
    print(f"{name:24s} mean {per.mean():8.1f}  waves0-3 {per[:, :4].mean():8.1f}  waves4-7 {per[:, 4:].mean():8.1f}  min {per.min():8.1f} max {per.max():8.1f}")
print("WG lifetime: sclk ticks", x[:, :, 4].mean(), "realtime ticks (100 MHz)", x[:, :, 5].mean(), "-> sclk GHz", (x[:, :, 4] / x[:, :, 5]).mean() * 0.1)
loop = (x[:, :, 0] + x[:, :, 1] + x[:, :, 2] + x[:, :, 3])
print("loop ticks per WG", loop.mean(), "per step", (loop / n).mean())

print("calib: 32 MFMAs (1024 sclk cycles if alone on the SIMD) =", x[:, :4, 7].mean(), "ticks; min", x[:, :4, 7].min())

"""mlp.0 forward GEMM (bias + GELU, storing GELU' for the backward: act = 2; and act = 1 storing the pre-activation) at the bench's launch shape,
timed with HIP events; prints a checksum of both outputs so that two builds of the library can be compared bit for bit (OASR_LIB selects the build)."""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
    d = 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, d, device="cuda", generator=g).to(BF)
    w = (torch.randn(4 * d, d, device="cuda", generator=g) * 0.03).to(BF)
    bias = torch.randn(4 * d, device="cuda", generator=g)
    out = torch.empty(M, 4 * d, device="cuda", dtype=BF)
    pre = torch.empty(M, 4 * d, device="cuda", dtype=BF)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for act in (2, 1):
        fn = lambda: ops.gemm(x, w, M, 4 * d, d, bias=bias, act=act, out=out, out_pre=pre)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            ev[0].record()
            for _ in range(4):
                fn()
            ev[1].record()
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) / 4)
        ts.sort()
        sub = slice(0, 4096)
        crc = zlib.crc32(out[sub].view(torch.int16).cpu().numpy().tobytes()) ^ zlib.crc32(pre[sub].view(torch.int16).cpu().numpy().tobytes())
        print(f"act={act} M={M} N=4096 K=1024: med {ts[2]:.3f} ms min {ts[0]:.3f} ms = {2.0 * M * 4 * d * d / ts[2] / 1e9:.0f} TF/s  crc {crc:08x}", flush=True)


if __name__ == "__main__":
    main()

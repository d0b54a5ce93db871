"""Where does the NT GEMM main loop wait?  Times the bare product with the activation operand laid
out normally, and with every activation row aliased to one row (row stride 0: the whole operand
is 6 KiB and never leaves the caches).  The arithmetic and the LDS traffic are identical, so the
difference is what the trip to L2 / HBM costs the loop."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402
from vitk import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = B * 197
    bf = torch.bfloat16
    for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        a = torch.randn(M, K, device="cuda").to(bf)
        a0 = a[:1].expand(M, K)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
        out = torch.empty(M, N, device="cuda", dtype=bf)
        fl = 2.0 * M * N * K
        for shape in (4,):
            vitk.lib.vitk_debug_force_gemm_shape(shape)
            line = f"{name:4s} N={N:4d} K={K:4d} shape{shape}:"
            for label, act in (("normal", a), ("row-aliased", a0), ("normal", a), ("row-aliased", a0)):
                ms = timeit(lambda: ops.gemm_nt(act, w, out))
                line += f"  {label} {ms:.3f} ms {fl / ms / 1e9:5.0f} TF/s"
            print(line, flush=True)
        vitk.lib.vitk_debug_force_gemm_shape(-1)


if __name__ == "__main__":
    main()

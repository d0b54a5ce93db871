#!/usr/bin/env python3
"""Does the attention kernels' time depend on how the head slices lie in memory?

Times vitk_attn_fwd / vitk_attn_bwd through the C-ABI on the same amount of data in two layouts:
  interleaved: q, k, v are column blocks of one [T, 3D] matrix (rows of 128 bytes, 4608 bytes apart)
  dense:       every (image, head) pair is a contiguous [N, 64] block (B*H images of one head, ld = 64)
With the instrumented library (make -C vit-pytorch_amd/csrc abl) VITK_ATTN_ABL selects the ablations of attn_abl.py.
usage: attn_layout_probe.py [B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VITK_LIB", os.path.join(ROOT, "tools", "bin", "libvitk_abl.so"))
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import torch  # noqa: E402
import vitk  # noqa: E402

B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, H0, DH = 197, 12, 64
vp = ctypes.c_void_p


def run(layout, abl):
    os.environ["VITK_ATTN_ABL"] = str(abl)
    if layout == "interleaved":
        B, H = B0, H0
        D = H * DH
        qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        q, k, v = (qkv.data_ptr() + i * D * 2 for i in range(3))
        dq, dk, dv = (dqkv.data_ptr() + i * D * 2 for i in range(3))
        ld_qkv, ld_o = 3 * D, D
    else:
        B, H = B0 * H0, 1
        D = DH
        bufs = [(torch.randn(B * N, DH, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(3)]
        dbufs = [torch.empty_like(b) for b in bufs]
        q, k, v = (b.data_ptr() for b in bufs)
        dq, dk, dv = (b.data_ptr() for b in dbufs)
        ld_qkv, ld_o = DH, DH
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
    o32 = torch.empty(B * N, D, dtype=torch.float32, device="cuda")
    dout = (torch.randn(B * N, D, device="cuda") * 0.5).to(torch.bfloat16)
    lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
    delta = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
    st = vp(torch.cuda.current_stream().cuda_stream)
    tf, tb = [], []
    for it in range(6):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rc = vitk.lib.vitk_attn_fwd(vp(q), vp(k), vp(v), vp(out.data_ptr()), vp(o32.data_ptr()), vp(lse.data_ptr()),
                                    B, H, N, DH, ld_qkv, ld_o, 0.125, vitk.BF16, st)
        assert rc == 0, rc
        e[1].record()
        rc = vitk.lib.vitk_attn_bwd(vp(q), vp(k), vp(v), vp(out.data_ptr()), vp(o32.data_ptr()), vp(dout.data_ptr()),
                                    vp(lse.data_ptr()), vp(delta.data_ptr()), vp(dq), vp(dk), vp(dv), B, H, N, DH,
                                    ld_qkv, ld_o, ld_qkv, 0.125, vitk.BF16, st)
        assert rc == 0, rc
        e[2].record()
        torch.cuda.synchronize()
        tf.append(e[0].elapsed_time(e[1]))
        tb.append(e[1].elapsed_time(e[2]))
    print(f"{layout:12s} abl {abl:3d}: fwd {min(tf[1:]) * 1e3:7.1f} us  bwd {min(tb[1:]) * 1e3:7.1f} us", flush=True)


for abl in (0, 131, 67, 3):
    for layout in ("interleaved", "dense", "interleaved", "dense"):
        run(layout, abl)

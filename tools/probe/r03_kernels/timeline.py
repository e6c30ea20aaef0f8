#!/usr/bin/env python
"""Phase timeline of the residual-conv forward kernel (debug build with s_memtime stamps, `make -C ganslate_amd/csrc
timeline`): GANSLATE_HIP_LIB=build/libganslate_hip_tl.so python tools/probe/timeline.py [--nw 16|8]
Prints, for two workgroups, the cycles of prologue / per-K-step phases / epilogue per stamped wave."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GANSLATE_HIP_LIB", str(ROOT / "build" / "libganslate_hip_tl.so"))
import torch  # noqa: E402
from ganslate_amd.hip.ops import HipOps  # noqa: E402
from ganslate_amd.hip import lib as L  # noqa: E402
from ganslate_amd.nn.native.spec import ConvSpec, lower  # noqa: E402

TLW, TLS = 8, 160


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nw", type=int, default=16)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--abl", type=int, default=0, help="hconvx ablation: 1 no LDS-DMA, 2 no fragment reads, 3 no barrier")
    ap.add_argument("--x", action="store_true", help="the self-pipelined kernel (hconvx.hip): 3 waits + barrier per K-step")
    args = ap.parse_args()
    ops = HipOps()
    ops.set_option("hconvw_waves", args.nw)
    ops.set_option("hconvx", (1 + args.abl) if args.x else 0)
    dev = ops.device
    spec = ConvSpec("conv", args.cin, 256, 3, 1, 1, pad_mode="reflect")
    N, H, W = 8, 64, 64
    low = lower(spec, H, W)
    x = torch.randn(N, H, W, spec.cin_p, device=dev).to(torch.bfloat16)
    fpack = (torch.randn(low.fwd_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(spec.cout_p, device=dev)
    y = torch.empty(N, *low.out_dims, spec.cout_p, device=dev, dtype=torch.bfloat16)
    slots = ops.stat_slots(low.fwd[0], N)
    part = torch.empty(N * slots * 2 * spec.cout_p, device=dev)
    buf = torch.zeros(2 * TLW * TLS, dtype=torch.int32, device=dev)
    lib = L.load()
    dbg = lib.gs_debug_timeline_x if args.x else lib.gs_debug_timeline
    dbg.restype = C.c_int
    dbg.argtypes = [C.c_void_p]

    def fwd():
        ops.gconv_classes(low.fwd, x, fpack, bias, y, stats=part, stats_slots=slots, stats_slot0s=[0])
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    assert dbg(C.c_void_p(buf.data_ptr())) == 0
    fwd()
    torch.cuda.synchronize()
    assert dbg(C.c_void_p(0)) == 0
    t = buf.cpu().numpy().astype("int64").reshape(2, TLW, TLS) & 0xffffffff
    nk = (args.cin // 64) * 9
    for blk in range(2):
        print(f"== workgroup {blk} (nw={args.nw}, {nk} K-steps); cycles relative to the workgroup's first stamp")
        t0 = t[blk, :, 0].min()
        for w in range(TLW):
            r = (t[blk, w] - t0) & 0xffffffff
            ks = r[8:8 + 4 * nk].reshape(nk, 4)
            import numpy as np
            prev_end = np.concatenate([[r[2]], ks[:-1, 3]])
            l_read = ks[:, 0] - prev_end           # M-end barrier(ks-1) -> this step's fragment reads landed
            if args.x:      # stamps per K-step at the head of unit 2: before the vmcnt wait, after it, after the barrier
                end = ks[:, 2]
                prev = np.concatenate([[r[2]], end[:-1]])
                print(f" wave {w}: start {r[0]:6d} prologue-ready {r[1]:6d} loop-start {r[2]:6d} last-barrier {end[-1]:6d} "
                      f"epi-start {r[3]:6d} stores-issued {r[4]:6d} done {r[5]:6d}")
                print(f"    per K-step mean: barrier->next vmcnt {(ks[:, 0] - prev).mean():7.1f}  vmcnt {(ks[:, 1] - ks[:, 0]).mean():7.1f}  "
                      f"barrier {(ks[:, 2] - ks[:, 1]).mean():7.1f}  step {(end[1:] - end[:-1]).mean():7.1f}")
                if w in (0, 5):
                    for k in range(nk):
                        print(f"      ks {k:2d}: {ks[k, 0] - prev[k]:6d} {ks[k, 1] - ks[k, 0]:6d} {ks[k, 2] - ks[k, 1]:6d}")
                continue
            l_bar = ks[:, 1] - ks[:, 0]            # wait at the L barrier
            m_iss = ks[:, 2] - ks[:, 1]            # MFMA issue
            m_bar = ks[:, 3] - ks[:, 2]            # weight wait + M barrier
            step = ks[1:, 3] - ks[:-1, 3]
            print(f" wave {w}: start {r[0]:6d} prologue-ready {r[1]:6d} loop-start {r[2]:6d} loop-end {ks[-1, 3]:6d} "
                  f"epi-start {r[3]:6d} stores-issued {r[4]:6d} done {r[5]:6d}")
            print(f"    per K-step mean: L-read {l_read.mean():7.1f}  L-barrier {l_bar.mean():7.1f}  M-issue {m_iss.mean():7.1f} "
                  f" M-barrier {m_bar.mean():7.1f}  step {step.mean():7.1f} (min {step.min()}, max {step.max()})")
            if w in (0, 4):
                print("    K-step table (L-read, L-bar, M-issue, M-bar):")
                for k in range(nk):
                    print(f"      ks {k:2d}: {l_read[k]:6d} {l_bar[k]:6d} {m_iss[k]:6d} {m_bar[k]:6d}")


if __name__ == "__main__":
    main()

"""Development probe: every MBConv block of EfficientNet-B2 at the bench shape (B clips x 10 s @ 16 kHz) -
expand -> depthwise as the two-kernel chain vs the fused kernel (csrc/effnet_fused.hip), per block, HIP events."""
import argparse
import ctypes

import torch

from audiocaption_amd import _lib, build
from audiocaption_amd.effnet_encoder import EfficientNet, EfficientNetB2

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--frames", type=int, default=1001)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
build.build()
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
net = EfficientNet()
B = args.batch
T, F = (args.frames + 1) // 2, 32
dev = "cuda"


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


tot_a = tot_b = 0.0
for i, blk in enumerate(net._blocks):
    pb, pa = blk.pad
    To, Fo = (T + pb + pa - blk.k) // blk.stride + 1, (F + pb + pa - blk.k) // blk.stride + 1
    rows = B * T * F
    if blk.expand != 1:
        x = torch.randn(rows, blk.cin, device=dev)
        we = torch.randn(blk.mid, blk.cin, device=dev) / blk.cin ** 0.5
        be = torch.randn(blk.mid, device=dev) * 0.1
        wd = torch.randn(blk.k, blk.k, blk.mid, device=dev) * 0.3
        sc, sh = torch.rand(blk.mid, device=dev) + 0.5, torch.randn(blk.mid, device=dev) * 0.1
        mid = torch.empty(rows, blk.mid, device=dev)
        y1 = torch.empty(B * To * Fo, blk.mid, device=dev)
        y2 = torch.empty(B * To * Fo, blk.mid, device=dev)
        pool = torch.zeros(B, blk.mid, device=dev)

        def chain():
            EfficientNetB2._gemm(x, we, be, mid, rows, blk.mid, blk.cin, act=2)
            assert lib.ac_effnet_depthwise(P(mid), P(wd), P(sc), P(sh), P(y1), P(pool), 1.0, B, T, F, blk.mid, blk.k,
                                           blk.stride, pb, pa, S()) == 0

        def fused():
            rc = lib.ac_effnet_expand_depthwise(P(x), P(we), P(be), P(wd), P(sc), P(sh), P(y2), P(pool), 1.0, B, T, F,
                                                blk.cin, blk.mid, blk.k, blk.stride, pb, pa, S())
            assert rc == 0, rc

        ta = timeit(chain)
        try:
            tb = timeit(fused)
            d = float((y1 - y2).abs().max()) / float(y1.abs().max())
        except AssertionError as e:
            tb, d = float("nan"), float("nan")
        gb_chain = (rows * blk.cin + 2 * rows * blk.mid + B * To * Fo * blk.mid) * 4 / 1e9
        gb_fused = (rows * blk.cin + B * To * Fo * blk.mid) * 4 / 1e9
        print(f"block {i:2d} k{blk.k} s{blk.stride} {blk.cin:4d}->{blk.mid:5d} @ {T}x{F}: chain {ta:7.1f} us "
              f"({gb_chain / ta * 1e3:5.2f} TB/s)  fused {tb:7.1f} us ({gb_fused / tb * 1e3:5.2f} TB/s)  diff {d:.1e}")
        tot_a += ta
        tot_b += tb if tb == tb else ta
    T, F = To, Fo
print(f"expand+depthwise over all blocks: chain {tot_a / 1e3:.2f} ms, fused {tot_b / 1e3:.2f} ms")

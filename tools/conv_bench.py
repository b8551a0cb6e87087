#!/usr/bin/env python
"""Per-layer timing of the Cnn14 conv kernels (HIP events on the launch stream), direct vs Winograd.
Development tool: `python tools/conv_bench.py [--batch 64] [--iters 5]`."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocaption_amd import build, kernels as K

LAYERS = [  # (name, H, Hp, W, Cin, Cout, mode)
    ("b1c2", 1001, 1024, 64, 64, 64, 1), ("b2c1", 500, 512, 32, 64, 128, 0), ("b2c2", 500, 512, 32, 128, 128, 1),
    ("b3c1", 250, 256, 16, 128, 256, 0), ("b3c2", 250, 256, 16, 256, 256, 1), ("b4c1", 125, 128, 8, 256, 512, 0),
    ("b4c2", 125, 128, 8, 512, 512, 1), ("b5c1", 62, 64, 4, 512, 1024, 0), ("b5c2", 62, 64, 4, 1024, 1024, 1),
    ("b6c1", 31, 32, 2, 1024, 2048, 0), ("b6c2", 31, 32, 2, 2048, 2048, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--algos", default="direct,winograd")
    ap.add_argument("--layers", default="")
    ap.add_argument("--map-mode", type=int, default=-1, help="block -> tile mapping of the gw kernels (-1: default)")
    ap.add_argument("--force-mode", type=int, default=-1, help="run every selected layer with this epilogue mode (0/1)")
    args = ap.parse_args()
    build.build()
    B = args.batch
    dev = "cuda:0"
    tot = {a: 0.0 for a in args.algos.split(",") if a != "block1"}
    for name, H, Hp, W, Cin, Cout, mode in LAYERS:
        if args.layers and name not in args.layers.split(","):
            continue
        if args.force_mode >= 0 and mode != 2:
            mode = args.force_mode
        x = torch.randn(B * Hp, W, Cin, device=dev)
        x.view(B, Hp, W, Cin)[:, H:] = 0
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        if mode == 0:
            out = torch.empty(B * Hp, W, Cout, device=dev)
        elif mode == 1:
            out = torch.empty(B * Hp // 2, W // 2, Cout, device=dev)
        else:
            out = torch.empty(B, H, Cout, device=dev)
        gflop = 2.0 * 9 * Cin * Cout * H * W * B / 1e9
        line = f"{name} {Cin:4d}->{Cout:4d} {H}x{W} mode{mode} {gflop:7.1f} GF"
        for algo in args.algos.split(","):
            if algo == "block1":
                continue
            if algo == "direct":
                wp = K.pack_conv_weight(w)
                fn = lambda: K.conv3x3_bn_relu(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode)
            elif algo == "winograd":
                wp = K.pack_conv_weight_winograd(w)
                fn = lambda: K.conv3x3_bn_relu_winograd(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode)
            elif algo == "bf16x3":
                wp = K.pack_conv_weight_bf16x3(w)
                fn = lambda: K.conv3x3_bn_relu_bf16x3(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode)
            elif algo == "wino1d":
                if Cout % 128 and not (Cout == 64 and W % 16 == 0):
                    continue
                wp = K.pack_conv_weight_wino1d_frag(w)
                fn = lambda: K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode, args.map_mode)
            elif algo.startswith("wino43"):
                if W not in (32, 16, 8, 4, 2) or Cout % 128 or (mode == 2) != (W == 2 and name == "b6c2"):
                    continue
                wp = K.pack_conv_weight_wino43_frag(w)
                tiles = int(algo[7:]) if len(algo) > 6 else 0          # "wino43", "wino43x2", "wino43x3"
                fn = lambda: K.conv3x3_bn_relu_wino43(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode, args.map_mode,
                                                      tiles_per_wave=tiles)
            elif algo == "f16x2":
                wp, inv = K.pack_conv_weight_f16x2_frag(w)
                sc2 = (sc * inv).contiguous()
                xh = x.half()
                outh = out if mode == 2 else torch.empty_like(out, dtype=torch.float16)
                fn = lambda: K.conv3x3_bn_relu_f16x2_gw(xh, wp, sc2, sh, outh, B, Hp, H, W, Cin, Cout, mode, args.map_mode)
            else:
                wp = K.pack_conv_weight_bf16x3_frag(w)
                fn = lambda: K.conv3x3_bn_relu_bf16x3_gw(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode)
            fn()
            torch.cuda.synchronize()
            if algo == args.algos.split(",")[0]:
                ref_out = out.clone()
            else:
                got = outh.float() if algo == "f16x2" else out
                line += f" (maxdiff {float((got - ref_out).abs().max()):.1e})"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.iters
            tot[algo] += ms
            line += f" | {algo} {ms * 1000:8.1f} us {gflop / ms:7.1f} TF"
        print(line, flush=True)
    if "block1" in args.algos.split(","):   # conv block 1: conv_first + F(2,3) conv2 against the one-kernel F(4,3) form
        H, Hp = 1001, 1024
        x0 = torch.randn(B * Hp, 64, device=dev)
        x0.view(B, Hp, 64)[:, H:] = 0
        w1 = torch.randn(64, 9, device=dev) * 0.3
        w2 = torch.randn(64, 64, 3, 3, device=dev) * (2.0 / (9 * 64)) ** 0.5
        s1, t1, s2, t2 = (torch.rand(64, device=dev) + 0.5 for _ in range(4))
        mid = torch.empty(B * Hp, 64, 64, device=dev)
        o1, o2 = torch.empty(B * Hp // 2, 32, 64, device=dev), torch.empty(B * Hp // 2, 32, 64, device=dev)
        wp23, wp43 = K.pack_conv_weight_wino1d_frag(w2), K.pack_conv_weight_wino43_frag(w2)

        def two():
            K.conv3x3_first(x0, w1, s1, t1, mid, B, Hp, H)
            K.conv3x3_bn_relu_wino1d(mid, wp23, s2, t2, o1, B, Hp, H, 64, 64, 64, 1)

        for name, fn in (("conv_first + wino1d", two),
                         ("block1 fused F(4,3)", lambda: K.conv3x3_block1_wino43(x0, w1, s1, t1, wp43, s2, t2, o2, B, Hp, H)),
                         ("conv2 alone, unfused F(4,3)", lambda: K.conv3x3_block1_conv2_wino43(mid, wp43, s2, t2, o2, B, Hp, H))):
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            print(f"block 1, {name}: {s.elapsed_time(e) / args.iters * 1000:8.1f} us  (maxdiff vs two-kernel F(2,3) {float((o1 - o2).abs().max()):.1e})")
    print("total ms:", {a: round(v, 3) for a, v in tot.items()})


if __name__ == "__main__":
    main()

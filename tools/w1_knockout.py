"""Development tool: price the ingredients of the F(2,3) conv kernel's K loop by removing them one at a time.

    python tools/w1_knockout.py --build          # here (no GPU): tools/bin/libw1_ko<bits>.so for every variant
    python tools/w1_knockout.py                  # on the GPU box: per layer and variant, microseconds per launch

Bits (csrc/conv3x3_wino1d.hip, W1_KO): 1 weight loads, 2 A-fragment reads, 4 plane stores (transform + split + LDS
writes), 8 patch loads, 16 barrier, 32 MFMAs.  Results of the knocked-out variants are wrong by construction."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [0, 1, 2, 4, 8, 12, 16, 32, 3, 15, 31, 47]
FORGET = {f"ko{v}": [f"-DW1_KO={v}"] for v in (0, 512, 1024, 1, 8, 520, 15)}
# --tune: named flag sets instead of knockouts
CLK = {f"clk{v}": ["-DW1_CLK", f"-DW1_KO={v}"] for v in (0, 32, 31, 15, 3)}
TUNE = {"base": [], "prio1": ["-DW1_PRIO=1"], "prio3": ["-DW1_PRIO=3"], "prio1sgb3": ["-DW1_PRIO=1", "-DW1_SGB=3"]}
BIN = os.path.join(ROOT, "tools", "bin")


def build(tune):
    from audiocaption_amd import build as B
    os.makedirs(BIN, exist_ok=True)
    src = os.path.join(ROOT, "audiocaption_amd", "csrc", "conv3x3_wino1d.hip")
    todo = {f"ko{v}": [f"-DW1_KO={v}"] for v in VARIANTS} if not tune else (CLK if tune == "clk" else (FORGET if tune == "forget" else TUNE))
    for name, flags in todo.items():
        out = os.path.join(BIN, f"libw1_{name}.so")
        cmd = [B._hipcc(), "-x", "hip", src, "-shared", "-o", out] + flags + B.FLAGS + B.NO_PACKED_F32 + \
            ["-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, check=True, stderr=subprocess.PIPE, text=True)
        regs = [ln.split("VGPRs:")[1].split()[0] for ln in r.stderr.splitlines() if "VGPRs:" in ln and "Spill" not in ln]
        spill = [ln.split("VGPRs Spill:")[1].split()[0] for ln in r.stderr.splitlines() if "VGPRs Spill:" in ln]
        print("built", out, "VGPRs", regs, "spills", spill)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--tune", action="store_true", help="the named tuning variants instead of the knockouts")
    ap.add_argument("--clk", action="store_true", help="shader clock (cycle counter / 100 MHz counter) per variant")
    ap.add_argument("--forget", action="store_true", help="loads issued but not waited for (W1_KO 64 / 128 / 256)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="b2c1,b2c2,b3c2,b4c2,b5c2,b6c2")
    args = ap.parse_args()
    if args.build:
        return build("clk" if args.clk else ("forget" if args.forget else args.tune))
    names = list(CLK) if args.clk else (list(FORGET) if args.forget else (list(TUNE) if args.tune else [f"ko{v}" for v in VARIANTS]))
    import torch
    from audiocaption_amd import kernels as K
    from tools.conv_bench import LAYERS
    B = args.batch
    dev = "cuda:0"
    P, I = ctypes.c_void_p, ctypes.c_int
    libs = {}
    for v in names:
        lib = ctypes.CDLL(os.path.join(BIN, f"libw1_{v}.so"))
        lib.ac_conv3x3_bn_relu_wino1d.restype = I
        lib.ac_conv3x3_bn_relu_wino1d.argtypes = [P] * 5 + [I] * 8 + [P, I, I, P]
        libs[v] = lib
    print("layer   " + " ".join(f"{v:>9s}" for v in names))
    for name, H, Hp, W, Cin, Cout, mode in LAYERS:
        if name not in args.layers.split(",") or Cout % 128:
            continue
        x = torch.randn(B * Hp, W, Cin, device=dev)
        x.view(B, Hp, W, Cin)[:, H:] = 0
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        out = torch.empty({0: (B * Hp, W, Cout), 1: (B * Hp // 2, W // 2, Cout), 2: (B, H, Cout)}[mode], device=dev)
        wp = K.pack_conv_weight_wino1d_frag(w)
        line = f"{name:7s} "
        ref = None
        for v in names:
            def fn():
                rc = libs[v].ac_conv3x3_bn_relu_wino1d(x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
                                                       B, Hp, H, W, Cin, Cout, mode, -1, None, 0, 0,
                                                       torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            line += f"{e0.elapsed_time(e1) / args.iters * 1e3:9.0f} "
            if args.clk:
                buf = (ctypes.c_ulonglong * 3)()
                libs[v].ac_w1_clk_read(buf, 1)
                if buf[1]:
                    line += f"[{buf[0] / buf[1] * 0.1:.2f} GHz, {buf[1] / buf[2] * 0.01:.1f} us/wg] "
            if args.tune:   # the tuning variants must agree bit for bit
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    line += "(DIFFERS) "
        print(line, flush=True)


if __name__ == "__main__":
    main()

"""Development tool: build -D variants of the F(4,3) conv kernel (csrc/conv3x3_wino43.hip) and time them per layer.

    python tools/w4_variants.py --build [name=flags ...]   # here (no GPU): tools/bin/libw4_<name>.so
    python tools/w4_variants.py [--layers ...] [--tiles N]  # on the GPU box: microseconds per launch, variants interleaved

Knockout bits (W4_KO): 1 weight loads, 2 A-fragment reads, 4 plane stores, 8 row loads, 16 barrier, 32 MFMAs (results
are then wrong).  W4_RING / W4_ADEPTH: prefetch depths."""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "bin")
DEFAULT = {"base": [], "ring6": ["-DW4_RING=6"], "a2": ["-DW4_ADEPTH=2"], "ring6a2": ["-DW4_RING=6", "-DW4_ADEPTH=2"],
           "ko15": ["-DW4_KO=15"], "ko31": ["-DW4_KO=31"], "ko32": ["-DW4_KO=32"], "ko1": ["-DW4_KO=1"], "ko2": ["-DW4_KO=2"],
           "ko4": ["-DW4_KO=4"], "ko8": ["-DW4_KO=8"], "ko16": ["-DW4_KO=16"], "ko64": ["-DW4_KO=64"], "ko79": ["-DW4_KO=79"],
           "clk": ["-DW4_CLK"], "clk15": ["-DW4_CLK", "-DW4_KO=15"], "clk79": ["-DW4_CLK", "-DW4_KO=79"]}


def build(specs):
    from audiocaption_amd import build as B
    os.makedirs(BIN, exist_ok=True)
    for f in glob.glob(os.path.join(BIN, "libw4_*.so")):
        os.remove(f)
    src = os.path.join(ROOT, "audiocaption_amd", "csrc", "conv3x3_wino43.hip")
    procs = []
    for name, flags in specs.items():
        out = os.path.join(BIN, f"libw4_{name}.so")
        cmd = [B._hipcc(), "-x", "hip", src, "-shared", "-o", out] + flags + B.FLAGS + B.NO_PACKED_F32 + \
            ["-Rpass-analysis=kernel-resource-usage"]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        if len(procs) >= 6:
            _drain(procs)
    _drain(procs)


def _drain(procs):
    for name, pr in procs:
        err = pr.communicate()[1]
        if pr.returncode:
            print(err)
            raise SystemExit(f"{name}: hipcc failed")
        cur, rep = None, {}
        for ln in err.splitlines():
            if "Function Name:" in ln:
                cur = ln.split("conv3x3_w4_kernelI")[1].split("EEv")[0].replace("Li", "").replace("E", ",") if "w4_kernelI" in ln else None
            elif cur and " VGPRs:" in ln:
                rep[cur] = [ln.split("VGPRs:")[1].split()[0]]
            elif cur and "VGPRs Spill:" in ln:
                rep[cur].append(ln.split("VGPRs Spill:")[1].split()[0])
        print("built", name, " ".join(f"<{k}>{v[0]}/{v[1]}" for k, v in sorted(rep.items())), flush=True)
    procs.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("specs", nargs="*", help="name=-DFLAG,-DFLAG ... (default: the built-in set)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--layers", default="b2c1,b2c2,b3c1,b3c2,b4c1,b4c2,b5c1,b5c2")
    ap.add_argument("--only", default="", help="comma-separated variant names to run (default: all built ones)")
    args = ap.parse_args()
    if args.build:
        specs = {s.split("=", 1)[0]: [f for f in s.split("=", 1)[1].split(",") if f] for s in args.specs} if args.specs else DEFAULT
        return build(specs)
    import torch
    from audiocaption_amd import kernels as K
    from tools.conv_bench import LAYERS
    names = sorted(os.path.basename(f)[6:-3] for f in glob.glob(os.path.join(BIN, "libw4_*.so")))
    names.sort(key=lambda n: (n != "base", n))
    if args.only:
        names = [n for n in names if n in args.only.split(",")]
    P, I = ctypes.c_void_p, ctypes.c_int
    libs = {}
    for v in names:
        lib = ctypes.CDLL(os.path.join(BIN, f"libw4_{v}.so"))
        lib.ac_conv3x3_bn_relu_wino43.restype = I
        lib.ac_conv3x3_bn_relu_wino43.argtypes = [P] * 5 + [I] * 9 + [P, I, I, P]
        libs[v] = lib
        if v.startswith("clk"):
            lib.ac_w4_clk_read.restype = I
            lib.ac_w4_clk_read.argtypes = [P, I]
    B, dev = args.batch, "cuda:0"
    print("layer   " + " ".join(f"{v:>9s}" for v in names), flush=True)
    tot = {v: 0.0 for v in names}
    for name, H, Hp, W, Cin, Cout, mode in LAYERS:
        if name not in args.layers.split(",") or Cout % 128 or W not in (32, 16, 8, 4):
            continue
        x = torch.randn(B * Hp, W, Cin, device=dev)
        x.view(B, Hp, W, Cin)[:, H:] = 0
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        out = torch.empty({0: (B * Hp, W, Cout), 1: (B * Hp // 2, W // 2, Cout)}[mode], device=dev)
        wp = K.pack_conv_weight_wino43_frag(w)
        best = {v: 1e9 for v in names}
        for _ in range(args.rounds):
            for v in names:
                def fn():
                    rc = libs[v].ac_conv3x3_bn_relu_wino43(x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
                                                           B, Hp, H, W, Cin, Cout, mode, -1, args.tiles, None, 0, 0,
                                                           torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
                fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], s.elapsed_time(e) / args.iters * 1000)
        for v in names:
            if v.startswith("clk"):
                buf = (ctypes.c_ulonglong * 5)()
                libs[v].ac_w4_clk_read(ctypes.cast(buf, P), 1)
                n = max(buf[4], 1)
                ghz = (buf[0] + buf[1] + buf[2]) / max(buf[3], 1) / 10.0   # cycles per 100 MHz tick / 10 = GHz
                print(f"  {v}: per workgroup cycles prologue {buf[0] / n:.0f}  K loop {buf[1] / n:.0f}  epilogue {buf[2] / n:.0f}  "
                      f"({n} workgroups, {ghz:.2f} GHz)", flush=True)
        for v in names:
            tot[v] += best[v]
        print(f"{name:7s} " + " ".join(f"{best[v]:9.1f}" for v in names), flush=True)
    print("total   " + " ".join(f"{tot[v]:9.1f}" for v in names))


if __name__ == "__main__":
    main()

"""Development tool: build -D variants of the one-kernel conv block 1 (csrc/conv3x3_block1_w4.hip) and time them.

    python tools/b1_variants.py --build [name=flags ...]   # here (no GPU): tools/bin/libb1_<name>.so
    python tools/b1_variants.py [--batch 64]                # on the GPU box: microseconds per launch, variants interleaved

B1_KO bits: 1 no staging, 2 no MFMAs, 4 no epilogue stores (results are then wrong).  B1_CLK: per-tile cycle counters.
A variant whose name starts with "valu" runs conv1 as the f32 chain on the vector ALUs, every other one on the matrix cores."""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "bin")
DEFAULT = {"base": [], "ko1": ["-DB1_KO=1"], "ko2": ["-DB1_KO=2"], "ko3": ["-DB1_KO=3"], "clk": ["-DB1_CLK"],
           "clk1": ["-DB1_CLK", "-DB1_KO=1"], "clk2": ["-DB1_CLK", "-DB1_KO=2"], "ring3": ["-DB1_RING=3"], "ring9": ["-DB1_RING=9"]}


def build(specs):
    from audiocaption_amd import build as B
    os.makedirs(BIN, exist_ok=True)
    for f in glob.glob(os.path.join(BIN, "libb1_*.so")):
        os.remove(f)
    src = os.path.join(ROOT, "audiocaption_amd", "csrc", "conv3x3_block1_w4.hip")
    procs = []
    for name, flags in specs.items():
        out = os.path.join(BIN, f"libb1_{name}.so")
        cmd = [B._hipcc(), "-x", "hip", src, "-shared", "-o", out] + flags + B.FLAGS + B.NO_PACKED_F32 + \
            ["-Rpass-analysis=kernel-resource-usage"]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, pr in procs:
        err = pr.communicate()[1]
        if pr.returncode:
            print(err)
            raise SystemExit(f"{name}: hipcc failed")
        vals = [ln.split(":")[-1].split()[0] for ln in err.splitlines() if " VGPRs:" in ln or "VGPRs Spill:" in ln or "SGPRs Spill" in ln]
        print("built", name, "(VGPRs, SGPR spill, VGPR spill) fused / unfused:", vals, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("specs", nargs="*")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    if args.build:
        specs = {s.split("=", 1)[0]: [f for f in s.split("=", 1)[1].split(",") if f] for s in args.specs} if args.specs else DEFAULT
        return build(specs)
    import torch
    from audiocaption_amd import kernels as K
    names = sorted(os.path.basename(f)[6:-3] for f in glob.glob(os.path.join(BIN, "libb1_*.so")))
    names.sort(key=lambda n: (n != "base", n))
    P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_ulonglong
    libs = {}
    for v in names:
        lib = ctypes.CDLL(os.path.join(BIN, f"libb1_{v}.so"))
        for sym in ("ac_conv3x3_block1_wino43", "ac_conv3x3_block1_wino43_mfma"):
            getattr(lib, sym).restype = I
            getattr(lib, sym).argtypes = [P] * 8 + [I, I, I, P, I, I, F, U, P, P]
        if v.startswith("clk"):
            lib.ac_b1_clk_read.restype = I
            lib.ac_b1_clk_read.argtypes = [P, I]
        libs[v] = lib
    B, dev, H, Hp = args.batch, "cuda:0", 1001, 1024
    x0 = torch.randn(B * Hp, 64, device=dev)
    x0.view(B, Hp, 64)[:, H:] = 0
    w1 = torch.randn(64, 9, device=dev) * 0.3
    w2 = torch.randn(64, 64, 3, 3, device=dev) * (2.0 / (9 * 64)) ** 0.5
    s1, t1, s2, t2 = (torch.rand(64, device=dev) + 0.5 for _ in range(4))
    out = torch.empty(B * Hp // 2, 32, 64, device=dev)
    wp = K.pack_conv_weight_wino43_frag(w2)
    best = {v: 1e9 for v in names}
    # every variant's RESULT against the library's f32-chain form before anything is timed (knock-outs are wrong by design)
    ref = torch.empty_like(out)
    K.conv3x3_block1_wino43(x0, w1, s1, t1, wp, s2, t2, ref, B, Hp, H, conv1="valu")
    verdict = {}
    for _ in range(args.rounds):
        for v in names:
            def fn():
                sym = "ac_conv3x3_block1_wino43" if v.startswith("valu") else "ac_conv3x3_block1_wino43_mfma"
                rc = getattr(libs[v], sym)(x0.data_ptr(), w1.data_ptr(), s1.data_ptr(), t1.data_ptr(), wp.data_ptr(),
                                                      s2.data_ptr(), t2.data_ptr(), out.data_ptr(), B, Hp, H, None, 0, 0, 0.0, 0,
                                                      None, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            fn()
            if v not in verdict:
                d = float((out - ref).abs().max()) / float(ref.abs().max())
                verdict[v] = f"result {'ok' if d < 2e-5 else 'WRONG'} ({d:.1e} of the largest output)"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], s.elapsed_time(e) / args.iters * 1000)
    for v in names:
        line = f"{v:8s} {best[v]:9.1f} us   {verdict[v]}"
        if v.startswith("clk"):
            buf = (ctypes.c_ulonglong * 7)()
            libs[v].ac_b1_clk_read(ctypes.cast(buf, P), 1)
            n = max(buf[4], 1)
            line += (f"   per tile: K loop {buf[0] / n:.0f} cycles, epilogue {buf[1] / n:.0f}; {n} tiles, "
                     f"{buf[2] / max(buf[3], 1) / 10.0:.2f} GHz; step 0 {buf[5] / n:.0f}, step 1 {buf[6] / n:.0f}")
        print(line, flush=True)


if __name__ == "__main__":
    main()

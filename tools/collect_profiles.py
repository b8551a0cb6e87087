"""Development tool: copy the round's profile set from gpurun_out/ (written by tools/round_profiles.sh on the GPU box) into
profiles/ and derive profiles/rNN_traffic_<tier>.json (HBM bytes, matrix-pipe busy fraction and clock of the dominant conv
kernel) from the four --pmc passes and the kernel stats of the same command.

    python tools/collect_profiles.py [r03] [wino1d]
"""
import glob
import json
import os
import re
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
TIER = sys.argv[2] if len(sys.argv) > 2 else "wino43"
src, dst = "gpurun_out", "profiles"
names = {f"{R}_bench_final.json": f"{R}_bench.json", f"{R}_bench_details.json": None, f"{R}_kernel_stats.txt": None,
         f"{R}_train_kernel_stats.txt": None, f"{R}_train_bench.json": None, f"{R}_effb2_bench.json": None,
         f"{R}_effb2_30s_beam4.json": None, f"{R}_effb2_kernel_stats.txt": None, f"{R}_traffic_effb2.json": None,
         f"{R}_bench_winograd.json": None, f"{R}_kernel_stats_winograd.txt": None}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b or a))
    else:
        print("missing", a)
for d in (f"{R}_pmc_{TIER}", f"{R}_pmc_effb2"):
    os.makedirs(os.path.join(dst, d), exist_ok=True)
    for f in glob.glob(os.path.join(src, d, "*.txt")):
        shutil.copy(f, os.path.join(dst, d, os.path.basename(f)))

# the dominant kernel of bench.py's roofline: conv2 + BN + ReLU + 2x2 pool of blocks 2-5 (MODE_POOL = 1), 4 launches per step
INST = {"wino43": r"conv3x3_w4_kernel<1, ", "wino1d": r"conv3x3_w1_kernel<1, 4, ", "f16x2": r"conv3x3_gw_kernel<128, 1, 1, 1, 256, false, 9, (8|4)>"}[TIER]


def pmc(counter):
    n, tot = 0, 0.0
    for line in open(os.path.join(dst, f"{R}_pmc_{TIER}", counter + ".txt")):
        parts = line.split(None, 4)
        if len(parts) == 5 and parts[0].isdigit() and re.search(INST, parts[4]):
            n += int(parts[0])
            tot += float(parts[1])
    return tot / n


def algorithmic_bytes(batch=64, frames=1001, taps=12):
    """Input once + pooled output once + packed weights once, mean over the four launches (f32 activations; the Winograd
    weights are 12 (F(2,3)) / 18 (F(4,3)) transformed taps x (hi, lo) bf16 per (Cin, Cout) pair)."""
    total, h, w, c = 0, frames // 2, 32, 128
    for _ in range(4):
        total += batch * h * w * c * 4 + batch * (h // 2) * (w // 2) * c * 4 + c * c * taps * 2 * 2
        h, w, c = h // 2, w // 2, c * 2
    return total // 4


fetch, write, busy, gui = pmc("FETCH_SIZE"), pmc("WRITE_SIZE"), pmc("SQ_VALU_MFMA_BUSY_CYCLES"), pmc("GRBM_GUI_ACTIVE")
calls, total = 0, 0.0
for line in open(os.path.join(dst, f"{R}_kernel_stats.txt")):
    m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+void \(anonymous namespace\)::" + INST, line)
    if m:
        calls += int(m.group(1))
        total += float(m.group(2))
avg_us = total / calls
hbm = (2 * fetch + write) * 1024   # FETCH_SIZE counts 32-byte requests in 64-byte units on gfx950 (MI355X_MICROARCH.md), KiB
cycles = 1024 * gui / 8
alg = algorithmic_bytes(taps=18) if TIER == "wino43" else (algorithmic_bytes() if TIER == "wino1d" else 172000000)
path = os.path.join(dst, f"{R}_traffic_{TIER}.json")
t = {"kernel": INST, "launches_per_step": 4,
     "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write, "hbm_bytes_per_launch": int(hbm),
     "algorithmic_bytes_per_launch": alg, "hbm_over_algorithmic": hbm / alg,
     "hbm_tb_per_s": hbm / (avg_us * 1e-6) / 1e12,
     "mfma_busy_frac": busy / cycles, "effective_clock_ghz": (gui / 8) / (avg_us * 1e-6) / 1e9,
     "rocprof_avg_launch_us": avg_us,
     "comment": "mean over the four MODE_POOL launches of a step (blocks 2-5).  algorithmic = f32 input once + pooled f32 "
                "output once + packed weights once.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB from separate --pmc "
                "passes; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = matrix pipe busy fraction."}
json.dump(t, open(path, "w"), indent=1)
b = json.load(open(os.path.join(dst, f"{R}_bench.json")))
print("headline", b["value"], b["ms_per_step"], "roofline", b["roofline"]["frac"], b["roofline"]["avg_launch_ms"], "rocprof us", avg_us)
for k in sorted(b):
    if k.startswith(("value_", "latency_", "train_", "effb2_", "steady_", "ms_blocking", "logmel_")):
        print(" ", k, b[k])
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"].get("cores"))
for f in (f"{R}_effb2_30s_beam4.json", f"{R}_effb2_bench.json", f"{R}_train_bench.json"):
    if os.path.exists(os.path.join(dst, f)):
        e = json.load(open(os.path.join(dst, f)))
        print(f, e["value"], e["ms_per_step"])
print(json.dumps({k: t[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "hbm_over_algorithmic", "hbm_tb_per_s",
                                    "mfma_busy_frac", "effective_clock_ghz", "rocprof_avg_launch_us")}))

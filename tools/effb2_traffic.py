"""Development tool: HBM bytes of one EfficientNet-B2 encoder forward from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE; profiles/pmc_summary.py tables) of `python tools/effb2_bench.py --method greedy --steps 1`.

    python tools/effb2_traffic.py FETCH_SIZE.txt WRITE_SIZE.txt forwards batch > profiles/rNN_traffic_effb2.json
"""
import json
import sys

ENCODER = ("logmel_kernel", "clamp_top_db", "block_max", "stem_kernel", "depthwise_kernel", "expand_dw_kernel",
           "pointwise_kernel", "pw_bf16x3_kernel", "se_gate", "gemm_", "rows_mean_w", "mean_lens", "FillFunctor<float>")


def total(path):
    s = 0.0
    for line in open(path):
        parts = line.split(None, 4)
        if len(parts) < 5 or not parts[0].isdigit():
            continue
        if any(k in parts[4] for k in ENCODER):
            s += float(parts[1])
    return s


fetch, write = total(sys.argv[1]), total(sys.argv[2])
forwards, batch = int(sys.argv[3]), int(sys.argv[4])
hbm = (2.0 * fetch + write) * 1024.0 / forwards
print(json.dumps({
    "scope": f"EfficientNet-B2 encoder forward (log-mel + stem + 23 MBConv blocks + head), batch {batch} x 10 s @ 16 kHz",
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python tools/effb2_bench.py "
              f"--method greedy --steps 1` ({forwards} encoder forwards in the run), summed over the encoder's kernels "
              "(tools/round_profiles.sh, tools/effb2_traffic.py)",
    "fetch_size_kib_per_forward": fetch / forwards, "write_size_kib_per_forward": write / forwards,
    "correction": "gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md HBM section): fetch "
                  "bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE uncorrected",
    "hbm_bytes_per_forward": hbm, "hbm_bytes_per_clip": hbm / batch, "algorithmic_bytes_per_clip": 100000000.0,
    "ratio_to_algorithmic": hbm / batch / 1e8}, indent=1))

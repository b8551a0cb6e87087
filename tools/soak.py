"""Development tool: soak test of the throughput schedule - N steps of forward_async (pair decode while the decode stream is
busy, encoders and chains on two streams) over four rotating batches, every result compared BIT FOR BIT with the blocking
call's result for that batch; then the same with mixed batch sizes (one clip ... full batch: K-sliced and plain launches,
lone and grouped chains).  Prints the number of mismatching results (must be 0)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the blocking call decodes with the one-launch kernel by default (round 5), whose logits differ from the launch chain's in the
# last bits (another summation order); the bit-for-bit comparison below is about the SCHEDULE, so both sides use the chain
os.environ.setdefault("AUDIOCAPTION_GREEDY", "chain")
import torch

import audiocaption_amd as A
from audiocaption_amd import build, procedural as P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--method", default="greedy", choices=["greedy", "beam"])
    args = ap.parse_args()
    build.build()
    vocab = 4368
    model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
    model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
    model = model.eval().cuda()
    L = int(32000 * args.seconds)
    sizes = [args.batch, args.batch, args.batch, args.batch, 1, 3, 17, args.batch // 2]
    inputs = []
    for k, n in enumerate(sizes):
        w = torch.from_numpy(P.synthetic_wav(n, L, seed=50 + k, varied=True)).cuda()
        lens = [L - 3200 * ((i * 7 + k) % 5) for i in range(n)] if k >= 4 else [L] * n
        inputs.append({"mode": "inference", "wav": w, "wav_len": lens, "specaug": False, "sample_method": args.method, "beam_size": 3, "max_length": 20})
    keys = ("seq", "logit", "attn_emb") if args.method == "greedy" else ("seq", "attn_emb")   # beam search leaves logit unset
    with torch.no_grad():
        want = []
        for d in inputs:
            o = model(dict(d))
            want.append({k: o[k].clone() for k in keys})
        bad = 0
        for phase, pick in (("uniform batches", lambda i: i % 4), ("mixed sizes", lambda i: (i * 5 + i // 3) % len(inputs))):
            pend = []
            for i in range(args.steps):
                j = pick(i)
                pend.append((j, model.forward_async(dict(inputs[j]))))
                if len(pend) >= 6:   # a bounded queue, like a server: results are taken while later batches are in flight
                    j0, p0 = pend.pop(0)
                    g = p0.result()
                    ok = all(torch.equal(g[k], want[j0][k]) for k in keys)
                    bad += int(not ok)
            for j0, p0 in pend:
                g = p0.result()
                bad += int(not all(torch.equal(g[k], want[j0][k]) for k in keys))
            print(f"{phase}: {args.steps} steps, mismatching results so far: {bad}", flush=True)
        if args.method == "greedy":
            # the blocking call on the one-launch decode: the same batch must give the same bits call after call (the cluster
            # exchange sums its four partials in part order) and the same ids as the chain
            os.environ["AUDIOCAPTION_GREEDY"] = "auto"
            first = {}
            for i in range(args.steps // 2):
                j = (i * 5 + i // 3) % len(inputs)
                o = model(dict(inputs[j]))
                if j not in first:
                    first[j] = {k: o[k].clone() for k in keys}
                    bad += int(not torch.equal(o["seq"], want[j]["seq"]))
                else:
                    bad += int(not all(torch.equal(o[k], first[j][k]) for k in keys))
            print(f"one-launch decode, blocking: {args.steps // 2} calls, mismatching results so far: {bad}", flush=True)
    print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())

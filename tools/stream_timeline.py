"""Development tool: where the time of a forward_async step goes.  HIP events on the encoder stream (around the encoder)
and on the decode stream (around the greedy chain) of every step, for the overlapped schedule and for the blocking call:

    python tools/stream_timeline.py [--steps 12] [--batch 64] [--max-length 20]

prints per step [encoder start, end | decode start, end] in ms relative to the first encoder start, and the means of the
encoder's and the chain's durations alone (blocking) and overlapped."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiocaption_amd as A
from audiocaption_amd import build, procedural as P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--max-length", type=int, default=20)
    args = ap.parse_args()
    build.build()
    vocab = 4368
    model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
    model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
    model = model.eval().cuda()
    L = int(32000 * args.seconds)
    wavs = [torch.from_numpy(P.synthetic_wav(args.batch, L, seed=s)).cuda() for s in range(4)]
    marks = []

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    enc_fwd, greedy = model.encoder.forward, model.decoder.greedy

    def enc_wrapped(d):
        a = ev()
        r = enc_fwd(d)
        marks.append(["enc", a, ev()])
        return r

    def greedy_wrapped(*a_, **k_):
        a = ev()
        r = greedy(*a_, **k_)
        marks.append(["dec", a, ev()])
        return r

    model.encoder.forward = enc_wrapped
    model.decoder.greedy = greedy_wrapped

    def req(i):
        return {"mode": "inference", "wav": wavs[i % 4], "wav_len": [L] * args.batch, "specaug": False,
                "sample_method": "greedy", "max_length": args.max_length}

    with torch.no_grad():
        for i in range(4):
            model.forward_async(req(i)).result()
        for sched in ("blocking", "overlapped"):
            torch.cuda.synchronize()
            marks.clear()
            if sched == "blocking":
                for i in range(args.steps):
                    model(req(i))
            else:
                pend = [model.forward_async(req(i)) for i in range(args.steps)]
                for p in pend:
                    p.result()
            torch.cuda.synchronize()
            base = marks[0][1]
            encs = [(base.elapsed_time(a), base.elapsed_time(b)) for k, a, b in marks if k == "enc"]
            decs = [(base.elapsed_time(a), base.elapsed_time(b)) for k, a, b in marks if k == "dec"]
            print(f"== {sched}: {args.steps} steps, batch {args.batch}, max_length {args.max_length}")
            for i, (e, d) in enumerate(zip(encs, decs)):
                print(f"  step {i:2d}  enc [{e[0]:8.2f} {e[1]:8.2f}] {e[1] - e[0]:6.2f} ms   dec [{d[0]:8.2f} {d[1]:8.2f}] {d[1] - d[0]:6.2f} ms")
            if not encs or not decs:   # the overlapped schedule groups chains: fewer decode marks than steps
                print(f"  {len(encs)} encoder and {len(decs)} decode marks recorded")
                continue
            inner = slice(2, -1) if min(len(encs), len(decs)) > 4 else slice(0, None)
            me = sum(b - a for a, b in encs[inner]) / len(encs[inner])
            md = sum(b - a for a, b in decs[inner]) / len(decs[inner])
            total = (max(decs[-1][1], encs[-1][1]) - encs[0][0]) / args.steps
            print(f"  mean encoder {me:.2f} ms, mean decode chain {md:.2f} ms, wall per step {total:.2f} ms")


if __name__ == "__main__":
    main()

"""Development: where do the workgroups of a CU-masked HIP stream run?  For a few (first_cu, n_cus) masks of
``ac_stream_create_cu_mask`` launch ``ac_placement_probe`` (every workgroup records its XCC_ID / HW_ID registers and stays
resident for a while, so the grid spreads over everything the mask allows) and print the distinct compute units per XCD;
also the time of the matrix-rate probe on the masked stream (it scales with 256 / n_cus when the mask is in effect), run
eagerly and replayed from a captured graph (the decode chain is a graph replay)."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from audiocaption_amd import _lib, build
    from audiocaption_amd.transformer_model import _masked_stream
    build.build()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    print("compute units:", total)
    blocks = 2048
    out = torch.zeros(blocks, 2, dtype=torch.int32, device=dev)
    acc = torch.zeros(1024 * 256, device=dev)
    for first, n in ((0, total), (0, 8), (0, 32), (0, 64), (64, total - 64), (8, 8), (1, 1)):
        st = _masked_stream(dev, first, n) if (first, n) != (0, total) else torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            out.zero_()
            _lib.check(lib.ac_placement_probe(out.data_ptr(), blocks, 256, 2000, _lib.stream()), "ac_placement_probe")
            st.synchronize()
            o = out.cpu().numpy()
            xcc = o[:, 0] & 0xf
            hw = o[:, 1]
            cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
            where = collections.defaultdict(set)
            for x, a, b, c in zip(xcc, se, sh, cu):
                where[int(x)].add((int(a), int(b), int(c)))
            n_used = sum(len(v) for v in where.values())
            # matrix-rate probe: 1024 workgroups x 200 iterations, eager
            def probe():
                _lib.check(lib.ac_mfma_bf16_probe(acc.data_ptr(), 1024, 200, _lib.stream()), "ac_mfma_bf16_probe")
            probe()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            probe()
            e1.record()
            st.synchronize()
            t_eager = e0.elapsed_time(e1)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _lib.check(lib.ac_mfma_bf16_probe(acc.data_ptr(), 1024, 200, _lib.stream()), "ac_mfma_bf16_probe")
        with torch.cuda.stream(st):
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            st.synchronize()
            t_graph = e0.elapsed_time(e1)
        print(f"mask bits [{first}, {first + n}): {n_used} distinct CUs; per XCD " +
              " ".join(f"{x}:{len(where[x])}" for x in sorted(where)) +
              f"; matrix probe {t_eager:.3f} ms eager, {t_graph:.3f} ms as a graph replay")
        if n <= 8:
            print("   ", {x: sorted(v) for x, v in where.items()})


if __name__ == "__main__":
    main()

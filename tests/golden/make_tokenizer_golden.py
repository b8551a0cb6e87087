"""Fixture generator (run in the build container, where /root/reference exists): the reference's DictTokenizer
(captioning/datasets/text_tokenizer.py:8-79) on a seeded vocabulary, captions and id matrices -> g12_tokenizer.json
(inputs and the reference's outputs only).  tests/test_train_oracle.py holds audiocaption_amd.text.DictTokenizer to it.

    python tests/golden/make_tokenizer_golden.py
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
for m in ("h5py", "toml", "wandb"):
    sys.modules.setdefault(m, types.ModuleType(m))
from captioning.datasets.text_tokenizer import DictTokenizer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(12)
corpus = ["a", "dog", "barks", "while", "rain", "falls", "on", "the", "roof", "birds", "sing", "loudly", "engine", "idles"]
tok = DictTokenizer(max_length=6)
for w in corpus:
    tok.add_word(w)
vocab = dict(tok.state_dict())
texts = [" ".join(rng.choice(corpus + ["zebra", "quietly"], size=int(n))) for n in rng.integers(1, 11, size=12)]
enc = tok(texts)
seqs = rng.integers(0, len(vocab), size=(24, 9))
seqs[:6, 0] = 1                      # some rows start with <start>
seqs[3:12, rng.integers(1, 9)] = 2   # an <end> somewhere
out = {"vocab": vocab, "max_length": 6, "texts": texts, "cap": enc["cap"].tolist(), "cap_len": np.asarray(enc["cap_len"]).tolist(),
       "seqs": seqs.tolist(), "decoded": tok.decode(seqs), "len": len(tok)}
with open(os.path.join(HERE, "g12_tokenizer.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote g12_tokenizer.json:", len(texts), "captions,", len(seqs), "id rows")

"""Token ids <-> text: the step right after the hot path.

Plugin-compatible with the reference's ``DictTokenizer`` (captioning/datasets/text_tokenizer.py:8-79) - same class and
method names, the same pickled ``word -> index`` dict as checkpoint format, ``<pad>`` 0, ``<start>`` 1, ``<end>`` 2, ``<unk>``
3 ahead of the corpus words - and with the prediction file the runners write (python_scripts/train_eval/base.py:212-224,
295-305).  Host code, written around what the accelerated path hands over: ONE (B, max_length) int64 CPU array per batch,
so ``decode`` works on the whole array (first end token per row found with array operations, one vocabulary lookup table)
and ``__call__`` fills one padded int64 matrix."""
import json
import os
import pickle

import numpy as np

SPECIALS = ("<pad>", "<start>", "<end>", "<unk>")


class DictTokenizer:

    def __init__(self, tokenizer_path=None, max_length=20):
        self.max_length = max_length
        self.loaded = False
        self._adopt({w: i for i, w in enumerate(SPECIALS)})
        if tokenizer_path is not None and os.path.exists(tokenizer_path):
            with open(tokenizer_path, "rb") as f:
                self.load_state_dict(pickle.load(f))
            self.loaded = True

    # ---- vocabulary ---------------------------------------------------------------------------------------------------
    def _adopt(self, word2idx):
        """Take a word -> index dict as THE vocabulary and derive what the other methods use from it."""
        self.word2idx = word2idx
        self.idx2word = {i: w for w, i in word2idx.items()}
        self.idx = len(word2idx)
        self._table = None                                         # index -> word array of ``decode``: built on first use
        self.bos, self.eos, self.pad = (word2idx[w] for w in ("<start>", "<end>", "<pad>"))
        self._unk = word2idx["<unk>"]

    def _lookup_table(self):
        """index -> word as an object array (None where the vocabulary has no word); rebuilt after the vocabulary changed."""
        if self._table is None:
            table = np.full(max(self.idx2word, default=-1) + 1, None, dtype=object)
            for i, w in self.idx2word.items():
                table[i] = w
            self._table = table
        return self._table

    def add_word(self, word):
        if word not in self.word2idx:          # O(1), like the reference (text_tokenizer.py:30-34)
            self.word2idx[word] = self.idx
            self.idx2word[self.idx] = word
            self.idx += 1
            self._table = None

    def encode_word(self, word):
        return self.word2idx.get(word, self._unk)

    def __len__(self):
        return len(self.word2idx)

    def state_dict(self):
        return self.word2idx

    def load_state_dict(self, state_dict):
        self._adopt(dict(state_dict))

    # ---- text -> ids (training captions) -------------------------------------------------------------------------------
    def __call__(self, texts):
        if not isinstance(texts, list):
            raise TypeError("DictTokenizer expects a list of caption strings")
        lookup, unk, keep = self.word2idx, self._unk, self.max_length
        ids = [[lookup.get(w, unk) for w in t.split()[:keep]] for t in texts]
        if not ids:
            return {"cap": np.zeros((0, 0), dtype=np.int64), "cap_len": np.zeros(0, dtype=np.int64)}
        lens = np.fromiter((len(r) + 2 for r in ids), dtype=np.int64, count=len(ids))
        caps = np.full((len(ids), int(lens.max())), self.pad, dtype=np.int64)
        caps[:, 0] = self.bos
        for r, (row, n) in enumerate(zip(ids, lens)):
            caps[r, 1:n - 1] = row
            caps[r, n - 1] = self.eos
        return {"cap": caps, "cap_len": lens}

    # ---- ids -> text (what the model returns) --------------------------------------------------------------------------
    def decode(self, batch_token_ids):
        ids = np.asarray(batch_token_ids)
        if ids.ndim == 1:
            ids = ids[None]
        ended = ids == self.eos
        stop = np.where(ended.any(axis=1), ended.argmax(axis=1), ids.shape[1])   # first end token of every row
        table = self._lookup_table()
        show = (np.arange(ids.shape[1])[None] < stop[:, None]) & (ids != self.bos)
        inside = (ids >= 0) & (ids < len(table))
        words = table[np.where(inside, ids, 0)]
        bad = show & (~inside | np.equal(words, None))
        if bad.any():                          # the reference's idx2word[token_id] raises for an id outside the vocabulary
            raise KeyError(int(ids[bad][0]))
        return [" ".join(row[m]) for row, m in zip(words, show)]


def write_predictions(key2pred, path):
    """The runners' prediction file: {"predictions": [{"filename": key, "tokens": caption}, ...]} (base.py:295-305)."""
    rows = []
    for key, pred in key2pred.items():
        rows.append({"filename": key, "tokens": pred[0] if isinstance(pred, (list, tuple)) else pred})
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"predictions": rows}, f, indent=4)

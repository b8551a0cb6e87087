"""Token ids <-> text.  Plugin-compatible with the reference's ``DictTokenizer``
(captioning/datasets/text_tokenizer.py:8-79: ``<pad>`` 0, ``<start>`` 1, ``<end>`` 2, ``<unk>`` 3, then the corpus words;
pickled word->index dict) and the prediction file the runners write (python_scripts/train_eval/base.py:212-224,
295-305).  Pure host code: it consumes the (B, max_length) int64 CPU tensor the models return."""
import json
import os
import pickle

import numpy as np


class DictTokenizer:

    def __init__(self, tokenizer_path=None, max_length=20):
        self.word2idx, self.idx2word, self.idx = {}, {}, 0
        for w in ("<pad>", "<start>", "<end>", "<unk>"):
            self.add_word(w)
        self.loaded = False
        if tokenizer_path is not None and os.path.exists(tokenizer_path):
            with open(tokenizer_path, "rb") as f:
                self.load_state_dict(pickle.load(f))
            self.loaded = True
        self.bos, self.eos, self.pad = self.word2idx["<start>"], self.word2idx["<end>"], self.word2idx["<pad>"]
        self.max_length = max_length

    def add_word(self, word):
        if word not in self.word2idx:
            self.word2idx[word] = self.idx
            self.idx2word[self.idx] = word
            self.idx += 1

    def encode_word(self, word):
        return self.word2idx.get(word, self.word2idx["<unk>"])

    def __call__(self, texts):
        assert isinstance(texts, list), "the input must be List[str]"
        rows = []
        for text in texts:
            tokens = [self.encode_word(t) for t in text.split()][:self.max_length]
            rows.append(np.array([self.bos] + tokens + [self.eos]))
        lens = np.array([len(r) for r in rows])
        caps = np.full((len(rows), int(lens.max())), self.pad, dtype=np.int64)
        for i, r in enumerate(rows):
            caps[i, :len(r)] = r
        return {"cap": caps, "cap_len": lens}

    def decode(self, batch_token_ids):
        out = []
        for ids in np.asarray(batch_token_ids):
            words = []
            for t in ids.tolist():
                if t == self.eos:
                    break
                if t == self.bos:
                    continue
                words.append(self.idx2word[t])
            out.append(" ".join(words))
        return out

    def __len__(self):
        return len(self.word2idx)

    def state_dict(self):
        return self.word2idx

    def load_state_dict(self, state_dict):
        self.word2idx = state_dict
        self.idx2word = {i: w for w, i in state_dict.items()}
        self.idx = len(state_dict)


def write_predictions(key2pred, path):
    """The runners' prediction file: {"predictions": [{"filename": key, "tokens": caption}, ...]} (base.py:295-305)."""
    data = [{"filename": k, "tokens": v[0] if isinstance(v, (list, tuple)) else v} for k, v in key2pred.items()]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"predictions": data}, f, indent=4)

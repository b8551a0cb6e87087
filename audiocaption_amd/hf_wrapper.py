"""``model(audio, audio_length)`` call surface of the reference's HF wrappers
(``Effb2TrmCaptioningModel.forward``, hf_wrapper.py:1162-1181): raw waveforms in, a CPU LongTensor of
token ids out, default ``sample_method="beam"``, ``beam_size=3``, ``max_length=20``.

The reference ships this surface only for its EffB2 encoder; here it wraps any model built from this
package (e.g. the Cnn14Rnn-Trm captioner), so ``demo.py``-style callers can switch without touching the
input_dict plumbing.
"""
import numpy as np
import torch
import torch.nn as nn


class CaptioningConfig:
    """Minimal stand-in for the HF config object: callers read ``model.config.sample_rate`` (README.md:35)."""

    def __init__(self, sample_rate=32000, vocab_size=4368, **kwargs):
        self.sample_rate = sample_rate
        self.vocab_size = vocab_size
        self.__dict__.update(kwargs)


class CaptioningModel(nn.Module):

    def __init__(self, model, config=None):
        super().__init__()
        self.model = model
        self.config = config or CaptioningConfig(vocab_size=model.vocab_size)

    @torch.no_grad()
    def forward(self, audio, audio_length, sample_method="beam", beam_size=3, max_length=20, temp=1.0):
        device = next(self.model.parameters()).device
        if not isinstance(audio, torch.Tensor):
            audio = torch.as_tensor(np.asarray(audio))
        input_dict = {
            "wav": audio.to(device), "wav_len": audio_length, "specaug": False, "mode": "inference",
            "sample_method": sample_method, "max_length": max_length, "temp": temp,
        }
        if sample_method == "beam":
            input_dict["beam_size"] = beam_size
        return self.model(input_dict)["seq"].cpu()

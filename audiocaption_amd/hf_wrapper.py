"""The reference's Hugging Face call surface on the MI355X path.

``Effb2TrmConfig`` / ``Effb2TrmCaptioningModel`` mirror ``captioning.models.hf_wrapper`` (hf_wrapper.py:1115-1181): the same
config keys and defaults, the same module tree -

    Effb2TrmCaptioningModel.model        ContraEncoderKdWrapper   (hf_wrapper.py:1071-1112)
        .model                           TransformerModel(EfficientNetB2(), TransformerDecoder(tie_weights=True))
        .stdnt_proj / .tchr_proj / .logit_scale     knowledge-distillation heads: in the checkpoint, unused at inference

- hence the same ``state_dict()`` keys as the published ``wsntxxn/effb2-trm-audio-captioning`` weights
(``model.model.encoder.backbone.eff_net._conv_stem.weight`` ..., ``model.stdnt_proj.weight``, ``model.logit_scale``), and
the same call: ``model(audio, audio_length, sample_method="beam", beam_size=3, max_length=20, temp=1.0)`` returning a CPU
LongTensor (B, max_length); ``model.config.sample_rate`` is what callers resample to (README.md:35).

With ``transformers`` importable the two classes ARE a ``PretrainedConfig`` / ``PreTrainedModel`` (``from_pretrained`` /
``save_pretrained`` work on a local directory); without it they fall back to plain classes with ``load_checkpoint``.

``CaptioningModel`` is the same call surface around ANY model of this package (the reference ships it only for its EffB2
model), so ``demo.py``-style callers can use the Cnn14Rnn-Trm captioner without the input_dict plumbing.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import merge_load_state_dict
from .effnet_encoder import EfficientNetB2
from .transformer_decoder import TransformerDecoder
from .transformer_model import CaptionMetaMixin, TransformerModel

try:   # the real base classes when the package is there (it is an optional dependency of the reference as well)
    from transformers import PretrainedConfig, PreTrainedModel
    HAVE_TRANSFORMERS = True
except Exception:   # noqa: BLE001
    HAVE_TRANSFORMERS = False

    class PretrainedConfig:   # minimal stand-ins: attribute bag + nn.Module
        def __init__(self, **kwargs):
            self.__dict__.update(kwargs)

    class PreTrainedModel(nn.Module):
        config_class = None

        def __init__(self, config):
            super().__init__()
            self.config = config

        @property
        def device(self):
            return next(self.parameters()).device

        def post_init(self):
            pass


def _input_dict(device, audio, audio_length, sample_method, beam_size, max_length, temp):
    if not isinstance(audio, torch.Tensor):
        audio = torch.as_tensor(np.asarray(audio))
    d = {"wav": audio.to(device), "wav_len": audio_length, "specaug": False, "mode": "inference",
         "sample_method": sample_method, "max_length": max_length, "temp": temp}
    if sample_method == "beam":
        d["beam_size"] = beam_size
    return d


class ContraEncoderKdWrapper(nn.Module, CaptionMetaMixin):
    """hf_wrapper.py:1071-1112.  Holds the captioner plus the contrastive knowledge-distillation heads that exist in the
    published state dict.  Inference passes straight through to the captioner (HIP path); with ``tchr_output`` in the input
    dict the symmetric contrastive loss between the projected clip embedding and the teacher's embedding is added as
    ``enc_kd_loss`` (hf_wrapper.py:1095-1111) - a few torch ops on ``fc_emb``, outside the accelerated path.

    The loss is HEAD-ONLY here: ``fc_emb`` comes out of the HIP encoder detached, so ``enc_kd_loss`` trains
    ``stdnt_proj`` / ``tchr_proj`` / ``logit_scale`` and sends no gradient into the encoder.  Distilling INTO the encoder
    (the reference's recipe: mode "train", encoder_output_dict merged into the training output, base.py:133) needs the
    encoder's backward, which this path does not build (SURVEY section 8 scope): mode "train" with ``tchr_output`` raises
    ``NotImplementedError`` instead of returning a loss that silently trains nothing but the heads."""

    def __init__(self, model, shared_dim, tchr_dim):
        super().__init__()
        self.model = model
        self.tchr_dim = tchr_dim
        fc = model.encoder.fc_emb_size if hasattr(model, "encoder") else model.fc_emb_size
        self.stdnt_proj = nn.Linear(fc, shared_dim)
        self.tchr_proj = nn.Linear(tchr_dim, shared_dim)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def forward(self, input_dict):
        if "tchr_output" in input_dict and input_dict.get("mode") == "train" and not input_dict.get("unsup", False):
            raise NotImplementedError(
                "ContraEncoderKdWrapper: encoder knowledge distillation in mode='train' needs the gradient of enc_kd_loss "
                "with respect to the encoder (hf_wrapper.py:1095-1111 on top of base.py:133); the accelerated training "
                "step returns no fc_emb and the HIP encoder has no backward.  Use mode='inference' (or unsup=True) for a "
                "head-only loss")
        out = self.model.encoder(input_dict) if input_dict.get("unsup", False) else self.model(input_dict)
        if "tchr_output" in input_dict:
            # CLIP-style loss: cosine similarities of every (student clip, teacher clip) pair, scaled, matched on the diagonal
            student = F.normalize(self.stdnt_proj(out["fc_emb"]), dim=-1)
            teacher = F.normalize(self.tchr_proj(input_dict["tchr_output"]["embedding"]), dim=-1)
            sim = self.logit_scale * (student @ teacher.t())
            target = torch.arange(sim.shape[0], device=sim.device)
            out["enc_kd_loss"] = 0.5 * (F.cross_entropy(sim, target) + F.cross_entropy(sim.t(), target))
        return out


class Effb2TrmConfig(PretrainedConfig):
    """hf_wrapper.py:1115-1141: same keys, same defaults."""
    model_type = "effb2_trm_captioning"

    def __init__(self, sample_rate=16000, tchr_dim=768, shared_dim=1024, fc_emb_dim=1408, attn_emb_dim=1408,
                 decoder_n_layers=2, decoder_we_tie_weights=True, decoder_emb_dim=256, decoder_dropout=0.2,
                 vocab_size=4981, **kwargs):
        self.sample_rate = sample_rate
        self.tchr_dim = tchr_dim
        self.shared_dim = shared_dim
        self.fc_emb_dim = fc_emb_dim
        self.attn_emb_dim = attn_emb_dim
        self.decoder_n_layers = decoder_n_layers
        self.decoder_we_tie_weights = decoder_we_tie_weights
        self.decoder_emb_dim = decoder_emb_dim
        self.decoder_dropout = decoder_dropout
        self.vocab_size = vocab_size
        super().__init__(**kwargs)


class Effb2TrmCaptioningModel(PreTrainedModel):
    """hf_wrapper.py:1144-1181."""
    config_class = Effb2TrmConfig
    base_model_prefix = "model"
    main_input_name = "audio"
    # decoder.classifier.weight IS decoder.word_embedding.weight when decoder_we_tie_weights (transformer_decoder.py:36-37)
    _tied_weights_keys = {"model.model.decoder.classifier.weight": "model.model.decoder.word_embedding.weight"}
    _keys_to_ignore_on_load_missing = [r"melspec_extractor\."]

    def __init__(self, config):
        super().__init__(config)
        encoder = EfficientNetB2()
        decoder = TransformerDecoder(emb_dim=config.decoder_emb_dim, vocab_size=config.vocab_size,
                                     fc_emb_dim=config.fc_emb_dim, attn_emb_dim=config.attn_emb_dim,
                                     dropout=config.decoder_dropout, nlayers=config.decoder_n_layers,
                                     tie_weights=config.decoder_we_tie_weights)
        model = TransformerModel(encoder, decoder)
        self.model = ContraEncoderKdWrapper(model, config.shared_dim, config.tchr_dim)
        if not config.decoder_we_tie_weights:
            self._tied_weights_keys = {}
        self.post_init()

    def tie_weights(self, *args, **kwargs):
        """Called by the HF loaders after the weights are in place (models are built on the meta device there, which
        drops the tie made at construction): the classifier shares the word-embedding matrix again."""
        try:
            super().tie_weights(*args, **kwargs)
        except Exception:   # noqa: BLE001 - the stand-in base class has none; the HF one may reject our key style
            pass
        if getattr(self.config, "decoder_we_tie_weights", False) and hasattr(self, "model"):
            dec = self.model.model.decoder
            dec.classifier.weight = dec.word_embedding.weight

    def _init_weights(self, module):
        """The sub-modules initialise themselves at construction (as the reference's do); nothing is re-drawn here."""

    def load_checkpoint(self, state_dict, strict=True, output_fn=lambda s: None):
        """Load a state dict in the PUBLISHED layout (keys ``model.model.encoder...``, ``model.model.decoder...``,
        ``model.stdnt_proj...``, ``model.tchr_proj...``, ``model.logit_scale``; hf_wrapper.py:1071-1160) or in the layout of
        a bare captioner (``encoder...`` / ``decoder...``: what the reference's trainer saves, run.py:209-216).  ``strict``:
        every key of the module must be present with the right shape (the two torchaudio mel buffers may be absent);
        otherwise the tolerant shape-filtered merge of train_util.py:188-202."""
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location="cpu")
        if "model" in state_dict and isinstance(state_dict["model"], dict):
            state_dict = state_dict["model"]
        if not any(k.startswith("model.") for k in state_dict):      # bare captioner -> wrapped names
            state_dict = {"model.model." + k: v for k, v in state_dict.items()}
            own = self.state_dict()
            for k in ("model.stdnt_proj.weight", "model.stdnt_proj.bias", "model.tchr_proj.weight", "model.tchr_proj.bias",
                      "model.logit_scale"):
                state_dict.setdefault(k, own[k])     # the distillation heads are not part of a bare captioner
        if strict:
            return self.load_state_dict(state_dict, strict=True)
        return merge_load_state_dict(state_dict, self, output_fn)

    @torch.no_grad()
    def forward(self, audio, audio_length, sample_method="beam", beam_size=3, max_length=20, temp=1.0):
        return self.model(_input_dict(self.device, audio, audio_length, sample_method, beam_size, max_length,
                                      temp))["seq"].cpu()


class CaptioningConfig:
    """Minimal stand-in for the HF config object: callers read ``model.config.sample_rate`` (README.md:35)."""

    def __init__(self, sample_rate=32000, vocab_size=4368, **kwargs):
        self.sample_rate = sample_rate
        self.vocab_size = vocab_size
        self.__dict__.update(kwargs)


class CaptioningModel(nn.Module):
    """``model(audio, audio_length)`` around any captioner of this package."""

    def __init__(self, model, config=None):
        super().__init__()
        self.model = model
        self.config = config or CaptioningConfig(vocab_size=model.vocab_size)

    @torch.no_grad()
    def forward(self, audio, audio_length, sample_method="beam", beam_size=3, max_length=20, temp=1.0):
        device = next(self.model.parameters()).device
        return self.model(_input_dict(device, audio, audio_length, sample_method, beam_size, max_length, temp))["seq"].cpu()

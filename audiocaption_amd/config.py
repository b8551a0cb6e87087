"""Plugin construction from the reference's YAML model section.

Same recursion as ``captioning.utils.train_util.init_model_from_config`` / ``init_obj_from_dict``
(train_util.py:63-94): every sub-dict that is not ``type``/``args``/``pretrained`` is built first and
passed as a keyword named after its key; ``type`` is a dotted class path.  Dotted paths of the
reference's hot-path classes are resolved to this package's MI355X implementations, so the
reference's YAML files work unchanged (including the stale ``Cnn14RnnEncoder`` name).
"""
import importlib
import os

import torch

ALIASES = {
    "captioning.models.cnn_encoder.Cnn14Encoder": "audiocaption_amd.cnn_encoder.Cnn14Encoder",
    "captioning.models.rnn_encoder.RnnEncoder": "audiocaption_amd.rnn_encoder.RnnEncoder",
    "captioning.models.crnn_trm_encoder.CrnnEncoder": "audiocaption_amd.crnn_trm_encoder.CrnnEncoder",
    "captioning.models.crnn_trm_encoder.Cnn14RnnEncoder": "audiocaption_amd.crnn_trm_encoder.Cnn14RnnEncoder",
    "captioning.models.transformer_decoder.TransformerDecoder": "audiocaption_amd.transformer_decoder.TransformerDecoder",
    "captioning.models.transformer_model.TransformerModel": "audiocaption_amd.transformer_model.TransformerModel",
    "captioning.models.cnn_encoder.EfficientNetB2": "audiocaption_amd.effnet_encoder.EfficientNetB2",
    "captioning.models.transformer_encoder.TransformerEncoder": "audiocaption_amd.transformer_encoder.TransformerEncoder",
    "captioning.models.crnn_trm_encoder.Cnn14TransformerEncoder": "audiocaption_amd.crnn_trm_encoder.Cnn14TransformerEncoder",
    "captioning.losses.loss.LabelSmoothingLoss": "audiocaption_amd.loss.LabelSmoothingLoss",
    "captioning.utils.lr_scheduler.ExponentialDecayScheduler": "audiocaption_amd.lr_scheduler.ExponentialDecayScheduler",
}


def get_cls_from_str(string):
    string = ALIASES.get(string, string)
    module_name, cls_name = string.rsplit(".", 1)
    return getattr(importlib.import_module(module_name), cls_name)


def init_obj_from_dict(config, **kwargs):
    obj_args = dict(config.get("args", {}))
    obj_args.update(kwargs)
    for k in config:
        if k not in ("type", "args") and isinstance(config[k], dict) and k not in kwargs:
            obj_args[k] = init_obj_from_dict(config[k])
    return get_cls_from_str(config["type"])(**obj_args)


def load_pretrained_model(model, pretrained, output_fn=print):
    """train_util.py:204-223: missing file is non-fatal; module hook first; tolerant key/shape merge."""
    if not isinstance(pretrained, dict) and not os.path.exists(pretrained):
        output_fn(f"pretrained {pretrained} not exist!")
        return
    if hasattr(model, "load_pretrained"):
        model.load_pretrained(pretrained, output_fn)
        return
    state_dict = pretrained if isinstance(pretrained, dict) else torch.load(pretrained, map_location="cpu")
    if "model" in state_dict:
        state_dict = state_dict["model"]
    merge_load_state_dict(state_dict, model, output_fn)


def merge_load_state_dict(state_dict, model, output_fn=print):
    """train_util.py:188-202: load the keys that exist with the same shape, report the rest."""
    own = model.state_dict()
    good = {k: v for k, v in state_dict.items() if k in own and own[k].shape == v.shape}
    output_fn(f"Loading pre-trained model, with mismatched keys {[k for k in state_dict if k not in good]}\n")
    own.update(good)
    model.load_state_dict(own, strict=True)
    return good.keys()


def init_model_from_config(config, print_fn=print):
    kwargs = {}
    for k in config:
        if k not in ("type", "args", "pretrained"):
            sub_model = init_model_from_config(config[k], print_fn)
            if "pretrained" in config[k]:
                load_pretrained_model(sub_model, config[k]["pretrained"], print_fn)
            kwargs[k] = sub_model
    return init_obj_from_dict(config, **kwargs)


def cnn14rnn_trm_config(vocab_size=4368, encoder_name="CrnnEncoder"):
    """The ``model:`` section of eg_configs/{clotho_v2,audiocaps}/waveform/cnn14rnn_trm.yaml (lines 7-38)."""
    return {
        "encoder": {
            "type": f"captioning.models.crnn_trm_encoder.{encoder_name}",
            "args": {"freeze_cnn": True, "freeze_cnn_bn": True},
            "cnn": {"type": "captioning.models.cnn_encoder.Cnn14Encoder", "args": {"sample_rate": 32000}},
            "rnn": {"type": "captioning.models.rnn_encoder.RnnEncoder",
                    "args": {"bidirectional": True, "hidden_size": 256, "dropout": 0.5, "num_layers": 3,
                             "spec_dim": -1, "fc_feat_dim": 2048, "attn_feat_dim": 2048}},
        },
        "decoder": {"type": "captioning.models.transformer_decoder.TransformerDecoder",
                    "args": {"vocab_size": vocab_size, "emb_dim": 256, "fc_emb_dim": 512, "attn_emb_dim": 512,
                             "nlayers": 2, "dropout": 0.2}},
        "type": "captioning.models.transformer_model.TransformerModel",
        "args": {},
    }


def effb2_trm_config(vocab_size=4981):
    """The model ``Effb2TrmCaptioningModel`` builds from ``Effb2TrmConfig`` defaults (hf_wrapper.py:1115-1160):
    EfficientNetB2 encoder (16 kHz), 2-layer decoder with the word embedding tied to the classifier."""
    return {
        "encoder": {"type": "captioning.models.cnn_encoder.EfficientNetB2", "args": {}},
        "decoder": {"type": "captioning.models.transformer_decoder.TransformerDecoder",
                    "args": {"vocab_size": vocab_size, "emb_dim": 256, "fc_emb_dim": 1408, "attn_emb_dim": 1408,
                             "nlayers": 2, "dropout": 0.2, "tie_weights": True}},
        "type": "captioning.models.transformer_model.TransformerModel",
        "args": {},
    }

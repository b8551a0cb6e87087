"""Make the reference's dotted class paths import THIS package's implementations.

``install()`` registers ``captioning.models.{cnn_encoder, rnn_encoder, crnn_trm_encoder,
transformer_decoder, transformer_model}`` in ``sys.modules`` so that the reference's own
``train_util.init_model_from_config`` (train_util.py:63-94), ``run.py``, ``inference.py`` and ``demo.py``
build the MI355X classes from unchanged YAML files.  If the reference package is importable its other
modules (datasets, losses, utils) stay the reference's; only the five hot-path modules are replaced.
See INTEGRATION.md.
"""
import importlib
import sys
import types

HOT_MODULES = {
    "captioning.models.cnn_encoder": "audiocaption_amd.cnn_encoder",
    "captioning.models.rnn_encoder": "audiocaption_amd.rnn_encoder",
    "captioning.models.crnn_trm_encoder": "audiocaption_amd.crnn_trm_encoder",
    "captioning.models.transformer_decoder": "audiocaption_amd.transformer_decoder",
    "captioning.models.transformer_model": "audiocaption_amd.transformer_model",
}


def install():
    for pkg in ("captioning", "captioning.models"):
        if pkg not in sys.modules:
            try:
                importlib.import_module(pkg)
            except Exception:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
    for ref_name, own_name in HOT_MODULES.items():
        mod = importlib.import_module(own_name)
        sys.modules[ref_name] = mod
        setattr(sys.modules["captioning.models"], ref_name.rsplit(".", 1)[1], mod)
    sys.modules["captioning"].models = sys.modules["captioning.models"]

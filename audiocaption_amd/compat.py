"""Make the reference's dotted class paths import THIS package's implementations.

``install()`` registers ``captioning.models.{cnn_encoder, rnn_encoder, crnn_trm_encoder,
transformer_decoder, transformer_model, transformer_encoder, hf_wrapper}`` in ``sys.modules`` so that the reference's own
``train_util.init_model_from_config`` (train_util.py:63-94), ``run.py``, ``inference.py`` and ``demo.py``
build the MI355X classes from unchanged YAML files.  If the reference package is importable its other
modules (datasets, losses, utils) stay the reference's; only the hot-path modules are replaced.
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
    "captioning.models.transformer_encoder": "audiocaption_amd.transformer_encoder",
    # Effb2TrmConfig / Effb2TrmCaptioningModel / ContraEncoderKdWrapper (hf_wrapper.py:1071-1181)
    "captioning.models.hf_wrapper": "audiocaption_amd.hf_wrapper",
}
# single classes patched into (or stubbed for) modules that also hold things outside the path
HOT_CLASSES = {
    "captioning.losses.loss": ("audiocaption_amd.loss", ["LabelSmoothingLoss"]),
    "captioning.utils.lr_scheduler": ("audiocaption_amd.lr_scheduler", ["ExponentialDecayScheduler"]),
}


def install(training=True):
    """``training``: also route ``LabelSmoothingLoss`` and ``ExponentialDecayScheduler`` (the reference's own scheduler
    cannot be constructed on torch >= 2.2, lr_scheduler.py:16) to this package."""
    for pkg in ("captioning", "captioning.models", "captioning.losses", "captioning.utils"):
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
    if training:
        for ref_name, (own_name, classes) in HOT_CLASSES.items():
            own = importlib.import_module(own_name)
            try:
                mod = importlib.import_module(ref_name)
            except Exception:
                mod = types.ModuleType(ref_name)
                sys.modules[ref_name] = mod
                setattr(sys.modules[ref_name.rsplit(".", 1)[0]], ref_name.rsplit(".", 1)[1], mod)
            for c in classes:
                setattr(mod, c, getattr(own, c))

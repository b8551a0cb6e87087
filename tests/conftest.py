import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def state4981():
    """Procedural weights at the AudioCaps vocabulary (the goldens were made with these)."""
    from audiocaption_amd import procedural as P
    return P.to_torch(P.cnn14rnn_trm_state(4981))


@pytest.fixture(scope="session")
def state_effb2():
    """Procedural weights of the EffB2-Transformer captioner."""
    from audiocaption_amd import procedural as P
    return P.to_torch(P.effb2_trm_state(4981))


@pytest.fixture(scope="session")
def hip_model(state4981):
    """The product model on cuda:0 with procedural weights (GPU tests only)."""
    import torch
    import audiocaption_amd as A
    from audiocaption_amd import build
    build.build()
    model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
    model.load_state_dict(state4981, strict=True)
    return model.eval().to("cuda:0")


@pytest.fixture(scope="session")
def diverse_models(state4981):
    """The product model with the two HIGH-ENTROPY decoder draws of the g4b / g5b fixtures (procedural.DIVERSE)."""
    import audiocaption_amd as A
    from audiocaption_amd import procedural as P
    models = {}
    for kind in ("greedy", "beam"):
        st = dict(state4981)
        st.update(P.to_torch(P.decoder_state_diverse(kind, vocab_size=4981)))
        m = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
        m.load_state_dict(st, strict=True)
        models[kind] = m.eval().to("cuda:0")
    return models

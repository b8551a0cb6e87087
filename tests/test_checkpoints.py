"""CPU tests of the checkpoint side of the boundary: every checkpoint layout the
reference's loaders accept is fabricated here (procedural weights, the reference's key names) and loaded through this
package's mirror of those loaders; and the reference's OWN plugin loader builds this package's classes from the
reference's unchanged YAML files after ``compat.install()`` (container only: /root/reference does not travel)."""
import os
import subprocess
import sys

import pytest
import torch

import audiocaption_amd as A
from audiocaption_amd import config, procedural as P

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


# ---------------------------------------------------------------------------------------------------------------------
# A17: Effb2TrmConfig / Effb2TrmCaptioningModel and the published state-dict layout
# ---------------------------------------------------------------------------------------------------------------------
def _published_effb2_checkpoint(state_effb2, seed=0):
    """A state dict in the layout of the published ``wsntxxn/effb2-trm-audio-captioning`` weights: the module tree of
    hf_wrapper.py:1144-1160 - Effb2TrmCaptioningModel.model = ContraEncoderKdWrapper(model = TransformerModel(encoder,
    decoder)) - plus the distillation heads of hf_wrapper.py:1079-1087 and torchaudio's two MelSpectrogram buffers
    (hf_wrapper.py:270-277: n_fft 512 -> window (512,), fb (257, 64))."""
    g = torch.Generator().manual_seed(seed)
    ck = {"model.model." + k: v.clone() for k, v in state_effb2.items()}
    ck["model.model.encoder.melspec_extractor.spectrogram.window"] = torch.hann_window(512, periodic=True)
    ck["model.model.encoder.melspec_extractor.mel_scale.fb"] = torch.rand(257, 64, generator=g)
    ck["model.stdnt_proj.weight"] = torch.randn(1024, 1408, generator=g)
    ck["model.stdnt_proj.bias"] = torch.randn(1024, generator=g)
    ck["model.tchr_proj.weight"] = torch.randn(1024, 768, generator=g)
    ck["model.tchr_proj.bias"] = torch.randn(1024, generator=g)
    ck["model.logit_scale"] = torch.tensor(2.5)
    return ck


def test_effb2_hf_config_defaults_and_module_tree(state_effb2):
    from audiocaption_amd.hf_wrapper import ContraEncoderKdWrapper, Effb2TrmCaptioningModel, Effb2TrmConfig
    cfg = Effb2TrmConfig()
    want = dict(sample_rate=16000, tchr_dim=768, shared_dim=1024, fc_emb_dim=1408, attn_emb_dim=1408, decoder_n_layers=2,
                decoder_we_tie_weights=True, decoder_emb_dim=256, decoder_dropout=0.2, vocab_size=4981)
    for k, v in want.items():                                    # hf_wrapper.py:1117-1140
        assert getattr(cfg, k) == v, k
    model = Effb2TrmCaptioningModel(cfg)
    assert model.config.sample_rate == 16000                     # README.md:35 reads it
    assert isinstance(model.model, ContraEncoderKdWrapper) and isinstance(model.model.model, A.TransformerModel)
    assert isinstance(model.model.model.encoder, A.EfficientNetB2)
    dec = model.model.model.decoder
    assert dec.classifier.weight is dec.word_embedding.weight    # decoder_we_tie_weights
    ck = _published_effb2_checkpoint(state_effb2)
    assert set(model.state_dict()) == set(ck)
    for k, v in model.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            assert tuple(v.shape) == tuple(ck[k].shape), k


def test_effb2_hf_model_loads_the_published_layout_strictly(state_effb2):
    from audiocaption_amd.hf_wrapper import Effb2TrmCaptioningModel, Effb2TrmConfig
    model = Effb2TrmCaptioningModel(Effb2TrmConfig())
    ck = _published_effb2_checkpoint(state_effb2)
    model.load_checkpoint(ck, strict=True)
    sd = model.state_dict()
    for k in ("model.model.encoder.backbone.eff_net._conv_stem.weight", "model.model.decoder.word_embedding.weight",
              "model.model.encoder.melspec_extractor.mel_scale.fb", "model.stdnt_proj.weight", "model.logit_scale"):
        assert torch.equal(sd[k], ck[k]), k
    # a file without the two torchaudio buffers (saved before they were persistent) still loads strictly
    ck2 = {k: v for k, v in ck.items() if "melspec_extractor" not in k}
    Effb2TrmCaptioningModel(Effb2TrmConfig()).load_checkpoint(ck2, strict=True)
    # a bare captioner state dict (what the trainer saves: encoder.* / decoder.*) goes under model.model.
    bare = Effb2TrmCaptioningModel(Effb2TrmConfig())
    bare.load_checkpoint({k: v for k, v in state_effb2.items()}, strict=True)
    assert torch.equal(bare.state_dict()["model.model.decoder.attn_proj.0.weight"], state_effb2["decoder.attn_proj.0.weight"])
    # a missing backbone tensor is an error under strict, a report under the tolerant merge (train_util.py:188-202)
    broken = {k: v for k, v in ck.items() if not k.endswith("_conv_head.weight")}
    with pytest.raises(RuntimeError):
        Effb2TrmCaptioningModel(Effb2TrmConfig()).load_checkpoint(broken, strict=True)
    said = []
    other_vocab = dict(ck)
    other_vocab["model.model.decoder.word_embedding.weight"] = torch.zeros(100, 256)
    other_vocab["model.model.decoder.classifier.weight"] = torch.zeros(100, 256)
    Effb2TrmCaptioningModel(Effb2TrmConfig()).load_checkpoint(other_vocab, strict=False, output_fn=said.append)
    assert "model.model.decoder.word_embedding.weight" in said[0]


def test_effb2_hf_save_and_from_pretrained_round_trip(state_effb2, tmp_path):
    from audiocaption_amd import hf_wrapper as H
    if not H.HAVE_TRANSFORMERS:
        pytest.skip("transformers not importable")
    model = H.Effb2TrmCaptioningModel(H.Effb2TrmConfig(vocab_size=4981))
    model.load_checkpoint(_published_effb2_checkpoint(state_effb2, seed=3), strict=True)
    model.save_pretrained(str(tmp_path))
    again = H.Effb2TrmCaptioningModel.from_pretrained(str(tmp_path))
    assert again.config.sample_rate == 16000 and again.config.vocab_size == 4981
    a, b = model.state_dict(), again.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    dec = again.model.model.decoder
    assert dec.classifier.weight.data_ptr() == dec.word_embedding.weight.data_ptr()   # still tied after loading


def test_compat_aliases_the_hf_wrapper_module():
    import importlib
    from audiocaption_amd import compat, hf_wrapper
    saved = {k: v for k, v in sys.modules.items() if k.startswith("captioning")}
    try:
        compat.install()
        mod = importlib.import_module("captioning.models.hf_wrapper")
        assert mod.Effb2TrmCaptioningModel is hf_wrapper.Effb2TrmCaptioningModel
        assert mod.Effb2TrmConfig is hf_wrapper.Effb2TrmConfig
    finally:
        for k in [k for k in sys.modules if k.startswith("captioning")]:
            del sys.modules[k]
        sys.modules.update(saved)


# ---------------------------------------------------------------------------------------------------------------------
# (f)2: Cnn14 pre-training checkpoints (PANNs / COLA / BLAT), trainer checkpoints, decoder checkpoints
# ---------------------------------------------------------------------------------------------------------------------
def _cnn_state(state4981):
    return {k[len("encoder.cnn."):]: v.clone() for k, v in state4981.items() if k.startswith("encoder.cnn.")}


def _fresh_cnn(freeze=False):
    torch.manual_seed(123)
    return A.Cnn14Encoder(32000, freeze=freeze)


def _assert_cnn_loaded(cnn, want, skipped=()):
    sd = cnn.state_dict()
    for k, v in want.items():
        if k.endswith("num_batches_tracked"):      # a counter; the procedural state stores it as shape (1,)
            continue
        if k in skipped:
            assert not torch.equal(sd[k], v), k
        else:
            assert torch.equal(sd[k], v), k


def test_panns_checkpoint_layout(tmp_path, state4981):
    """PANNs ``Cnn14_mAP=0.431.pth``: {"model": {...}} with the same block names plus the AudioSet head and the
    torchlibrosa front-end (cnn_encoder.py:376-393 "PANNs" branch; mismatched keys are reported and skipped)."""
    want = _cnn_state(state4981)
    ck = {"model": dict(want)}
    ck["model"]["fc_audioset.weight"] = torch.zeros(527, 2048)
    ck["model"]["fc_audioset.bias"] = torch.zeros(527)
    ck["model"]["spectrogram_extractor.stft.conv_real.weight"] = torch.zeros(513, 1, 1024)
    ck["model"]["logmel_extractor.melW"] = torch.zeros(513, 64)
    path = str(tmp_path / "Cnn14_mAP=0.431.pth")
    torch.save(ck, path)
    said = []
    cnn = _fresh_cnn(freeze=True)
    config.load_pretrained_model(cnn, path, said.append)           # train_util.py:204-223 -> the module's own hook
    _assert_cnn_loaded(cnn, want)
    assert "fc_audioset.weight" in said[0] and "logmel_extractor.melW" in said[0] and "conv_block1.conv1.weight" not in said[0]
    # freeze=True: what was loaded is frozen, what was not stays trainable (cnn_encoder.py:406-412)
    assert all(not p.requires_grad for n, p in cnn.named_parameters())
    part = {"model": {k: v for k, v in want.items() if not k.startswith("fc1.")}}
    torch.save(part, path)
    cnn = _fresh_cnn(freeze=True)
    config.load_pretrained_model(cnn, path, said.append)
    assert cnn.fc1.weight.requires_grad and cnn.fc1.bias.requires_grad
    assert not cnn.conv_block3.conv1.weight.requires_grad and not cnn.bn0.weight.requires_grad
    # freeze=False leaves requires_grad alone
    cnn = _fresh_cnn(freeze=False)
    config.load_pretrained_model(cnn, path, said.append)
    assert cnn.conv_block3.conv1.weight.requires_grad


def test_cola_and_blat_checkpoint_layouts(tmp_path, state4981):
    want = _cnn_state(state4981)
    cola = {"model": {"backbone." + k: v for k, v in want.items()}}          # cnn_encoder.py:386-391
    cola["model"]["head.projection.weight"] = torch.zeros(512, 2048)
    p1 = str(tmp_path / "cola.pth")
    torch.save(cola, p1)
    cnn = _fresh_cnn()
    config.load_pretrained_model(cnn, p1, lambda s: None)
    _assert_cnn_loaded(cnn, want)
    blat = {"state_dict": {"audio_encoder." + k: v for k, v in want.items()}}  # cnn_encoder.py:394-401
    blat["state_dict"]["text_encoder.embeddings.word_embeddings.weight"] = torch.zeros(10, 8)
    blat["state_dict"]["logit_scale"] = torch.tensor(1.0)
    p2 = str(tmp_path / "contrastive_pretrain_cnn14_bertm.pth")
    torch.save(blat, p2)
    cnn = _fresh_cnn()
    config.load_pretrained_model(cnn, p2, lambda s: None)
    _assert_cnn_loaded(cnn, want)
    p3 = str(tmp_path / "unknown.pth")
    torch.save({"weights": want}, p3)
    with pytest.raises(Exception, match="Unkown checkpoint format"):       # the reference's own message, cnn_encoder.py:403
        config.load_pretrained_model(_fresh_cnn(), p3, lambda s: None)
    said = []
    config.load_pretrained_model(_fresh_cnn(), str(tmp_path / "missing.pth"), said.append)   # non-fatal, train_util.py:207-209
    assert "not exist" in said[0]


def test_shape_mismatches_are_skipped_not_fatal(state4981):
    """train_util.py:188-202: keys with another shape (here: a 16 kHz checkpoint's nothing, a different vocabulary)."""
    model = A.init_model_from_config(A.cnn14rnn_trm_config(4368), print_fn=lambda s: None)
    before = model.decoder.word_embedding.weight.detach().clone()
    said = []
    config.load_pretrained_model(model, dict(state4981), said.append)      # vocabulary 4981 -> 4368 model
    assert "decoder.word_embedding.weight" in said[0] and "decoder.classifier.weight" in said[0]
    assert torch.equal(model.decoder.word_embedding.weight, before)
    assert torch.equal(model.encoder.rnn.network.weight_hh_l1, state4981["encoder.rnn.network.weight_hh_l1"])


def test_trainer_checkpoint_holds_trainable_parameters_and_buffers_only(tmp_path, state4981):
    """run.py:209-216 / base.py:231-244: ``best.pth`` / ``swa.pth`` carry the TRAINABLE parameters plus every buffer - the
    frozen Cnn14's conv / BN affine weights are not in it (they come from the PANNs file of the YAML) - next to
    ``epoch``, ``tokenizer`` ...; ``resume_checkpoint`` hands the whole dict to load_pretrained_model (base.py:247-248)."""
    model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
    model.load_state_dict(state4981, strict=True)
    saving_keys = [n for n, p in model.named_parameters() if p.requires_grad] + [n for n, _ in model.named_buffers()]
    sd = model.state_dict()
    ckpt = {"model": {k: sd[k].clone() for k in saving_keys}, "epoch": 7, "not_improve_cnt": 0,
            "tokenizer": {"word2idx": {"<pad>": 0}, "idx2word": {0: "<pad>"}}}
    assert "encoder.cnn.conv_block2.conv1.weight" not in ckpt["model"]          # frozen: not saved
    assert "encoder.cnn.conv_block2.bn1.running_mean" in ckpt["model"]          # buffer: saved
    assert "encoder.cnn.melspec_extractor.mel_scale.fb" in ckpt["model"]        # torchaudio buffer: saved as well
    assert "decoder.pos_encoder.pe" not in ckpt["model"]                        # frozen Parameter (model_util.py:181)
    path = str(tmp_path / "swa.pth")
    torch.save(ckpt, path)
    torch.manual_seed(5)
    fresh = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
    frozen_before = fresh.encoder.cnn.conv_block2.conv1.weight.detach().clone()
    config.load_pretrained_model(fresh, torch.load(path, map_location="cpu"), lambda s: None)
    got = fresh.state_dict()
    for k in saving_keys:
        assert torch.equal(got[k], sd[k]), k
    assert torch.equal(fresh.encoder.cnn.conv_block2.conv1.weight, frozen_before)   # untouched by this file
    # ... and the encoder's own weights then come from the PANNs-layout file named in the YAML (cnn14rnn_trm.yaml:17)
    panns = str(tmp_path / "Cnn14_mAP=0.431.pth")
    torch.save({"model": _cnn_state(state4981)}, panns)
    cfg = A.cnn14rnn_trm_config(4981)
    cfg["encoder"]["cnn"]["pretrained"] = panns
    built = A.init_model_from_config(cfg, print_fn=lambda s: None)
    assert torch.equal(built.encoder.cnn.conv_block2.conv1.weight, state4981["encoder.cnn.conv_block2.conv1.weight"])


def test_decoder_load_pretrained_takes_the_decoder_entries(tmp_path, state4981):
    """transformer_decoder.py:56-72: a whole-model checkpoint, ``decoder.`` prefix stripped."""
    path = str(tmp_path / "model.pth")
    torch.save({"model": {k: v for k, v in state4981.items() if k.startswith("decoder.")}}, path)
    torch.manual_seed(9)
    dec = A.TransformerDecoder(emb_dim=256, vocab_size=4981, fc_emb_dim=512, attn_emb_dim=512, dropout=0.2, nlayers=2,
                               freeze=True)
    config.load_pretrained_model(dec, path, lambda s: None)
    assert torch.equal(dec.model.layers[1].linear1.weight, state4981["decoder.model.layers.1.linear1.weight"])
    assert not any(p.requires_grad for p in dec.parameters())               # freeze=True + everything loaded


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own plugin loader on its unchanged YAML files (container only)
# ---------------------------------------------------------------------------------------------------------------------
_REF_LOADER = r'''
import sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, {repo!r}); sys.path.insert(0, {ref!r})
for name, attrs in (("toml", dict(loads=lambda s: {{}}, load=lambda f: {{}})), ("h5py", {{}}), ("wandb", dict(run=None))):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m     # non-arithmetic imports of train_util / loss
from audiocaption_amd import compat
compat.install()
from captioning.utils import train_util                                            # the REFERENCE's loader
import audiocaption_amd as A
said = []
for yaml_path, vocab, total in (("eg_configs/audiocaps/waveform/cnn14rnn_trm.yaml", 4981, 90395840),
                                ("eg_configs/clotho_v2/waveform/cnn14rnn_trm.yaml", 4368, 90081984)):
    cfg = train_util.load_config({ref!r} + "/" + yaml_path)                        # inherit_from resolved by the reference
    model = train_util.init_model_from_config(cfg["model"], print_fn=said.append)
    assert type(model) is A.TransformerModel, type(model)
    assert type(model.encoder) is A.CrnnEncoder and type(model.encoder.cnn) is A.Cnn14Encoder
    assert type(model.encoder.rnn) is A.RnnEncoder and type(model.decoder) is A.TransformerDecoder
    assert model.decoder.vocab_size == vocab and model.decoder.nlayers == 2
    assert sum(p.numel() for p in model.parameters()) == total, sum(p.numel() for p in model.parameters())
    assert not any(p.requires_grad for p in model.encoder.cnn.parameters())         # freeze_cnn: True
    assert model.encoder.freeze_cnn_bn
loss = train_util.init_obj_from_dict(cfg["loss"])
assert type(loss).__module__ == "audiocaption_amd.loss", type(loss)
assert any("not exist" in s for s in said), said                                    # the PANNs file of the YAML is absent here
print("REF-LOADER-OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the build container")
def test_reference_loader_builds_this_package_from_unchanged_yaml():
    code = _REF_LOADER.format(repo=REPO, ref=REF)
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert r.returncode == 0 and "REF-LOADER-OK" in r.stdout, r.stderr[-3000:]

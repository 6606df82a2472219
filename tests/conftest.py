import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cases():
    with open(os.path.join(GOLD, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def manifests():
    with open(os.path.join(GOLD, "manifests.json")) as f:
        return json.load(f)


def oracle_state_dict(cfg_name, manifest, max_seq_len, wseed=1234, end_bias=0.0):
    """Seeded state_dict for the oracle, built from the committed key manifest
    (names + shapes of the reference's state_dict) -- no reference needed."""
    from doc2tex_amd import synth
    from oracle import restatement as R

    cfg = synth.make_config(cfg_name, max_seq_len=max_seq_len)
    sd = {}
    for k, shape in manifest.items():
        if k.endswith("image_positional_encoder.pe"):
            continue  # 8 GB table in the reference; the oracle builds the crop it needs
        dt = torch.long if k.endswith("num_batches_tracked") else torch.float32
        t = synth.synth_tensor(k, shape, dt, seed=wseed, end_bias=end_bias)
        if t is None:
            if k.endswith("pos_embed"):
                gh, gw = R.resnet_out_hw(*cfg["max_dimension"])
                t = R.sincos_2d_table(shape[-1], -(-gh // 2), -(-gw // 2))
            elif k.endswith("pos_enc.pe"):
                t = R.word_pos_table(shape[1], shape[0])
        assert t is not None and list(t.shape) == list(shape), k
        sd[k] = t
    return cfg, sd


def engine_model(cfg_name, max_seq_len, wseed=1234, end_bias=0.0, beam_size=None, device="cuda"):
    """doc2tex_amd.Model on the GPU with the same seeded weights."""
    from doc2tex_amd import Model, synth

    cfg = synth.make_config(cfg_name, device=device, max_seq_len=max_seq_len, beam_size=beam_size)
    m = Model(cfg)
    tmpl = {k: v for k, v in m.state_dict().items() if not k.endswith("image_positional_encoder.pe")}
    for k in tmpl:
        assert tmpl[k].dtype in (torch.float32, torch.int64), k
    sd = synth.synth_state_dict(tmpl, seed=wseed, end_bias=end_bias)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("image_positional_encoder.pe") for k in missing), (missing, unexpected)
    m.eval()
    return cfg, m.to(device)

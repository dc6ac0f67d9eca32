"""CPU: the CLIP text-encoder oracle is PINNED against a real upstream implementation (transformers.CLIPTextModel
built from a config with the same synthetic weights), plus parameter-count and file-format checks."""
import pytest
import torch

from faceposegenerator_amd import spec as S, weights as W
from oracle import clip_oracle as CO


def _hf_model(cfg, sd):
    from transformers import CLIPTextConfig, CLIPTextModel
    hc = CLIPTextConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        max_position_embeddings=cfg.max_position_embeddings, vocab_size=cfg.vocab_size, hidden_act="gelu",
                        layer_norm_eps=cfg.layer_norm_eps, bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=0)
    m = CLIPTextModel(hc).eval()
    strip = not any(k.startswith("text_model.") for k in m.state_dict())
    missing, unexpected = m.load_state_dict({(k[len("text_model."):] if strip else k): v for k, v in sd.items()}, strict=False)
    assert not [k for k in missing if "position_ids" not in k] and not unexpected
    return m


def test_published_parameter_count():
    assert S.count_params(S.clip_text_param_shapes()) == 340_387_840


def test_oracle_matches_transformers_clip_text_model():
    cfg = S.TINY_CLIP
    sd = W.synth_clip(cfg, 5)
    m = _hf_model(cfg, sd)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, cfg.vocab_size, (3, 77), generator=g)
    with torch.no_grad():
        ref = m(ids)[0]
        got = CO.clip_text_forward(sd, cfg, ids)
    assert got.shape == (3, 77, cfg.hidden_size)
    assert (ref - got).abs().max().item() < 2e-5
    # causality: changing a later token must not change earlier positions
    ids2 = ids.clone()
    ids2[:, 40:] = (ids2[:, 40:] + 7) % cfg.vocab_size
    with torch.no_grad():
        got2 = CO.clip_text_forward(sd, cfg, ids2)
    assert torch.equal(got[:, :40], got2[:, :40]) and not torch.equal(got[:, 40:], got2[:, 40:])


def test_text_encoder_dir_roundtrip(tmp_path):
    cfg = S.TINY_CLIP
    sd = W.synth_clip(cfg, 5)
    W.save_text_encoder(str(tmp_path), cfg, sd)
    cfg2, sd2 = W.load_text_encoder(str(tmp_path))
    assert cfg2 == cfg and sd2.keys() == sd.keys() and all(torch.equal(sd2[k], sd[k]) for k in sd)
    assert W.load_text_encoder(str(tmp_path / "nope")) is None

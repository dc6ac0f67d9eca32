"""GPU parity of the text-conditioning row: new kernel features (causal attention, GELU epilogue,
token embedding), the CLIP text encoder against its oracle (which tests/test_clip_cpu.py pins to
transformers.CLIPTextModel) at reduced and FULL size (340 M parameters), and string prompts through the pipeline API."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"bf16": (6e-2, 2e-2), "f16": (1e-2, 3e-3)}          # max-abs on O(1) outputs, rel-RMS (see test_engine_gpu.py)


@pytest.fixture(scope="module", params=["bf16", "f16"])
def eng(request, lib):
    from faceposegenerator_amd import spec as S
    from faceposegenerator_amd.engine import HipEngine
    return HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, request.param)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _tol(eng):
    return 2.0 ** -7 if eng.dtype_name == "bf16" else 2.0 ** -9


@pytest.mark.parametrize("b,heads,n", [(2, 16, 77), (3, 2, 77), (1, 4, 200), (2, 1, 64)])
def test_causal_attention(eng, b, heads, n):
    c = heads * 64
    qkv = _rand((b * n, 3 * c), 60).to(eng.tdt)
    q, k, v = [t.float().view(b, n, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(b * n, c)
    p = qkv.data_ptr()
    out = eng.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, b, heads, n, n, n, causal=True)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    assert err <= _tol(eng) * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("m,n,k,tile", [(154, 512, 128, 0), (300, 4096, 1024, 0), (77, 256, 128, 4), (200, 100, 64, 5)])
def test_gemm_gelu_epilogue(eng, m, n, k, tile):
    a = _rand((m, k), 1).to(eng.tdt)
    w = _rand((n, k), 2, k ** -0.5).to(eng.tdt)
    bias = _rand((n,), 3)
    ref = F.gelu(a.float() @ w.float().t() + bias)
    out = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, act=1, tile=tile)
    out_direct = eng.gemm([(a, k, 1, 1, 1, 0)], w, n, m, 1, 1, bias=bias, act=1, tile=tile, flags=4)
    torch.cuda.synchronize()
    assert (out.float() - ref).abs().max().item() <= _tol(eng) * max(1.0, ref.abs().max().item())
    assert torch.equal(out, out_direct)


def test_embed_tokens(eng):
    from faceposegenerator_amd import _lib as L
    tok, pos = _rand((1000, 128), 5), _rand((77, 128), 6)
    ids = torch.randint(0, 1000, (3, 77), generator=torch.Generator().manual_seed(7)).to(DEV)
    out = torch.empty((3 * 77, 128), dtype=eng.tdt, device=DEV)
    L.check(eng.lib.idb_embed_tokens(ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), out.data_ptr(), 3, 77, 128, eng.dt,
                                     torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = (tok[ids] + pos[None]).reshape(3 * 77, 128).to(eng.tdt)
    assert torch.equal(out, ref)


def _stats(got, ref):
    d = (got.float().cpu() - ref).abs()
    return d.max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def test_tiny_text_encoder_matches_oracle(eng):
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.text_encoder import ClipTextEncoder
    from oracle import clip_oracle as CO
    cfg = S.TINY_CLIP
    sd = W.synth_clip(cfg, 5)
    te = ClipTextEncoder(eng, cfg, sd)
    ids = torch.randint(0, cfg.vocab_size, (3, 77), generator=torch.Generator().manual_seed(1))
    ref = CO.clip_text_forward(sd, cfg, ids)
    got = te(ids)[0]
    mx, rel = _stats(got, ref)
    print(f"[{eng.dtype_name}] tiny CLIP: max-abs {mx:.3e} rel-rms {rel:.3e}")
    assert mx < TOL[eng.dtype_name][0] and rel < TOL[eng.dtype_name][1]
    with pytest.raises(ValueError):
        te.encode(torch.full((1, 77), cfg.vocab_size))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_size_text_encoder_matches_oracle(lib, dtype):
    """SD-2.1-base's CLIP-H text model shapes (23 layers, width 1024, 340,387,840 parameters), synthetic weights."""
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.engine import HipEngine
    from faceposegenerator_amd.text_encoder import ClipTextEncoder
    from oracle import clip_oracle as CO
    cfg = S.SD21_CLIP
    sd = W.synth_clip(cfg, 99)
    e = HipEngine(S.TINY_UNET, S.TINY_VAE, None, None, DEV, dtype)
    te = ClipTextEncoder(e, cfg, sd)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, 49000, (2, 77), generator=g)
    ids[:, 0], ids[0, 20:], ids[1, 9:] = cfg.bos_token_id, 0, 0
    ids[0, 19], ids[1, 8] = cfg.eos_token_id, cfg.eos_token_id
    with torch.no_grad():
        ref = CO.clip_text_forward(sd, cfg, ids)
    got = te.encode(ids)
    mx, rel = _stats(got, ref)
    print(f"[{dtype}] full-size CLIP-H: max-abs {mx:.3e} rel-rms {rel:.3e} (|ref| max {ref.abs().max():.2f})")
    assert rel < TOL[dtype][1] and mx < TOL[dtype][0] * max(1.0, ref.abs().max().item())
    del te, e
    torch.cuda.empty_cache()


def _write_tokenizer(d):
    def bytes_to_unicode():
        bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
        cs, n = bs[:], 0
        for b in range(256):
            if b not in bs:
                bs.append(b)
                cs.append(256 + n)
                n += 1
        return [chr(c) for c in cs]
    chars = bytes_to_unicode()
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": 256 + i for i, c in enumerate(chars)})
    vocab.update({"fa": 600, "fac": 601, "face</w>": 602, "<|startoftext|>": 998, "<|endoftext|>": 999})
    os.makedirs(d, exist_ok=True)
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\nf a\nfa c\nfac e</w>\n")
    json.dump({"bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "pad_token": "!",
               "model_max_length": 77, "tokenizer_class": "CLIPTokenizer"}, open(os.path.join(d, "tokenizer_config.json"), "w"))


def test_string_prompts_through_the_pipeline(lib, tmp_path):
    """The reference's call form — pipe(prompt=str, negative_prompt=str, ...) (inference_ID-Booth.py:138) — on a local
    diffusers-layout model directory with text_encoder/ and tokenizer/ (synthetic weights and vocabulary)."""
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.driver import NEGATIVE_PROMPT
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from faceposegenerator_amd.scheduler import DDPMScheduler
    from oracle import clip_oracle as CO, sd21_oracle as O
    root = str(tmp_path / "model")
    clip_cfg = S.ClipTextConfig(**{**S.TINY_CLIP.__dict__, "hidden_size": 128})
    usd, vsd, csd = W.synth_unet(S.TINY_UNET, 7), W.synth_vae(S.TINY_VAE, 8), W.synth_clip(clip_cfg, 5)
    W.save_model_dir(root, usd, vsd, S.TINY_UNET, S.TINY_VAE)
    W.save_text_encoder(root, clip_cfg, csd)
    _write_tokenizer(os.path.join(root, "tokenizer"))
    pipe = StableDiffusionPipeline.from_pretrained(root, torch_dtype=torch.float16).to(DEV)
    pipe.scheduler = DDPMScheduler.from_pretrained(root, subfolder="scheduler")
    pipe.set_progress_bar_config(disable=True)
    prompt = "face portrait photo of male sks person, forest background"
    steps = 2
    out = pipe(prompt=prompt, negative_prompt=NEGATIVE_PROMPT, output_type="latent", generator=torch.Generator().manual_seed(3),
               num_inference_steps=steps, guidance_scale=5.0, width=128, height=128)
    # oracle: same tokenizer (host-side), CLIP oracle, sampler oracle
    tok = pipe.tokenizer
    ids = tok([prompt, NEGATIVE_PROMPT], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert ids[0, 0].item() == 998 and 602 in ids[0].tolist() and ids[0, -1].item() == 0
    emb = CO.clip_text_forward(csd, clip_cfg, ids)
    noise = O.draw_noise(torch.Generator().manual_seed(3), 1, steps, (16, 16))
    ref = O.sample(usd, S.TINY_UNET, emb[0:1], emb[1:2], noise, steps, 5.0)
    mx, rel = _stats(out.images, ref)
    print(f"string-prompt pipeline (f16): latents max-abs {mx:.3e} rel-rms {rel:.3e}")
    assert rel < 4 * TOL["f16"][1]
    pe, ne = pipe.encode_prompt([prompt, prompt], None, True)
    assert pe.shape == (2, 77, 128) and ne.shape == (2, 77, 128)
    assert (pe[0].cpu() - emb[0]).abs().max().item() < TOL["f16"][0]
    with pytest.raises(ValueError):
        pipe.encode_prompt([prompt, prompt], ["a"], True)

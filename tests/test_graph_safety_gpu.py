"""Captured HIP graphs must keep every buffer they reference alive: a replay that writes through a freed address corrupts
whatever tensor the caching allocator handed that memory to next.  Two regressions found by review of round 1:
  * the x0 history of the DPM-Solver++ sampler was a local of the capturing call;
  * the GroupNorm / split-K workspaces were re-allocated on growth while older graphs kept the old address.
Each test replays a graph after filling the allocator's free memory with sentinel tensors and checks the sentinels."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sentinels(n, nbytes, value):
    return [torch.full((nbytes,), value, dtype=torch.uint8, device=DEV) for _ in range(n)]


def _intact(sent, value):
    return all(bool((s == value).all().item()) for s in sent)


def test_multistep_graph_replay_touches_no_freed_memory(lib):
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    from faceposegenerator_amd.scheduler import DPMSolverMultistepScheduler
    ucfg = S.TINY_UNET
    pipe = StableDiffusionPipeline(ucfg, S.TINY_VAE, W.synth_unet(ucfg, 7), W.synth_vae(S.TINY_VAE, 8), torch_dtype="f16").to(DEV)
    pipe.scheduler = DPMSolverMultistepScheduler.from_config(pipe.scheduler.config, variance_type="fixed_small")
    g = torch.Generator().manual_seed(13)
    pe, ne = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g), torch.randn(2, 77, ucfg.cross_attention_dim, generator=g)
    init = torch.randn(2, 4, 16, 16, generator=g)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=6, guidance_scale=7.5, height=128, width=128,
              output_type="latent", latents=init)
    first = pipe(**kw).images.clone()                     # eager warm-up + capture + first replay
    torch.cuda.synchronize()
    # the history buffer is 2*B*4*h*w fp32 = 16 KiB here: take every free block of that class (and some larger ones)
    sent = _sentinels(256, 2 * 2 * 4 * 16 * 16 * 4, 0xA5) + _sentinels(32, 1 << 20, 0xA5)
    torch.cuda.synchronize()
    again = pipe(**kw).images
    torch.cuda.synchronize()
    assert _intact(sent, 0xA5), "graph replay wrote into memory it no longer owns"
    assert torch.equal(first, again)
    ent = next(v for k, v in pipe._engine()._graphs.items() if k[-2] is True)     # key: (..., multistep, lora groups)
    assert ent["hist"] is not None


def test_workspace_growth_keeps_older_graphs_valid(lib):
    from faceposegenerator_amd import spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    ucfg = S.TINY_UNET
    pipe = StableDiffusionPipeline(ucfg, S.TINY_VAE, W.synth_unet(ucfg, 7), W.synth_vae(S.TINY_VAE, 8), torch_dtype="f16").to(DEV)
    eng = pipe._engine()

    def call(B, seed):
        g = torch.Generator().manual_seed(seed)
        pe, ne = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g), torch.randn(B, 77, ucfg.cross_attention_dim, generator=g)
        return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2, guidance_scale=5.0, height=128, width=128,
                    output_type="latent", generator=torch.Generator().manual_seed(seed)).images.clone()

    small = call(1, 3)
    gn0, ws0 = eng._gn_ws.data_ptr(), eng._ws.data_ptr()
    call(40, 4)                                            # CFG batch 80 > 64: the GroupNorm workspace grows
    assert eng._gn_ws.data_ptr() != gn0, "test premise: this batch must outgrow the initial GroupNorm workspace"
    assert any(t.data_ptr() == gn0 for t in eng._retired)
    torch.cuda.synchronize()
    sent = _sentinels(64, 1 << 20, 0x5A) + _sentinels(4, 64 << 20, 0x5A)
    torch.cuda.synchronize()
    again = call(1, 3)                                     # replays the graph captured before the growth
    torch.cuda.synchronize()
    assert _intact(sent, 0x5A), "an older graph wrote into a workspace that had been freed"
    assert torch.equal(small, again)
    assert ws0 == eng._ws.data_ptr() or any(t.data_ptr() == ws0 for t in eng._retired)

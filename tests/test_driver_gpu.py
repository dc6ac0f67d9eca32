"""End-to-end on the GPU with the reduced-width graph: the batched driver (prompt policy, per-identity noise streams, per-identity
LoRA) -> pipeline -> uint8 images -> PNG/JPG sink in the reference's layout (inference_ID-Booth.py:86-156), and the
align-and-crop stage on the generated images (utils/detect_align_crop_data.py:172-181)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_driver_pipeline_sink_and_crop(lib, tmp_path):
    from faceposegenerator_amd import driver as D, face_align as FA, spec as S, weights as W
    from faceposegenerator_amd.pipeline import StableDiffusionPipeline
    ucfg = S.TINY_UNET
    pipe = StableDiffusionPipeline(ucfg, S.TINY_VAE, W.synth_unet(ucfg, 7), W.synth_vae(S.TINY_VAE, 8), torch_dtype="bf16").to(DEV)
    cfg = D.PolicyConfig(num_prompts=3, num_inference_steps=3, height=128, width=128, models_to_test=("DreamBooth", "ID-Booth"))
    ids = ["ID_1", "ID_2"]
    items = D.build_work_list(ids, {"ID_1": "M", "ID_2": "F"}, cfg)
    embed = D.synthetic_embed_fn(ucfg.cross_attention_dim)
    loras = {}

    def lora_for(model, which_id):
        return loras.setdefault((model, which_id), W.synth_lora(ucfg, seed=len(loras) + 1))

    imgs, order = D.generate(pipe, items, embed, cfg, lora_for=lora_for, rank=0, world=1, max_batch=2)
    assert imgs.shape == (len(items), 128, 128, 3) and imgs.dtype == torch.uint8 and len(order) == len(items) == 2 * 2 * 3
    # one work item recomputed on its own: same LoRA, same noise stream position, same embeddings -> the same picture
    k = 4
    it = order[k]
    pipe.load_lora_weights(lora_for(it.model_name, it.which_id))
    noise = D.draw_noise_sequential(it.id_number, 1, cfg.num_inference_steps, (4, 16, 16), it.stream_offset)
    neg = D.NEGATIVE_PROMPT if not cfg.do_not_use_negative_prompt else ""
    one = pipe(prompt_embeds=embed([it.prompt]), negative_prompt_embeds=embed([neg]), num_inference_steps=cfg.num_inference_steps,
               guidance_scale=cfg.guidance_scale, height=128, width=128, output_type="uint8", noise=noise).images
    d = (one[0].int() - imgs[k].int()).abs()
    assert d.max().item() <= 6 and (d <= 1).float().mean().item() > 0.97      # batch of 2 vs batch of 1: other GEMM tiles, bf16
    # different identities / models give different pictures
    assert (imgs[0].int() - imgs[6].int()).abs().max().item() > 8
    paths = D.save_outputs(imgs.cpu(), order, str(tmp_path), cfg)
    from PIL import Image
    first = [p for p in paths if p.endswith(order[0].file_name())][0]
    assert np.array_equal(np.array(Image.open(first)), imgs[0].cpu().numpy())
    # align-and-crop on the GPU-resident outputs: landmarks = template pushed through a similarity into the 128x128 frame
    m = np.array([[0.9, -0.1, 12.0], [0.1, 0.9, 8.0]])
    lm = FA.ARCFACE_TEMPLATE @ m[:, :2].T + m[:, 2]
    crops = FA.norm_crop(imgs[:4], np.stack([lm] * 4))
    assert crops.shape == (4, 112, 112, 3) and crops.dtype == torch.uint8 and crops.is_cuda
    from oracle import face_align_oracle as FO
    assert np.array_equal(crops[2].cpu().numpy(), FO.norm_crop(imgs[2].cpu().numpy(), lm))

"""CPU tests of the generation driver: the reference's prompt/seed policy (SURVEY.md Appendix E known answers),
identity sharding, and the N > 1 path — the one all-gather of uint8 images — on gloo with world_size 2."""
import os
import socket

import pytest
import torch

from faceposegenerator_amd import driver as D

IDS = ["ID_10", "ID_2", "ID_1", "ids.json"]
GENDERS = {"ID_1": "M", "ID_2": "F", "ID_10": "M"}


def test_natural_sort():
    assert sorted(["ID_10", "ID_2", "ID_1"], key=D.natural_keys) == ["ID_1", "ID_2", "ID_10"]
    assert D.atoi("12") == 12 and D.atoi("ab") == "ab"


def test_prompt_policy_known_answers():
    cfg = D.PolicyConfig()
    combos = D.prompt_combinations(cfg)
    assert len(combos) == 21 and combos[0] == "" and combos[1] == "forest background" and combos[11] == "forest background"
    items = D.build_work_list(IDS, GENDERS, cfg)
    assert len(items) == 3 * 3 * 21 and [items[0].which_id, items[63].which_id, items[126].which_id] == ["ID_1", "ID_2", "ID_10"]
    want = ["face side-portrait photo of male sks person, city street background",
            "face side-portrait photo of male sks person, beach background",
            "face portrait photo of male sks person, forest background",
            "face portrait photo of male sks person, construction site background",
            "face side-portrait photo of male sks person, laboratory background",
            "face portrait photo of male sks person, bus background"]
    assert [it.prompt for it in items[:6]] == want                                  # id 0, DreamBooth, prompts 0-5
    idb = [it for it in items if it.id_number == 0 and it.model_name == "ID-Booth"]
    assert ["side" in it.prompt for it in idb[:6]] == [True, False, True, True, True, True]
    assert [it.prompt.split(", ")[1] for it in idb[:3]] == [it.prompt.split(", ")[1] for it in items[:3]]   # same backgrounds
    assert any(it.prompt.endswith(", ") for it in items if it.id_number == 0 and it.model_name == "DreamBooth")  # "" entry
    assert "female" in items[63].prompt
    assert [it.stream_offset for it in items[:21]] == list(range(21))
    assert items[0].file_name() == "0_0_" + want[0] + ".png"
    assert cfg.output_folder() == "Generated_Samples/FacePortrait_Photo_21_Gender_Pose_Background"
    assert D.NEGATIVE_PROMPT.startswith("cartoon, cgi") and D.NEGATIVE_PROMPT.endswith("landscape")
    with pytest.raises(KeyError):
        D.build_work_list(["ID_7"], {}, cfg)


def test_other_flag_combinations():
    age = D.PolicyConfig(add_age=True, add_background=True, add_gender=False, add_pose=False, num_prompts=10)
    items = D.build_work_list(["ID_1"], None, age)
    assert len(D.prompt_combinations(age)) == 44 and len(items) == 30
    assert all(it.prompt.startswith("face portrait photo of") for it in items)
    plain = D.PolicyConfig(add_background=False, add_gender=False, add_pose=False, num_prompts=3)
    assert {it.prompt for it in D.build_work_list(["ID_1"], None, plain)} == {D.ORIGINAL_PROMPT}
    assert len(D.prompt_combinations(D.PolicyConfig(num_prompts=100))) == 100


def test_sequential_noise_matches_per_call_generator_stream():
    """Batching the 21 prompts of an (identity, model) pair must not change any call's noise: call k of the shared
    generator draws its latents, then one tensor per step (inference_ID-Booth.py:111 sits outside the prompt loop)."""
    steps, shape = 3, (4, 8, 8)
    ref_gen = torch.Generator().manual_seed(5)
    calls = [[torch.randn((1,) + shape, generator=ref_gen) for _ in range(steps + 1)] for _ in range(4)]
    got = D.draw_noise_sequential(5, 4, steps, shape)
    for k in range(4):
        for s in range(steps + 1):
            assert torch.equal(got[s, k], calls[k][s][0])
    tail = D.draw_noise_sequential(5, 2, steps, shape, first_call=2)
    assert torch.equal(tail, got[:, 2:])


def test_sharding_covers_every_item_once():
    items = D.build_work_list([f"ID_{i}" for i in range(1, 12)], {f"ID_{i}": "M" for i in range(1, 12)})
    for world in (1, 2, 8):
        seen = []
        for r in range(world):
            mine = D.shard_work(items, r, world)
            assert all(it.id_number % world == r for it in mine)
            seen += mine
        assert sorted(id(x) for x in seen) == sorted(id(x) for x in items)
    assert D.shard_identities(11, 3, 8) == [3]


class _FakePipe:
    """Stands in for the GPU pipeline in the collective test: returns images that encode the work item."""
    device = torch.device("cpu")
    vae_scale_factor = 8

    class unet_config:
        in_channels = 4

    def load_lora_weights(self, x):
        self.lora = x

    def __call__(self, prompt_embeds, negative_prompt_embeds, noise, height, width, **kw):
        from types import SimpleNamespace
        b = prompt_embeds.shape[0]
        assert noise.shape == (kw["num_inference_steps"] + 1, b, 4, height // 8, width // 8)
        img = torch.zeros((b, height, width, 3), dtype=torch.uint8)
        for k in range(b):
            img[k] = int(prompt_embeds[k, 0, 0].abs().mul(1000).item()) % 251
        return SimpleNamespace(images=img)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = D.PolicyConfig(num_prompts=3, num_inference_steps=2, height=16, width=16)
    ids = ["ID_1", "ID_2", "ID_3"]
    items = D.build_work_list(ids, {i: "M" for i in ids}, cfg)
    embed = D.synthetic_embed_fn(8, n_ctx=4)
    imgs, order = D.generate(_FakePipe(), items, embed, cfg, lora_for=lambda m, i: {"m": m, "i": i}, rank=rank, world=world,
                             max_batch=2)
    q.put((rank, imgs[:, 0, 0, 0].tolist(), [(it.id_number, it.model_name, it.prompt_index) for it in order]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_all_gather_on_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, vals0, order0), (r1, vals1, order1) = res
    assert vals0 == vals1 and order0 == order1                    # every rank holds the same gathered images
    assert len(vals0) == 3 * 3 * 3
    # rank-major order: identities 0 and 2 (rank 0) first, then identity 1 (rank 1)
    assert [o[0] for o in order0] == [0] * 9 + [2] * 9 + [1] * 9
    # the pixel value encodes the prompt embedding -> gathered image k belongs to work item order[k]
    cfg = D.PolicyConfig(num_prompts=3, num_inference_steps=2, height=16, width=16)
    ids = ["ID_1", "ID_2", "ID_3"]
    items = {(it.id_number, it.model_name, it.prompt_index): it for it in D.build_work_list(ids, {i: "M" for i in ids}, cfg)}
    embed = D.synthetic_embed_fn(8, n_ctx=4)
    for v, key in zip(vals0, order0):
        assert v == int(embed([items[key].prompt])[0, 0, 0].abs().mul(1000).item()) % 251


def test_sink_writes_reference_layout(tmp_path):
    cfg = D.PolicyConfig(num_prompts=2, height=8, width=8, models_to_test=("DreamBooth", "ID-Booth"))
    items = D.build_work_list(["ID_1"], {"ID_1": "F"}, cfg)
    imgs = torch.arange(len(items) * 8 * 8 * 3, dtype=torch.int64).remainder(255).to(torch.uint8).reshape(len(items), 8, 8, 3)
    paths = D.save_outputs(imgs, items, str(tmp_path), cfg)
    d = tmp_path / cfg.output_folder() / "DreamBooth" / f"ID_1_{D.CHECKPOINT}_{D.ARCH}"
    assert (d / items[0].file_name()).is_file()
    comp = tmp_path / cfg.output_folder() / "Comparison" / f"ID_1_{D.CHECKPOINT}_{D.ARCH}_5.0.jpg"
    assert comp.is_file() and len(paths) == len(items) + 1
    from PIL import Image
    import numpy as np
    assert np.array_equal(np.array(Image.open(d / items[0].file_name())), imgs[0].numpy())
    assert Image.open(comp).size == (8 * 2, 8 * 2)          # nrow = num_prompts, one row per model


def test_plan_calls_groups_identities_of_one_model_into_single_calls():
    """BASELINE configs[2] as stated: 8 identities x 8 prompts -> ONE batch-64 call with 8 LoRA groups; the reference's own work list
    (1 identity x 3 models x 21 prompts) stays three calls; order and stream offsets are kept inside every identity."""
    ids = [f"ID_{i + 1}" for i in range(8)]
    cfg = D.PolicyConfig(num_prompts=8, models_to_test=("ID-Booth",))
    items = D.build_work_list(ids, {i: "M" for i in ids}, cfg)
    calls = D.plan_calls(items, 64, 8)
    assert len(calls) == 1 and [len(sub) for sub in calls[0]] == [8] * 8
    assert [sub[0].which_id for sub in calls[0]] == ids and all([it.stream_offset for it in sub] == list(range(8)) for sub in calls[0])
    assert [[it for it in sub] for sub in D.plan_calls(items, 64, 1)[0]] == [[it for it in items[:8]]] and len(D.plan_calls(items, 64, 1)) == 8
    # a cap of 32 images per call: two calls of 4 identities
    calls = D.plan_calls(items, 32, 8)
    assert [len(c) for c in calls] == [4, 4] and sum(len(s) for c in calls for s in c) == 64
    # 3 models x 21 prompts of one identity: nothing to group
    cfg3 = D.PolicyConfig()
    items3 = D.build_work_list(["ID_1"], {"ID_1": "F"}, cfg3)
    calls3 = D.plan_calls(items3, 64, 8)
    assert len(calls3) == 3 and all(len(c) == 1 and len(c[0]) == 21 for c in calls3)
    # every item appears exactly once, whatever the grouping
    for g in (1, 2, 8):
        flat = [it for c in D.plan_calls(items, 64, g) for s in c for it in s]
        assert sorted(map(id, flat)) == sorted(map(id, items))

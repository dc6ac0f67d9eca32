"""Host engine: drives the gfx950 kernels of libidb_kernels.so through the SD-2.1 UNet forward, the
CFG + DDPM sampling loop and the VAE decoder.

It replaces what ``StableDiffusionPipeline.__call__`` executes inside diffusers/torch for
``/root/reference/inference_ID-Booth.py:138``.  PyTorch is used only for device
memory, streams and HIP-graph capture; every arithmetic step is a C-ABI kernel call (include/idb_kernels.h).

Data layout in HBM
  * activations: NHWC in the operand dtype (bf16 default, f16 optional) — a [B,H,W,C] feature map and
    the [B, H*W, C] token matrix of the transformer blocks are the same buffer, so the NCHW<->NLC
    permutes of the reference graph disappear;
  * weights: packed once at load to [N][K] operand dtype (conv: [Cout][tap][Cin]); Q/K/V and
    cross-attention K/V projections are row-concatenated; the 1x1 conv_shortcut of a ResnetBlock2D is
    concatenated along K behind conv2 so the block's second conv, shortcut and residual add are ONE
    GEMM; GEGLU projection rows are interleaved so value and gate land in the same lane;
  * fp32 everywhere precision matters: latents, scheduler state, time-embedding path, GroupNorm /
    LayerNorm statistics, softmax, accumulators, biases.
  * per-call invariants are hoisted out of the 30-step loop: the time-embedding MLP and the 22
    time_emb_proj projections for all timesteps (one [steps, sum(Cout)] table) and the 16
    cross-attention K/V projections of the prompt embeddings.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import spec as S

SD = Dict[str, torch.Tensor]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Arena:
    """Size-matched free-list allocator over torch uint8 blocks.  The op sequence of a forward is
    static, so after one eager warm-up no new device memory is requested (HIP-graph capturable)."""

    def __init__(self, device):
        self.device = device
        self.blocks: List[torch.Tensor] = []
        self.free_ids: List[int] = []
        self.by_ptr: Dict[int, int] = {}
        self.live: Dict[int, int] = {}
        self.total_bytes = 0

    def alloc(self, shape: Sequence[int], dtype: torch.dtype) -> torch.Tensor:
        numel = 1
        for d in shape:
            numel *= int(d)
        nbytes = max(256, numel * torch.empty((), dtype=dtype).element_size())
        best, best_sz = -1, None
        for i in self.free_ids:
            sz = self.blocks[i].numel()
            if sz >= nbytes and sz <= 2 * nbytes + 4096 and (best_sz is None or sz < best_sz):
                best, best_sz = i, sz
        if best < 0:
            blk = torch.empty(((nbytes + 255) // 256) * 256, dtype=torch.uint8, device=self.device)
            self.blocks.append(blk)
            best = len(self.blocks) - 1
            self.by_ptr[blk.data_ptr()] = best
            self.total_bytes += blk.numel()
        else:
            self.free_ids.remove(best)
        blk = self.blocks[best]
        self.live[blk.data_ptr()] = best
        nb = numel * torch.empty((), dtype=dtype).element_size()
        return blk[:nb].view(dtype).view(*shape)

    def free(self, t: Optional[torch.Tensor]) -> None:
        if t is None:
            return
        i = self.live.pop(t.data_ptr(), None)
        if i is None:
            raise RuntimeError("Arena.free: tensor is not a live arena allocation (double free?)")
        self.free_ids.append(i)

    def reset(self) -> None:
        self.live.clear()
        self.free_ids = list(range(len(self.blocks)))


class HipEngine:
    def __init__(self, ucfg: S.UNetConfig, vcfg: S.VAEConfig, unet_sd: Optional[SD], vae_sd: Optional[SD],
                 device="cuda:0", dtype: str = "bf16"):
        if not torch.cuda.is_available():
            raise L.IdbError("HipEngine needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = L.load()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        L.check(self.lib.idb_device_check(self.device.index or 0), "idb_device_check")
        if dtype not in ("bf16", "f16", "fp8"):
            raise ValueError("dtype must be 'bf16', 'f16' or 'fp8'")
        # "fp8" (BASELINE configs[4]): the 3x3 convolutions of the UNet's ResnetBlock2Ds run on the fp8 MFMA path — GroupNorm+SiLU
        # writes e4m3 activations with one fixed scale, weights are e4m3 with a scale per output channel (idb_gemm_fp8) — and
        # everything else (attention, projections, residual stream, VAE) is the f16 path.
        self.fp8 = dtype == "fp8"
        self.dtype_name = dtype
        self.dt = L.IDB_BF16 if dtype == "bf16" else L.IDB_F16
        self.tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
        self.ucfg, self.vcfg = ucfg, vcfg
        self.ugraph = S.unet_graph(ucfg)
        self.vgraph = S.vae_graph(vcfg)
        self.arena = Arena(self.device)
        self._ws = torch.empty(64 << 20, dtype=torch.uint8, device=self.device)
        self._gn_ws = torch.empty(1 << 20, dtype=torch.uint8, device=self.device)
        self._retired: List[torch.Tensor] = []      # outgrown workspaces that captured graphs may still reference
        self._counters = torch.zeros(1 << 16, dtype=torch.int32, device=self.device)   # split-K tickets (self-resetting)
        # GroupNorm single-launch hand-off counters (self-resetting); opt-in: measured slower than two launches (idb_norm.hip)
        self._gn_sync = torch.zeros(1 << 14, dtype=torch.int32, device=self.device) if os.environ.get("IDB_GN_SYNC") == "1" else None
        # first GroupNorm pass produced by the GEMM that writes the tensor (idb_gemm_desc.gn_partials); IDB_GN_FUSE=0 disables
        self._gn_fuse = os.environ.get("IDB_GN_FUSE", "1") != "0"
        # LayerNorm folded into the consuming projection (to_q/k/v, attn2.to_q, GEGLU-in): row statistics from the producer's epilogue,
        # rstd / mean correction in the consumer's; IDB_LN_FOLD=0 keeps the idb_layernorm launches
        self._ln_fold = os.environ.get("IDB_LN_FOLD", "1") != "0"
        self._fold_cache: Dict[Tuple[int, int, int, bool], bool] = {}
        self._ff_chunk_bytes = int(os.environ.get("IDB_FF_CHUNK_MB", "160")) << 20       # 0: the feed-forward in one piece
        self._gn_epi = os.environ.get("IDB_GN_EPILOGUE", "1") != "0"      # GroupNorm statistics from non-split GEMM epilogues
        # GroupNorm(+SiLU) applied by normalizer waves inside the consuming conv / proj_in (idb_gemm_desc.gn_in_*) wherever the plan is a
        # one-workgroup-per-CU loader-wave plan (the batch-1 UNet); IDB_GN_CONV=0: idb_groupnorm + idb_gemm everywhere
        self._gn_conv = os.environ.get("IDB_GN_CONV", "1") != "0" and dtype != "fp8"
        # measured per shape at B_eff 2 (tools/bench_gnfuse.py, profiles/r03): Transformer2DModel.norm + proj_in at the 64x64 level 17.9 ->
        # 13.3 us (no SiLU, 5 K-steps, 320-channel table); every 3x3 conv LOSES in that tap-major form (34 -> 44 us on conv 320->320: every
        # tap re-normalises its pixels), deeper proj_in lose to the table prologue: proj_in fuses up to 320 channels.  The resnet convs
        # fuse through the PATCH-resident conv instead (idb_conv_patch_kernel<GN>: the patch loaders normalise each halo patch once per
        # chunk): 30 of the 44 resnet convs per CFG forward fuse and lose their gn_apply launch, the fused convs cost 3-7 us more each; batch 1 +0.5 %
        # (6.592 -> 6.622 / 6.632 images/s, one box).  IDB_GN_CONV_RESNET=0: idb_groupnorm + idb_gemm for every resnet conv
        self._gn_conv_resnet = os.environ.get("IDB_GN_CONV_RESNET", "1") == "1"
        self._gn_conv_max_c = int(os.environ.get("IDB_GN_CONV_MAX_C", "320"))
        self._gn_conv_cache: Dict[tuple, bool] = {}
        # weights in the K-tiled 16-row-block layout (idb_tile_weight): a workgroup's K loop reads each of its row blocks as one
        # contiguous stream instead of 128-byte pieces at a K*2-byte stride; IDB_W_TILED=0 keeps [n][K] rows
        self._w_tiled = os.environ.get("IDB_W_TILED", "1") != "0"
        self.groups = 1                                # > 1: set_lora_groups is active (one merged LoRA set per group of samples)
        self.wg: Dict[str, torch.Tensor] = {}          # grouped copies [G][...] of the LoRA-affected operands (tiled matrices, u / v)
        self._rep = 1
        self.x8_scale: Dict[str, float] = {}           # fp8 path: e4m3 scale of each GroupNorm+SiLU output (fp8_act_scale)
        self.w: Dict[str, torch.Tensor] = {}
        self.w_rows: Dict[str, torch.Tensor] = {}      # [n][K] row form of the LoRA-affected matrices (set_lora writes here, then re-tiles)
        self.master: Dict[str, torch.Tensor] = {}
        self.tproj_off: Dict[str, int] = {}
        self.tproj_total = 0
        self._pinned: set = set()
        self.launch_log: Optional[list] = None
        self.last_forward_launches = 0
        self.taps: Optional[dict] = None      # tests: {module name: fp32 copy [B*h*w, C] of the block's output} (eager runs only)
        self.lora_loaded = False
        if unet_sd is not None:
            self._pack_unet(unet_sd)
        if vae_sd is not None:
            self._pack_vae(vae_sd)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------
    # weight packing
    # ------------------------------------------------------------------------------------
    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _pack_conv(self, w: torch.Tensor) -> torch.Tensor:
        cout, cin, kh, kw = w.shape
        src = self._f32(w)
        dst = torch.empty((cout, kh * kw * cin), dtype=self.tdt, device=self.device)
        L.check(self.lib.idb_pack_conv_weight(src.data_ptr(), dst.data_ptr(), cout, cin, kh * kw, self.dt, _stream()),
                "idb_pack_conv_weight")
        return dst

    def _pack_mat(self, w: torch.Tensor, geglu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows, cols = w.shape
        src = w if (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()) else self._f32(w)
        dst = out if out is not None else torch.empty((rows, cols), dtype=self.tdt, device=self.device)
        L.check(self.lib.idb_pack_matrix(src.data_ptr(), dst.data_ptr(), rows, cols, int(geglu), self.dt, _stream()),
                "idb_pack_matrix")
        return dst

    def _empty_tiled(self, n: int, k: int) -> torch.Tensor:
        t = torch.empty((self.lib.idb_tiled_weight_bytes(n, k) // 2,), dtype=self.tdt, device=self.device)
        t._tiled = (n, k)
        return t

    def _tile_static_weights(self) -> None:
        """Every packed [n][K] GEMM weight that is not LoRA-affected -> tiled layout (the row form is dropped)."""
        if not self._w_tiled:
            return
        for key, t in list(self.w.items()):
            if key in self.w_rows or getattr(t, "_tiled", None) is not None or key.endswith("attentions.0.v.w"):
                continue                                  # (the VAE attention's to_v matrix is used as an A operand: rows)
            if t.dtype == self.tdt and t.ndim == 2 and (key.endswith(".w") or key.endswith(".wln") or key.endswith(".wsplit")) and t.shape[1] % 64 == 0:
                self.w[key] = self.tile_weight(t.contiguous())

    def tile_weight(self, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n][k] operand-dtype rows -> the K-tiled 16-row-block layout (idb_tile_weight; idb_gemm_desc.w_layout = 1).  The result is
        tagged ``_tiled = (n, k)``; ``gemm`` reads the tag.  ``out``: an existing tiled buffer to refill in place (LoRA switch under a
        captured graph)."""
        n, k = w.shape
        if out is None:
            out = torch.empty((self.lib.idb_tiled_weight_bytes(n, k) // 2,), dtype=self.tdt, device=self.device)
            out._tiled = (n, k)
        L.check(self.lib.idb_tile_weight(w.data_ptr(), out.data_ptr(), n, k, self.dt, _stream()), "idb_tile_weight")
        return out

    @staticmethod
    def _geglu_perm(rows: int) -> torch.Tensor:
        p = torch.arange(rows)
        blk, t = p // 32, p % 32
        return torch.where(t < 16, 16 * blk + t, rows // 2 + 16 * blk + (t - 16))

    def _pack_resnet(self, sd: SD, name: str, has_temb: bool, split_at: int = 0) -> None:
        w = self.w
        for i in ("1", "2"):
            w[f"{name}.gn{i}.g"] = self._f32(sd[f"{name}.norm{i}.weight"])
            w[f"{name}.gn{i}.b"] = self._f32(sd[f"{name}.norm{i}.bias"])
        w[f"{name}.conv1.w"] = self._pack_conv(sd[f"{name}.conv1.weight"])
        w[f"{name}.conv1.b"] = self._f32(sd[f"{name}.conv1.bias"])
        if split_at:                                    # up-path resnet: norm1 + conv1 over cat[x, skip] as TWO 3x3 sources (fused GroupNorm)
            w1 = sd[f"{name}.conv1.weight"]
            w[f"{name}.conv1.wsplit"] = torch.cat([self._pack_conv(w1[:, :split_at].contiguous()), self._pack_conv(w1[:, split_at:].contiguous())],
                                                  dim=1).contiguous()
        w2 = self._pack_conv(sd[f"{name}.conv2.weight"])
        b2 = self._f32(sd[f"{name}.conv2.bias"])
        if self.fp8 and has_temb:                       # UNet resnets: e4m3 copies of the two 3x3 convs, the 1x1 shortcut stays f16
            w[f"{name}.conv1.w8"], w[f"{name}.conv1.s8"] = self.pack_weight_fp8(sd[f"{name}.conv1.weight"])
            w[f"{name}.conv2.w8"], w[f"{name}.conv2.s8"] = self.pack_weight_fp8(sd[f"{name}.conv2.weight"])
            for i in ("1", "2"):                        # scale of this GroupNorm+SiLU's e4m3 output, from ITS affine parameters
                self.x8_scale[f"{name}.gn{i}"] = self.fp8_act_scale(sd[f"{name}.norm{i}.weight"], sd[f"{name}.norm{i}.bias"])
        if f"{name}.conv_shortcut.weight" in sd:
            ws = sd[f"{name}.conv_shortcut.weight"]
            ws = self._pack_mat(ws.reshape(ws.shape[0], ws.shape[1]))
            if self.fp8 and has_temb:
                w[f"{name}.sc.w"] = ws
            w2 = torch.cat([w2, ws], dim=1).contiguous()          # [Cout][9*Cout + Cin]
            b2 = self._f32(sd[f"{name}.conv2.bias"].float() + sd[f"{name}.conv_shortcut.bias"].float())      # host-side fp32 add of two load-time vectors
            w[f"{name}.has_shortcut"] = torch.ones(1)
        w[f"{name}.conv2.w"] = w2
        w[f"{name}.conv2.b"] = b2

    def _pack_unet(self, sd: SD) -> None:
        w, g = self.w, self.ugraph
        w["conv_in.w"] = self._f32(sd["conv_in.weight"])
        w["conv_in.b"] = self._f32(sd["conv_in.bias"])
        for k in ("linear_1", "linear_2"):
            w[f"te.{k}.w"] = self._f32(sd[f"time_embedding.{k}.weight"])
            w[f"te.{k}.b"] = self._f32(sd[f"time_embedding.{k}.bias"])
        resnets: List[S.ResnetSpec] = []
        attns: List[S.AttnSpec] = []
        for blk in g.down:
            resnets += blk["resnets"]
            attns += blk["attns"]
            if blk["down"]:
                w[blk["down"] + ".w"] = self._pack_conv(sd[blk["down"] + ".weight"])
                w[blk["down"] + ".b"] = self._f32(sd[blk["down"] + ".bias"])
        resnets += g.mid["resnets"]
        attns.append(g.mid["attn"])
        for blk in g.up:
            resnets += blk["resnets"]
            attns += blk["attns"]
            if blk["up"]:
                w[blk["up"] + ".w"] = self._pack_conv(sd[blk["up"] + ".weight"])
                w[blk["up"] + ".b"] = self._f32(sd[blk["up"] + ".bias"])
        tw, tb, off = [], [], 0
        for r in resnets:
            self._pack_resnet(sd, r.name, True, split_at=(r.cin - r.skip_channels) if getattr(r, "skip_channels", 0) else 0)
            self.tproj_off[r.name] = off
            tw.append(sd[f"{r.name}.time_emb_proj.weight"])
            tb.append(sd[f"{r.name}.time_emb_proj.bias"])
            off += r.cout
        self.tproj_total = off
        w["tproj.w"] = self._f32(torch.cat(tw, dim=0))
        w["tproj.b"] = self._f32(torch.cat(tb, dim=0))
        for a in attns:
            n, c = a.name, a.channels
            b = f"{n}.transformer_blocks.0"
            w[f"{n}.norm.g"] = self._f32(sd[f"{n}.norm.weight"])
            w[f"{n}.norm.b"] = self._f32(sd[f"{n}.norm.bias"])
            for pj in ("proj_in", "proj_out"):
                w[f"{n}.{pj}.w"] = self._pack_mat(sd[f"{n}.{pj}.weight"])
                w[f"{n}.{pj}.b"] = self._f32(sd[f"{n}.{pj}.bias"])
            for i in ("1", "2", "3"):
                w[f"{n}.ln{i}.g"] = self._f32(sd[f"{b}.norm{i}.weight"])
                w[f"{n}.ln{i}.b"] = self._f32(sd[f"{b}.norm{i}.bias"])
            for attn in ("attn1", "attn2"):
                for t in S.LORA_TARGETS:
                    self.master[f"{b}.{attn}.{t}"] = self._f32(sd[f"{b}.{attn}.{t}.weight"])
            for key, shp in ((f"{n}.qkv.w", (3 * c, c)), (f"{n}.o1.w", (c, c)), (f"{n}.q2.w", (c, c)),
                             (f"{n}.kv2.w", (2 * c, self.ucfg.cross_attention_dim)), (f"{n}.o2.w", (c, c)),
                             (f"{n}.qkv.wln", (3 * c, c)), (f"{n}.q2.wln", (c, c))):
                self.w_rows[key] = torch.empty(shp, dtype=self.tdt, device=self.device)
                w[key] = self._empty_tiled(*shp) if self._w_tiled else self.w_rows[key]
            w[f"{n}.o1.b"] = self._f32(sd[f"{b}.attn1.to_out.0.bias"])
            w[f"{n}.o2.b"] = self._f32(sd[f"{b}.attn2.to_out.0.bias"])
            w[f"{n}.ff1.w"] = self._pack_mat(sd[f"{b}.ff.net.0.proj.weight"], geglu=True)
            w[f"{n}.ff1.b"] = self._f32(sd[f"{b}.ff.net.0.proj.bias"][self._geglu_perm(8 * c)])
            # folded-LayerNorm operands (idb_gemm_desc.ln_*): W' = W * gamma along K (one rounding), u = row sums of the ROUNDED W',
            # v = W beta (+ the layer's bias: idb_gemm takes no bias with ln_stats) in fp32; the LoRA-affected ones (qkv, q2) are
            # (re)built by set_lora.  All of it through the C ABI (idb_pack_matrix_scaled, idb_ln_fold_vectors): no torch arithmetic
            wf = self._f32(sd[f"{b}.ff.net.0.proj.weight"])
            w[f"{n}.ff1.wln"] = torch.empty((8 * c, c), dtype=self.tdt, device=self.device)
            L.check(self.lib.idb_pack_matrix_scaled(wf.data_ptr(), w[f"{n}.ff1.wln"].data_ptr(), 8 * c, c, 1, w[f"{n}.ln3.g"].data_ptr(), self.dt,
                                                    _stream()), "idb_pack_matrix_scaled")
            w[f"{n}.ff1.u"] = torch.empty((8 * c,), dtype=torch.float32, device=self.device)
            w[f"{n}.ff1.v"] = torch.empty((8 * c,), dtype=torch.float32, device=self.device)
            L.check(self.lib.idb_ln_fold_vectors(wf.data_ptr(), None, None, 0, 0.0, w[f"{n}.ff1.wln"].data_ptr(), w[f"{n}.ln3.b"].data_ptr(),
                                                 w[f"{n}.ff1.b"].data_ptr(), w[f"{n}.ff1.u"].data_ptr(), w[f"{n}.ff1.v"].data_ptr(), 8 * c, c, 1,
                                                 self.dt, _stream()), "idb_ln_fold_vectors")
            del wf
            w[f"{n}.qkv.v"] = torch.empty((3 * c,), dtype=torch.float32, device=self.device)
            w[f"{n}.q2.v"] = torch.empty((c,), dtype=torch.float32, device=self.device)
            w[f"{n}.qkv.u"] = torch.empty((3 * c,), dtype=torch.float32, device=self.device)     # filled IN PLACE by set_lora: captured
            w[f"{n}.q2.u"] = torch.empty((c,), dtype=torch.float32, device=self.device)          # graphs hold these addresses
            w[f"{n}.ff2.w"] = self._pack_mat(sd[f"{b}.ff.net.2.weight"])
            w[f"{n}.ff2.b"] = self._f32(sd[f"{b}.ff.net.2.bias"])
        w["conv_norm_out.g"] = self._f32(sd["conv_norm_out.weight"])
        w["conv_norm_out.b"] = self._f32(sd["conv_norm_out.bias"])
        w["conv_out.w"] = self._pack_conv(sd["conv_out.weight"])
        w["conv_out.b"] = self._f32(sd["conv_out.bias"])
        self._attn_specs = attns
        self._resnet_specs = resnets
        self._tile_static_weights()
        self.set_lora(None)

    def _attn_dst(self, a: S.AttnSpec, attn: str, t: str) -> Tuple[torch.Tensor, int]:
        """(packed matrix, row offset) that holds projection `t` of `attn`."""
        n, c = a.name, a.channels
        if attn == "attn1":
            if t == "to_out.0":
                return self.w_rows[f"{n}.o1.w"], 0
            return self.w_rows[f"{n}.qkv.w"], {"to_q": 0, "to_k": c, "to_v": 2 * c}[t]
        if t == "to_q":
            return self.w_rows[f"{n}.q2.w"], 0
        if t == "to_out.0":
            return self.w_rows[f"{n}.o2.w"], 0
        return self.w_rows[f"{n}.kv2.w"], {"to_k": 0, "to_v": c}[t]

    def set_lora(self, lora: Optional[SD], scale: float = 1.0, alphas: Optional[Dict[str, float]] = None) -> None:
        """(Re)build the 128 LoRA-affected matrices: W' = W + scale*(alpha/r) * B A in fp32, then one
        rounding to the operand dtype (peft merged form; inference_ID-Booth.py:107).  ``lora`` uses
        normalized keys ``<module>.lora_A/lora_B.weight``; None restores the base weights."""
        used = 0
        for a in self._attn_specs:
            b = f"{a.name}.transformer_blocks.0"
            for attn in ("attn1", "attn2"):
                for t in S.LORA_TARGETS:
                    key = f"{b}.{attn}.{t}"
                    mat, row0 = self._attn_dst(a, attn, t)
                    master = self.master[key]
                    rows, cols = master.shape
                    dst_ptr = mat.data_ptr() + row0 * mat.shape[1] * 2
                    assert mat.shape[1] == cols
                    la = None if lora is None else lora.get(key + ".lora_A.weight")
                    # folded-LayerNorm copy of the projections that read a LayerNorm output: attn1 to_q/k/v (norm1), attn2 to_q (norm2)
                    fold = None
                    if attn == "attn1" and t != "to_out.0":
                        fold = (self.w_rows[f"{a.name}.qkv.wln"], self.w[f"{a.name}.qkv.v"], row0, self.w[f"{a.name}.ln1.g"], self.w[f"{a.name}.ln1.b"],
                                self.w[f"{a.name}.qkv.u"])
                    elif attn == "attn2" and t == "to_q":
                        fold = (self.w_rows[f"{a.name}.q2.wln"], self.w[f"{a.name}.q2.v"], 0, self.w[f"{a.name}.ln2.g"], self.w[f"{a.name}.ln2.b"],
                                self.w[f"{a.name}.q2.u"])
                    if la is None:
                        L.check(self.lib.idb_pack_matrix(master.data_ptr(), dst_ptr, rows, cols, 0, self.dt, _stream()),
                                "idb_pack_matrix")
                        if fold is not None:
                            fmat, fv, fr0, gam, bet, fu = fold
                            L.check(self.lib.idb_lora_merge_scaled(master.data_ptr(), None, None, fmat.data_ptr() + fr0 * cols * 2, rows, cols, 0, 0.0,
                                                                   gam.data_ptr(), self.dt, _stream()), "idb_lora_merge_scaled")
                            L.check(self.lib.idb_ln_fold_vectors(master.data_ptr(), None, None, 0, 0.0, fmat.data_ptr() + fr0 * cols * 2, bet.data_ptr(),
                                                                 None, fu.data_ptr() + 4 * fr0, fv.data_ptr() + 4 * fr0, rows, cols, 0, self.dt,
                                                                 _stream()), "idb_ln_fold_vectors")
                        continue
                    lb = lora[key + ".lora_B.weight"]
                    rank = la.shape[0]
                    if tuple(la.shape) != (rank, cols) or tuple(lb.shape) != (rows, rank):
                        raise ValueError(f"LoRA shapes for {key}: A {tuple(la.shape)} B {tuple(lb.shape)} do not match "
                                         f"W {rows}x{cols}")
                    alpha = (alphas or {}).get(key, float(rank))
                    la_d, lb_d = self._f32(la), self._f32(lb)
                    L.check(self.lib.idb_lora_merge(master.data_ptr(), la_d.data_ptr(), lb_d.data_ptr(), dst_ptr, rows, cols,
                                                    rank, float(scale * alpha / rank), self.dt, _stream()), "idb_lora_merge")
                    if fold is not None:
                        # u = row sums of the rounded folded operand (what the MFMA multiplies), v = (W + s B A) beta: written IN PLACE
                        # (captured graphs hold these addresses)
                        fmat, fv, fr0, gam, bet, fu = fold
                        sc = float(scale * alpha / rank)
                        L.check(self.lib.idb_lora_merge_scaled(master.data_ptr(), la_d.data_ptr(), lb_d.data_ptr(), fmat.data_ptr() + fr0 * cols * 2,
                                                               rows, cols, rank, sc, gam.data_ptr(), self.dt, _stream()), "idb_lora_merge_scaled")
                        L.check(self.lib.idb_ln_fold_vectors(master.data_ptr(), la_d.data_ptr(), lb_d.data_ptr(), rank, sc, fmat.data_ptr() + fr0 * cols * 2,
                                                             bet.data_ptr(), None, fu.data_ptr() + 4 * fr0, fv.data_ptr() + 4 * fr0, rows, cols, 0,
                                                             self.dt, _stream()), "idb_ln_fold_vectors")
                    used += 1
        if self._w_tiled:                                # the forward reads the tiled copies: refill them IN PLACE (captured graphs)
            for key, rows_t in self.w_rows.items():
                self.tile_weight(rows_t, out=self.w[key])
        if lora is not None:
            n_pairs = sum(1 for k in lora if k.endswith(".lora_A.weight"))
            if used != n_pairs:
                raise ValueError(f"LoRA file has {n_pairs} adapter pairs but {used} matched UNet attention projections")
        self.lora_loaded = lora is not None      # stream-ordered: the next launch on this stream sees the merged weights

    def set_lora_groups(self, loras: Sequence[Optional[SD]], scale: float = 1.0, alphas: Optional[Dict[str, float]] = None) -> None:
        """A mixed-identity batch in ONE call (BASELINE configs[2]: 8 identities x 8 prompts): group g of the batch — samples
        [g*B/G, (g+1)*B/G) — uses the merged weights of ``loras[g]``.  Each group's 128 matrices (and folded-LayerNorm vectors) are
        built by ``set_lora`` and copied into slot g of grouped buffers; the GEMMs then select the matrix per row tile
        (idb_gemm_desc.w_groups): the arithmetic is exactly the merged form of every identity, no epilogue work, no extra pass over x."""
        G = len(loras)
        if G <= 1:
            self.groups = 1
            self.set_lora(loras[0] if G else None, scale, alphas)
            return
        vec_keys = [k for a in self._attn_specs for k in (f"{a.name}.qkv.u", f"{a.name}.qkv.v", f"{a.name}.q2.u", f"{a.name}.q2.v")]
        if not hasattr(self, "_wg_by_G"):
            self._wg_by_G = {}                           # one buffer set per group count: captured graphs keep reading theirs
        if G in self._wg_by_G:
            self.wg = self._wg_by_G[G]
        else:
            self.wg = self._wg_by_G[G] = {}
            for key in self.w_rows:
                t = torch.empty((G, self.w[key].numel()), dtype=self.tdt, device=self.device)
                t._groups, t._gshape = G, tuple(self.w_rows[key].shape)
                if self._w_tiled:
                    t._tiled = t._gshape
                self.wg[key] = t
            for key in vec_keys:
                t = torch.empty((G, self.w[key].numel()), dtype=torch.float32, device=self.device)
                t._groups = G
                self.wg[key] = t
        for g, lora in enumerate(loras):
            self.set_lora(lora, scale, alphas)           # stream-ordered: fills self.w[...] / u / v for this identity
            for key in self.w_rows:
                self.wg[key][g].copy_(self.w[key].reshape(-1))
            for key in vec_keys:
                self.wg[key][g].copy_(self.w[key])
        self.groups = G

    def _Wl(self, key: str) -> torch.Tensor:
        """LoRA-affected operand `key`: the grouped buffer while set_lora_groups is active, else the single merged one."""
        return self.wg[key] if self.groups > 1 else self.w[key]

    def _pack_vae(self, sd: SD) -> None:
        w, g = self.w, self.vgraph
        w["v.pq.w"] = self._f32(sd["post_quant_conv.weight"].reshape(self.vcfg.latent_channels, -1))
        w["v.pq.b"] = self._f32(sd["post_quant_conv.bias"])
        w["v.conv_in.w"] = self._f32(sd["decoder.conv_in.weight"])
        w["v.conv_in.b"] = self._f32(sd["decoder.conv_in.bias"])
        for nm in ("decoder.mid_block.resnets.0", "decoder.mid_block.resnets.1"):
            self._pack_resnet(sd, nm, False)
        a = "decoder.mid_block.attentions.0"
        w[f"{a}.gn.g"] = self._f32(sd[f"{a}.group_norm.weight"])
        w[f"{a}.gn.b"] = self._f32(sd[f"{a}.group_norm.bias"])
        for t, short in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "o")):
            w[f"{a}.{short}.w"] = self._pack_mat(sd[f"{a}.{t}.weight"])
            w[f"{a}.{short}.b"] = self._f32(sd[f"{a}.{t}.bias"])
        for blk in g.up:
            for name, _, _ in blk["resnets"]:
                self._pack_resnet(sd, name, False)
            if blk["up"]:
                w[blk["up"] + ".w"] = self._pack_conv(sd[blk["up"] + ".weight"])
                w[blk["up"] + ".b"] = self._f32(sd[blk["up"] + ".bias"])
        w["v.norm_out.g"] = self._f32(sd["decoder.conv_norm_out.weight"])
        w["v.norm_out.b"] = self._f32(sd["decoder.conv_norm_out.bias"])
        w["v.conv_out.w"] = self._pack_conv(sd["decoder.conv_out.weight"])
        w["v.conv_out.b"] = self._f32(sd["decoder.conv_out.bias"])
        self._tile_static_weights()

    def pack_vae_encoder(self, sd: SD) -> None:
        """AutoencoderKL encoder weights (train_ID-Booth.py:1001): packed on first use — the sampling path never needs them.
        quant_conv (1x1, 8 -> 8) is folded into conv_out in fp32 (a 1x1 conv after a 3x3 conv is a linear map of its output
        channels): W'[o] = sum_j Wq[o][j] W[j], b' = Wq b + bq, rounded to the operand dtype once."""
        w, cfg = self.w, self.vcfg
        w["ve.conv_in.w"] = self._f32(sd["encoder.conv_in.weight"])
        w["ve.conv_in.b"] = self._f32(sd["encoder.conv_in.bias"])
        for blk in S.vae_encoder_blocks(cfg):
            for name, _, _ in blk["resnets"]:
                self._pack_resnet(sd, name, False)
            if blk["down"]:
                w[blk["down"] + ".w"] = self._pack_conv(sd[blk["down"] + ".weight"])
                w[blk["down"] + ".b"] = self._f32(sd[blk["down"] + ".bias"])
        for nm in ("encoder.mid_block.resnets.0", "encoder.mid_block.resnets.1"):
            self._pack_resnet(sd, nm, False)
        a = "encoder.mid_block.attentions.0"
        w[f"{a}.gn.g"] = self._f32(sd[f"{a}.group_norm.weight"])
        w[f"{a}.gn.b"] = self._f32(sd[f"{a}.group_norm.bias"])
        for t, short in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "o")):
            w[f"{a}.{short}.w"] = self._pack_mat(sd[f"{a}.{t}.weight"])
            w[f"{a}.{short}.b"] = self._f32(sd[f"{a}.{t}.bias"])
        w["ve.norm_out.g"] = self._f32(sd["encoder.conv_norm_out.weight"])
        w["ve.norm_out.b"] = self._f32(sd["encoder.conv_norm_out.bias"])
        wq = sd["quant_conv.weight"].double().reshape(2 * cfg.latent_channels, 2 * cfg.latent_channels)
        co = sd["encoder.conv_out.weight"].double()
        folded = torch.einsum("oj,jikl->oikl", wq, co).float()
        fb = (wq @ sd["encoder.conv_out.bias"].double() + sd["quant_conv.bias"].double()).float()
        w["ve.conv_out.w"] = self._pack_conv(folded)
        w["ve.conv_out.b"] = self._f32(fb)
        self._tile_static_weights()
        self.has_vae_encoder = True
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------
    # kernel wrappers
    # ------------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if nbytes > self._ws.numel():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("split-K workspace would grow during graph capture; run one eager warm-up first")
            self._retired.append(self._ws)      # captured graphs hold the old address: keep it alive (and private to them)
            self._ws = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=self.device)
        return self._ws

    def gemm(self, srcs, w: torch.Tensor, n: int, batch: int, oh: int, ow: int, bias=None, sbias=None,
             residual=None, geglu=False, stride=1, out_f32=False, out_scale=0.0, split_k=0, tile=0,
             out: Optional[torch.Tensor] = None, flags: int = 0, act: int = 0, pad_mode: int = 0, gn_stats: int = 0,
             gn_stats_always: bool = False, row_stats: bool = False, ln=None, gn_in=None) -> torch.Tensor:
        """srcs: list of (tensor, channels, taps, in_h, in_w, upsample); sbias: (tensor, elem_offset, ld).
        row_stats: also emit the per-row partial sums a folded LayerNorm of the output needs (``out._rs = (buffer, tiles)``) when the
        plan can; ln = (stats, tiles, u, v, eps): A holds raw rows, w the gamma-scaled weights (idb_gemm_desc.ln_*)."""
        m = batch * oh * ow
        ncols = n // 2 if geglu else n
        own_out = out is None
        if out is None:
            out = self.arena.alloc((m, ncols), torch.float32 if out_f32 else self.tdt)
        d = L.GemmDesc()
        d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = self.dt, batch, oh, ow, stride, n, len(srcs)
        for i, (t, ch, taps, ih, iw, up) in enumerate(srcs):
            d.src[i].ptr, d.src[i].channels, d.src[i].taps = t.data_ptr(), ch, taps
            d.src[i].in_h, d.src[i].in_w, d.src[i].upsample = ih, iw, up
        d.w, d.bias = w.data_ptr(), _ptr(bias)
        if sbias is not None:
            d.sample_bias = sbias[0].data_ptr() + 4 * sbias[1]
            d.sample_bias_ld = sbias[2]
        d.residual, d.geglu = _ptr(residual), int(geglu)
        d.out, d.out_dtype, d.out_ld = out.data_ptr(), (L.IDB_F32 if out_f32 else self.dt), out.shape[-1]
        d.split_k, d.tile, d.out_scale, d.flags, d.act = split_k, tile, out_scale, flags, act
        d.pad_mode = pad_mode
        if gn_in is not None:       # (partials, chunks, groups, eps, gamma, beta, silu, nsrc): GroupNorm(+SiLU) of the first nsrc sources in-kernel
            d.gn_in_partials, d.gn_in_chunks, d.gn_in_groups, d.gn_in_eps = gn_in[0].data_ptr(), gn_in[1], gn_in[2], gn_in[3]
            d.gn_in_gamma, d.gn_in_beta, d.gn_in_silu, d.gn_in_nsrc = gn_in[4].data_ptr(), gn_in[5].data_ptr(), int(gn_in[6]), gn_in[7]
        d.w_layout = 1 if getattr(w, "_tiled", None) is not None else 0
        G = getattr(w, "_groups", 0) or 0
        if G > 1:                                      # grouped weights: rows [r*M/rep + g*rpg, ... + rpg) of every CFG half r use matrix g
            rep = self._rep
            if m % (rep * G):
                raise ValueError(f"grouped weights: {m} rows do not divide into {rep} x {G} groups")
            d.w_groups, d.w_group_rows, d.w_group_stride = G, m // (rep * G), w.shape[1] * 2
            tile_id = C.c_int32()
            L.check(self.lib.idb_gemm_plan(C.byref(d), C.byref(tile_id), None, None), "idb_gemm_plan")
            bm = {1: 128, 2: 128, 3: 64, 4: 64, 5: 128, 6: 64, 7: 64, 8: 128, 9: 128}[tile_id.value % 10] * (2 if tile_id.value // 10 >= 8 else 1)
            if tile_id.value // 10 == 4 or d.w_group_rows % bm:
                # this launch will run group by group on row slices (_gemm_per_group): no row statistics out, no folded LayerNorm in
                row_stats = False
                if ln is not None:
                    if own_out:
                        self.arena.free(out)
                    return None
        rs_buf = None
        if row_stats and self._ln_fold:
            nt = self.lib.idb_gemm_row_stats_tiles(C.byref(d))
            if nt > 0:
                rs_buf = self.arena.alloc((m * nt * 2,), torch.float32)
                d.row_stats_out = rs_buf.data_ptr()
        if ln is not None:
            if self.lib.idb_gemm_folds_layernorm(C.byref(d)) == 0:      # this plan cannot fold (split-K): caller keeps idb_layernorm
                if own_out:
                    self.arena.free(out)
                if rs_buf is not None:
                    self.arena.free(rs_buf)
                return None
            d.ln_stats, d.ln_tiles, d.ln_u, d.ln_v, d.ln_eps = ln[0].data_ptr(), ln[1], ln[2].data_ptr(), ln[3].data_ptr(), ln[4]
        gn_part = None
        if gn_stats and self._gn_fuse and (oh * ow) % 64 == 0 and oh * ow <= 4096 and not geglu and not out_f32 and n % gn_stats == 0:
            # the GroupNorm that consumes `out` next gets its first pass from this GEMM: from its split-K reduce launch, or — without a
            # split — from its own LDS-staged epilogue when the column tiles hold whole groups (idb_kernels.h); otherwise the library
            # would add a statistics launch and the ordinary two-pass GroupNorm is at least as good
            mode = self.lib.idb_gemm_emits_gn_partials(C.byref(d), gn_stats)     # 1: split-K reduce launch, 2: the GEMM's own epilogue
            if mode == 2 and not self._gn_epi:
                mode = 0
            if mode > 0 or gn_stats_always:      # gn_stats_always: tests of the library's extra-statistics-launch path
                gn_part = self.arena.alloc((batch * (oh * ow // 64) * gn_stats * 2,), torch.float32)
                d.gn_partials, d.gn_groups = gn_part.data_ptr(), gn_stats
        d.counters, d.counters_len = self._counters.data_ptr(), self._counters.numel()
        need = self.lib.idb_gemm_workspace_bytes(C.byref(d))
        ws = self._workspace(need) if need else None
        log = self.launch_log
        if log is not None:                      # bench.py's per-kernel roofline accounting (eager pass only)
            tile, sk, blocks = C.c_int32(), C.c_int32(), C.c_int32()
            L.check(self.lib.idb_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk), C.byref(blocks)), "idb_gemm_plan")
            k_total = sum(ch * taps for (_, ch, taps, _, _, _) in srcs)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = self.lib.idb_gemm(C.byref(d), _ptr(ws), need, _stream())
        if rc == -2 and G > 1:
            # the plan's tile height does not divide the group's rows (cross-attention K/V of 77 tokens, tiny grids): one launch per
            # (CFG half, group) on row slices — the same arithmetic
            self._gemm_per_group(d, G, m // (self._rep * G), w, ws, need, ln)
        else:
            L.check(rc, "idb_gemm")
        if gn_part is not None:
            out._gn = (gn_part, oh * ow // 64, gn_stats)
        if rs_buf is not None:
            out._rs = (rs_buf, nt)
        if log is not None:
            ev1.record()
            log.append({"tile": tile.value + (1000 if gn_in is not None else 0), "split_k": sk.value, "blocks": blocks.value, "m": m, "n": n, "k": k_total,
                        "flops": 2.0 * m * n * k_total, "ev": (ev0, ev1), "desc": d, "ws": (ws, need),
                        "bytes": 2.0 * (m * k_total / (9 if srcs[0][2] == 9 else 1) + n * k_total + m * ncols)})
        return out

    def _gemm_per_group(self, d, G: int, rpg: int, w: torch.Tensor, ws, need: int, ln) -> None:
        """Fallback of the grouped-weights GEMM for plain [M][K] sources: row slice [j*rpg, (j+1)*rpg) with matrix j % G."""
        if d.nsrc != 1 or d.src[0].taps != 1 or d.src[0].in_h != 1 or d.gn_partials or d.row_stats_out:
            raise L.IdbError("grouped weights: per-group fallback needs one plain [M][K] source without statistics outputs")
        m, k, esz = d.batch, d.src[0].channels, 2
        osz = 4 if d.out_dtype == L.IDB_F32 else 2
        base = {"a": d.src[0].ptr, "out": d.out, "res": d.residual, "w": d.w, "u": d.ln_u, "v": d.ln_v, "st": d.ln_stats}
        for j in range(m // rpg):
            g = j % G
            d.batch = rpg
            d.src[0].ptr = base["a"] + j * rpg * k * esz
            d.out = base["out"] + j * rpg * d.out_ld * osz
            d.residual = None if not base["res"] else base["res"] + j * rpg * d.out_ld * esz
            d.w = base["w"] + g * d.w_group_stride
            if ln is not None:
                d.ln_u, d.ln_v = base["u"] + g * d.n * 4, base["v"] + g * d.n * 4
                d.ln_stats = base["st"] + j * rpg * d.ln_tiles * 8
            d.w_groups = 0
            need_j = self.lib.idb_gemm_workspace_bytes(C.byref(d))
            wsj = self._workspace(need_j) if need_j else None
            L.check(self.lib.idb_gemm(C.byref(d), _ptr(wsj), need_j, _stream()), "idb_gemm (per group)")

    def fuses_groupnorm(self, src_shapes, w: torch.Tensor, n: int, batch: int, oh: int, ow: int, groups: int, nsrc_norm: int) -> bool:
        """Would idb_gemm apply the GroupNorm of the first `nsrc_norm` sources inside the kernel for this shape?  src_shapes:
        [(channels, taps)].  Host-only (plan query), cached per shape."""
        if not self._gn_conv:
            return False
        key = (tuple(src_shapes), n, batch, oh, ow, groups, nsrc_norm, getattr(w, "_tiled", None) is not None)
        hit = self._gn_conv_cache.get(key)
        if hit is None:
            d = L.GemmDesc()
            dummy = self._gn_ws.data_ptr()
            d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = self.dt, batch, oh, ow, 1, n, len(src_shapes)
            for i, (ch, taps) in enumerate(src_shapes):
                d.src[i].ptr, d.src[i].channels, d.src[i].taps, d.src[i].in_h, d.src[i].in_w = dummy, ch, taps, oh, ow
            d.w, d.out, d.out_dtype, d.out_ld = dummy, dummy, self.dt, n
            d.w_layout = 1 if getattr(w, "_tiled", None) is not None else 0
            d.gn_in_partials, d.gn_in_chunks, d.gn_in_groups, d.gn_in_eps, d.gn_in_nsrc = dummy, 1, groups, 1e-5, nsrc_norm
            d.gn_in_gamma = d.gn_in_beta = dummy
            hit = self._gn_conv_cache[key] = self.lib.idb_gemm_fuses_groupnorm(C.byref(d)) > 0
        return hit

    def linear(self, x: torch.Tensor, w: torch.Tensor, n: int, k: int, **kw) -> torch.Tensor:
        m = x.numel() // k
        return self.gemm([(x, k, 1, 1, 1, 0)], w, n, m, 1, 1, **kw)

    def groupnorm(self, x0, c0, x1, c1, batch, hw, gamma, beta, eps, silu, groups=None) -> torch.Tensor:
        groups = groups or self.ucfg.norm_num_groups
        out = self.arena.alloc((batch * hw, c0 + c1), self.tdt)
        need = self.lib.idb_groupnorm_workspace_bytes(batch, hw, groups)
        if need > self._gn_ws.numel():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("GroupNorm workspace would grow during graph capture")
            self._retired.append(self._gn_ws)   # as _workspace: older graphs keep writing their partials here
            self._gn_ws = torch.empty(need * 2, dtype=torch.uint8, device=self.device)
        pin, pin_chunks = None, 0
        st = getattr(x0, "_gn", None)
        if st is not None:
            x0._gn = None
            if x1 is None and st[1] * 64 == hw and st[2] == groups:
                pin, pin_chunks = st[0], st[1]           # statistics produced by the GEMM that wrote x0
        L.check(self.lib.idb_groupnorm(x0.data_ptr(), c0, _ptr(x1), c1, batch, hw, groups, eps, gamma.data_ptr(),
                                       beta.data_ptr(), int(silu), out.data_ptr(), self.dt, self._gn_ws.data_ptr(),
                                       self._gn_ws.numel(), _ptr(self._gn_sync), 0 if self._gn_sync is None else self._gn_sync.numel(),
                                       _ptr(pin), pin_chunks, _stream()), "idb_groupnorm")
        if st is not None:
            self.arena.free(st[0])
        return out

    X8_SIGMAS = 8.0

    @staticmethod
    def fp8_act_scale(gamma: torch.Tensor, beta: torch.Tensor) -> float:
        """Scale of an e4m3 GroupNorm(+SiLU) output, derived per layer at weight load: y = silu(gamma * xhat + beta) with |y| <= |gamma| |xhat|
        + |beta|, so max|gamma| * X8_SIGMAS + max|beta| maps to the largest finite e4m3 value (448) and only |xhat| > 8 sigma could
        saturate (the kernel clamps).  e4m3 is a floating format: a generous range costs nothing but the sub-normal end
        (scale * 2^-9 ~ 1e-4 here).  Round 2 used one fixed 8 / 448 for every layer, valid only for gamma <= 1."""
        return float(HipEngine.X8_SIGMAS * gamma.abs().max().item() + beta.abs().max().item()) / 448.0

    def groupnorm_fp8(self, x0, c0, x1, c1, batch, hw, gamma, beta, eps, silu, x_scale: float, groups=None) -> torch.Tensor:
        groups = groups or self.ucfg.norm_num_groups
        out = self.arena.alloc((batch * hw, c0 + c1), torch.uint8)
        need = self.lib.idb_groupnorm_workspace_bytes(batch, hw, groups)
        if need > self._gn_ws.numel():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("GroupNorm workspace would grow during graph capture")
            self._retired.append(self._gn_ws)
            self._gn_ws = torch.empty(need * 2, dtype=torch.uint8, device=self.device)
        pin, pin_chunks = None, 0
        st = getattr(x0, "_gn", None)
        if st is not None:
            x0._gn = None
            if x1 is None and st[1] * 64 == hw and st[2] == groups:
                pin, pin_chunks = st[0], st[1]
        L.check(self.lib.idb_groupnorm_fp8(x0.data_ptr(), c0, _ptr(x1), c1, batch, hw, groups, eps, gamma.data_ptr(), beta.data_ptr(), int(silu),
                                           out.data_ptr(), 1.0 / x_scale, self.dt, self._gn_ws.data_ptr(), self._gn_ws.numel(), _ptr(pin),
                                           pin_chunks, _stream()), "idb_groupnorm_fp8")
        if st is not None:
            self.arena.free(st[0])
        return out

    def gn_statistics(self, x0, c0, x1, c1, batch, hw, groups):
        """(partials [batch][chunks][groups][2] fp32, chunks) of GroupNorm(groups) over cat[x0, x1]: taken from the launch that
        produced x0 when it emitted them (``_gn``), else one idb_groupnorm_stats launch."""
        st = getattr(x0, "_gn", None)
        if st is not None:
            x0._gn = None
            if x1 is None and st[1] * 64 == hw and st[2] == groups:
                return st[0], st[1]
            self.arena.free(st[0])
        part = self.arena.alloc((batch * 64 * groups * 2,), torch.float32)
        chunks = C.c_int32()
        L.check(self.lib.idb_groupnorm_stats(x0.data_ptr(), c0, _ptr(x1), c1, batch, hw, groups, part.data_ptr(), part.numel() * 4,
                                             C.byref(chunks), self.dt, _stream()), "idb_groupnorm_stats")
        return part, chunks.value

    # ---- fp8 (e4m3) GEMM path: kernel level of BASELINE configs[4] (idb_gemm8.hip) -------------------------------------------
    def quantize_fp8(self, x: torch.Tensor, scale: float) -> torch.Tensor:
        """operand-dtype tensor -> uint8 storage of e4m3(x / scale) (saturating)."""
        out = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        L.check(self.lib.idb_quantize_fp8(x.data_ptr(), out.data_ptr(), x.numel(), 1.0 / float(scale), self.dt, _stream()), "idb_quantize_fp8")
        return out

    def pack_weight_fp8(self, w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """torch Conv2d [cout,cin,kh,kw] / Linear [out,in] fp32 -> (fp8 [cout][taps][cin padded to 128] as uint8, scales [cout] fp32)."""
        w4 = w if w.ndim == 4 else w[:, :, None, None]
        cout, cin, kh, kw = w4.shape
        src = self._f32(w4)
        cpad = (cin + 127) // 128 * 128
        dst = torch.empty((cout, kh * kw * cpad), dtype=torch.uint8, device=self.device)
        scales = torch.empty((cout,), dtype=torch.float32, device=self.device)
        L.check(self.lib.idb_pack_weight_fp8(src.data_ptr(), dst.data_ptr(), scales.data_ptr(), cout, cin, kh * kw, _stream()), "idb_pack_weight_fp8")
        return dst, scales

    def gemm_fp8(self, x8: torch.Tensor, x_scale: float, channels: int, taps: int, in_h: int, in_w: int, w8: torch.Tensor, w_scale: torch.Tensor,
                 n: int, batch: int, oh: int, ow: int, bias=None, sbias=None, residual=None, stride: int = 1, upsample: int = 0,
                 out: Optional[torch.Tensor] = None, gn_stats: int = 0) -> torch.Tensor:
        """gn_stats = G: also the first pass of the GroupNorm(G) that reads ``out`` next (``out._gn``), when the kernel's own epilogue can
        emit it (160-wide tiles holding whole groups; idb_gemm_fp8 would add a launch otherwise, no better than the two-pass GroupNorm)."""
        m = batch * oh * ow
        if out is None:
            out = self.arena.alloc((m, n), self.tdt)
        d = L.GemmFp8Desc()
        d.out_dtype, d.batch, d.out_h, d.out_w, d.stride, d.n = self.dt, batch, oh, ow, stride, n
        d.x, d.channels, d.taps, d.in_h, d.in_w, d.upsample, d.x_scale = x8.data_ptr(), channels, taps, in_h, in_w, upsample, float(x_scale)
        d.w, d.w_scale, d.bias = w8.data_ptr(), w_scale.data_ptr(), _ptr(bias)
        if sbias is not None:
            d.sample_bias = sbias[0].data_ptr() + 4 * sbias[1]
            d.sample_bias_ld = sbias[2]
        d.residual, d.out, d.out_ld = _ptr(residual), out.data_ptr(), out.shape[-1]
        gn_part = None
        if gn_stats and self._gn_fuse and self._gn_epi and n % 160 == 0 and n % gn_stats == 0 and 160 % (n // gn_stats) == 0 and \
                (n // gn_stats) % 2 == 0 and (oh * ow) % 64 == 0 and oh * ow <= 4096 and out.shape[-1] == n:
            gn_part = self.arena.alloc((batch * (oh * ow // 64) * gn_stats * 2,), torch.float32)
            d.gn_partials, d.gn_groups = gn_part.data_ptr(), gn_stats
        L.check(self.lib.idb_gemm_fp8(C.byref(d), _stream()), "idb_gemm_fp8")
        if gn_part is not None:
            out._gn = (gn_part, oh * ow // 64, gn_stats)
        return out

    def layernorm(self, x, rows, c, gamma, beta) -> torch.Tensor:
        out = self.arena.alloc((rows, c), self.tdt)
        L.check(self.lib.idb_layernorm(x.data_ptr(), out.data_ptr(), rows, c, 1e-5, gamma.data_ptr(), beta.data_ptr(),
                                       self.dt, _stream()), "idb_layernorm")
        return out

    def ln_linear(self, x, rows, c, pfx: str, ln_name: str, wname: str, n: int, **kw) -> torch.Tensor:
        """LayerNorm(x) @ W^T: folded into ONE GEMM when x carries row statistics (``_rs``) and the consumer's plan runs the LDS-staged
        epilogue; else idb_layernorm + the plain GEMM.  Weights: ``{pfx}.{wname}.w`` (plain), ``.wln`` / ``.u`` / ``.v`` (folded)."""
        W = self.w
        rs = getattr(x, "_rs", None)
        if rs is not None and self._ln_fold:
            kf = {k: v for k, v in kw.items() if k != "bias"}          # the bias is part of .v
            lw = self._Wl if wname in ("qkv", "q2") else W.__getitem__
            out = self.linear(x, lw(f"{pfx}.{wname}.wln"), n, c, ln=(rs[0], rs[1], lw(f"{pfx}.{wname}.u"), lw(f"{pfx}.{wname}.v"), 1e-5), **kf)
            if out is not None:
                return out
        t = self.layernorm(x, rows, c, W[f"{pfx}.{ln_name}.g"], W[f"{pfx}.{ln_name}.b"])
        out = self.linear(t, (self._Wl if wname in ("qkv", "q2") else W.__getitem__)(f"{pfx}.{wname}.w"), n, c, **kw)
        self.arena.free(t)
        return out

    def folds(self, x, rows: int, c: int, n: int, geglu: bool = False) -> bool:
        """Will ``ln_linear`` fold the LayerNorm of ``x`` [rows][c] into its [n][c] projection?  (The consumer's plan must run the
        LDS-staged epilogue or the persistent variant; a split-K plan does not.)  Asked BEFORE the producer GEMM so that it emits
        row statistics only for a consumer that reads them.  Cached per shape."""
        if not self._ln_fold:
            return False
        key = (rows, c, n, geglu)
        hit = self._fold_cache.get(key)
        if hit is None:
            d = L.GemmDesc()
            d.dtype, d.batch, d.out_h, d.out_w, d.stride, d.n, d.nsrc = self.dt, rows, 1, 1, 1, n, 1
            d.src[0].ptr, d.src[0].channels, d.src[0].taps, d.src[0].in_h, d.src[0].in_w = x.data_ptr(), c, 1, 1, 1
            d.w, d.geglu = x.data_ptr(), int(geglu)
            d.out, d.out_dtype, d.out_ld = x.data_ptr(), self.dt, (n // 2 if geglu else n)
            hit = self._fold_cache[key] = self.lib.idb_gemm_folds_layernorm(C.byref(d)) > 0
        return hit

    def _free_rs(self, x) -> None:
        rs = getattr(x, "_rs", None)
        if rs is not None:
            x._rs = None
            self.arena.free(rs[0])

    def attention(self, q, q_ld, k_ptr, v_ptr, kv_ld, batch, heads, n_q, n_kv, n_kv_alloc, causal: bool = False) -> torch.Tensor:
        out = self.arena.alloc((batch * n_q, heads * 64), self.tdt)
        L.check(self.lib.idb_attention(q.data_ptr(), q_ld, k_ptr, v_ptr, kv_ld, out.data_ptr(), heads * 64, batch, heads,
                                       n_q, n_kv, n_kv_alloc, 0.125, int(causal), self.dt, _stream()), "idb_attention")
        return out

    def _free(self, t: Optional[torch.Tensor]) -> None:
        if t is not None and t.data_ptr() not in self._pinned:
            self.arena.free(t)

    def _tap(self, name: str, t: torch.Tensor) -> torch.Tensor:
        if self.taps is not None:
            self.taps[name] = t.float()          # a copy outside the arena
        return t

    # ------------------------------------------------------------------------------------
    # UNet
    # ------------------------------------------------------------------------------------
    def _resnet(self, name, xa, ca, xb, cb, cout, batch, h, w_, sbias, eps, groups=None, out_stats: bool = False) -> torch.Tensor:
        W = self.w
        cin = ca + cb
        G0 = groups or self.ucfg.norm_num_groups
        short = f"{name}.has_shortcut" in W
        if self.fp8 and f"{name}.conv1.w8" in W:
            # fp8 MFMA path: GroupNorm+SiLU -> e4m3, conv on v_mfma_scale_f32_16x16x128_f8f6f4; the 1x1 shortcut over the raw inputs
            # stays an f16 GEMM whose result enters the second conv as its residual
            sb = None if sbias is None else (sbias[0], sbias[1] + self.tproj_off[name], sbias[2])
            s1, s2 = self.x8_scale[f"{name}.gn1"], self.x8_scale[f"{name}.gn2"]
            n1 = self.groupnorm_fp8(xa, ca, xb, cb, batch, h * w_, W[f"{name}.gn1.g"], W[f"{name}.gn1.b"], eps, True, s1, groups)
            h1 = self.gemm_fp8(n1, s1, cin, 9, h, w_, W[f"{name}.conv1.w8"], W[f"{name}.conv1.s8"], cout, batch, h, w_,
                               bias=W[f"{name}.conv1.b"], sbias=sb, gn_stats=G0)
            self.arena.free(n1)
            n2 = self.groupnorm_fp8(h1, cout, None, 0, batch, h * w_, W[f"{name}.gn2.g"], W[f"{name}.gn2.b"], eps, True, s2, groups)
            self.arena.free(h1)
            if short:
                srcs = [(xa, ca, 1, h, w_, 0)] + ([(xb, cb, 1, h, w_, 0)] if xb is not None else [])
                res = self.gemm(srcs, W[f"{name}.sc.w"], cout, batch, h, w_)
            else:
                res = xa
            out = self.gemm_fp8(n2, s2, cout, 9, h, w_, W[f"{name}.conv2.w8"], W[f"{name}.conv2.s8"], cout, batch, h, w_,
                                bias=W[f"{name}.conv2.b"], residual=res, gn_stats=G0 if out_stats else 0)
            self.arena.free(n2)
            if short:
                self.arena.free(res)
            return out
        sb = None if sbias is None else (sbias[0], sbias[1] + self.tproj_off[name], sbias[2])
        G = groups or self.ucfg.norm_num_groups
        hw = h * w_
        # ---- norm1 + SiLU + conv1: inside the conv (normalizer waves) where the plan allows, else idb_groupnorm + idb_gemm
        w1 = W.get(f"{name}.conv1.wsplit") if xb is not None else W[f"{name}.conv1.w"]
        shapes1 = [(ca, 9)] + ([(cb, 9)] if xb is not None else [])
        if w1 is not None and self._gn_conv_resnet and self.fuses_groupnorm(shapes1, w1, cout, batch, h, w_, G, len(shapes1)):
            part, chunks = self.gn_statistics(xa, ca, xb, cb, batch, hw, G)
            srcs1 = [(xa, ca, 9, h, w_, 0)] + ([(xb, cb, 9, h, w_, 0)] if xb is not None else [])
            h1 = self.gemm(srcs1, w1, cout, batch, h, w_, bias=W[f"{name}.conv1.b"], sbias=sb, gn_stats=G,
                           gn_in=(part, chunks, G, eps, W[f"{name}.gn1.g"], W[f"{name}.gn1.b"], True, len(srcs1)))
            self.arena.free(part)
        else:
            n1 = self.groupnorm(xa, ca, xb, cb, batch, hw, W[f"{name}.gn1.g"], W[f"{name}.gn1.b"], eps, True, groups)
            h1 = self.gemm([(n1, cin, 9, h, w_, 0)], W[f"{name}.conv1.w"], cout, batch, h, w_, bias=W[f"{name}.conv1.b"], sbias=sb,
                           gn_stats=G)                       # norm2 consumes h1 next
            self.arena.free(n1)
        # ---- norm2 + SiLU + conv2 (+ 1x1 shortcut over the raw inputs | + residual)
        short = f"{name}.has_shortcut" in W
        shapes2 = [(cout, 9)] + ([(ca, 1)] + ([(cb, 1)] if xb is not None else []) if short else [])
        if self._gn_conv_resnet and self.fuses_groupnorm(shapes2, W[f"{name}.conv2.w"], cout, batch, h, w_, G, 1):
            part, chunks = self.gn_statistics(h1, cout, None, 0, batch, hw, G)
            srcs = [(h1, cout, 9, h, w_, 0)]
            if short:
                srcs += [(xa, ca, 1, h, w_, 0)] + ([(xb, cb, 1, h, w_, 0)] if xb is not None else [])
            out = self.gemm(srcs, W[f"{name}.conv2.w"], cout, batch, h, w_, bias=W[f"{name}.conv2.b"], residual=None if short else xa,
                            gn_stats=G if out_stats else 0, gn_in=(part, chunks, G, eps, W[f"{name}.gn2.g"], W[f"{name}.gn2.b"], True, 1))
            self.arena.free(part)
            self.arena.free(h1)
            return out
        n2 = self.groupnorm(h1, cout, None, 0, batch, hw, W[f"{name}.gn2.g"], W[f"{name}.gn2.b"], eps, True, groups)
        self.arena.free(h1)
        if short:
            srcs = [(n2, cout, 9, h, w_, 0), (xa, ca, 1, h, w_, 0)]
            if xb is not None:
                srcs.append((xb, cb, 1, h, w_, 0))
            out = self.gemm(srcs, W[f"{name}.conv2.w"], cout, batch, h, w_, bias=W[f"{name}.conv2.b"], gn_stats=G if out_stats else 0)
        else:
            assert xb is None and ca == cout
            out = self.gemm([(n2, cout, 9, h, w_, 0)], W[f"{name}.conv2.w"], cout, batch, h, w_, bias=W[f"{name}.conv2.b"],
                            residual=xa, gn_stats=G if out_stats else 0)
        self.arena.free(n2)
        return out

    def _transformer(self, a: S.AttnSpec, x, batch, h, w_, kv, n_ctx, out_stats: bool = False) -> torch.Tensor:
        W, n, c = self.w, a.name, a.channels
        hw = h * w_
        m = batch * hw
        G = self.ucfg.norm_num_groups
        if c <= self._gn_conv_max_c and self.fuses_groupnorm([(c, 1)], W[f"{n}.proj_in.w"], c, batch, h, w_, G, 1):
            # Transformer2DModel.norm (no SiLU) inside proj_in
            part, chunks = self.gn_statistics(x, c, None, 0, batch, hw, G)
            h0 = self.gemm([(x, c, 1, h, w_, 0)], W[f"{n}.proj_in.w"], c, batch, h, w_, bias=W[f"{n}.proj_in.b"],
                           row_stats=self.folds(x, m, c, 3 * c), gn_in=(part, chunks, G, 1e-6, W[f"{n}.norm.g"], W[f"{n}.norm.b"], False, 1))
            self.arena.free(part)
        else:
            xn = self.groupnorm(x, c, None, 0, batch, hw, W[f"{n}.norm.g"], W[f"{n}.norm.b"], 1e-6, False)
            h0 = self.linear(xn, W[f"{n}.proj_in.w"], c, c, bias=W[f"{n}.proj_in.b"], row_stats=self.folds(xn, m, c, 3 * c))
            self.arena.free(xn)
        # self-attention
        qkv = self.ln_linear(h0, m, c, n, "ln1", "qkv", 3 * c)
        self._free_rs(h0)
        p = qkv.data_ptr()
        o = self.attention(qkv, 3 * c, p + 2 * c, p + 4 * c, 3 * c, batch, a.heads, hw, hw, hw)
        self.arena.free(qkv)
        h1 = self.linear(o, self._Wl(f"{n}.o1.w"), c, c, bias=W[f"{n}.o1.b"], residual=h0, row_stats=self.folds(o, m, c, c))
        self.arena.free(o)
        self.arena.free(h0)
        # cross-attention (K/V of the prompt embeddings are per-call constants)
        q2 = self.ln_linear(h1, m, c, n, "ln2", "q2", c)
        self._free_rs(h1)
        kp = kv.data_ptr()
        o = self.attention(q2, c, kp, kp + 2 * c, 2 * c, batch, a.heads, hw, n_ctx, n_ctx)
        self.arena.free(q2)
        h2 = self.linear(o, self._Wl(f"{n}.o2.w"), c, c, bias=W[f"{n}.o2.b"], residual=h1, row_stats=self.folds(o, m, c, 8 * c, True))
        self.arena.free(o)
        self.arena.free(h1)
        # GEGLU feed-forward
        rows_per = m
        if self._ff_chunk_bytes and m * 4 * c * 2 > 2 * self._ff_chunk_bytes:
            # ff.net.0 -> ff.net.2 in row chunks whose 4C-wide intermediate (1.34 GB at the 64x64 level of batch 64) fits the 256 MB
            # Infinity Cache, so the second GEMM reads what the first has just written before it leaves for HBM.  Measured (batch 64,
            # three A/B pairs): 160 MB chunks +0.3...+0.6 %, 224-320 MB neutral, 96 MB -1.9 %, 48 MB -3.8 % (smaller grids, more launches);
            # batch 16 / 32: +0.2 / +0.4 %
            rows_per = max(1024, self._ff_chunk_bytes // (4 * c * 2) // 1024 * 1024)
        if rows_per >= m:
            gg = self.ln_linear(h2, m, c, n, "ln3", "ff1", 8 * c, bias=W[f"{n}.ff1.b"], geglu=True)
            self._free_rs(h2)
            h3 = self.linear(gg, W[f"{n}.ff2.w"], c, 4 * c, bias=W[f"{n}.ff2.b"], residual=h2)
            self.arena.free(gg)
        else:
            h3 = self.arena.alloc((m, c), self.tdt)
            rs = getattr(h2, "_rs", None)
            for r0 in range(0, m, rows_per):
                r1 = min(m, r0 + rows_per)
                hc = h2[r0:r1]
                if rs is not None:
                    hc._rs = (rs[0][r0 * rs[1] * 2:r1 * rs[1] * 2], rs[1])
                gg = self.ln_linear(hc, r1 - r0, c, n, "ln3", "ff1", 8 * c, bias=W[f"{n}.ff1.b"], geglu=True)
                self.linear(gg, W[f"{n}.ff2.w"], c, 4 * c, bias=W[f"{n}.ff2.b"], residual=hc, out=h3[r0:r1])
                self.arena.free(gg)
            self._free_rs(h2)
        self.arena.free(h2)
        # out_stats: the next consumer is a non-concatenated GroupNorm (a resnet's norm1, conv_norm_out)
        out = self.gemm([(h3, c, 1, h, w_, 0)], W[f"{n}.proj_out.w"], c, batch, h, w_, bias=W[f"{n}.proj_out.b"], residual=x,
                        gn_stats=G if out_stats else 0)
        self.arena.free(h3)
        return out

    def cross_kv(self, ctx: torch.Tensor, batch: int, n_ctx: int) -> Dict[str, torch.Tensor]:
        """ctx: [batch*n_ctx, cross_dim] operand dtype -> per attention module [batch, n_ctx, 2C]
        (to_k | to_v of attn2; constant over the sampling loop)."""
        out = {}
        cd = self.ucfg.cross_attention_dim
        for a in self._attn_specs:
            t = torch.empty((batch * n_ctx, 2 * a.channels), dtype=self.tdt, device=self.device)
            out[a.name] = self.linear(ctx, self._Wl(f"{a.name}.kv2.w"), 2 * a.channels, cd, out=t)
        return out

    def time_tables(self, timesteps: torch.Tensor) -> torch.Tensor:
        """timesteps fp32 [n] on device -> time_emb_proj(silu(time_embedding(sinusoid(t)))) for all 22
        resnets, concatenated: [n, sum(Cout)] fp32."""
        n = timesteps.numel()
        cfg, W = self.ucfg, self.w
        te = cfg.time_embed_dim
        s = torch.empty((n, cfg.time_proj_dim), dtype=torch.float32, device=self.device)
        a = torch.empty((n, te), dtype=torch.float32, device=self.device)
        b = torch.empty((n, te), dtype=torch.float32, device=self.device)
        out = torch.empty((n, self.tproj_total), dtype=torch.float32, device=self.device)
        st = _stream()
        L.check(self.lib.idb_timestep_sinusoid(timesteps.data_ptr(), s.data_ptr(), n, cfg.time_proj_dim, st), "sinusoid")
        L.check(self.lib.idb_linear_f32(s.data_ptr(), W["te.linear_1.w"].data_ptr(), W["te.linear_1.b"].data_ptr(),
                                        a.data_ptr(), n, te, cfg.time_proj_dim, 0, st), "linear_f32")
        L.check(self.lib.idb_linear_f32(a.data_ptr(), W["te.linear_2.w"].data_ptr(), W["te.linear_2.b"].data_ptr(),
                                        b.data_ptr(), n, te, te, 1, st), "linear_f32")
        L.check(self.lib.idb_linear_f32(b.data_ptr(), W["tproj.w"].data_ptr(), W["tproj.b"].data_ptr(), out.data_ptr(),
                                        n, self.tproj_total, te, 1, st), "linear_f32")
        self._last_temb = b
        return out

    def unet_nhwc(self, lat: torch.Tensor, rep: int, sbias, kv: Dict[str, torch.Tensor], n_ctx: int,
                  eps_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """lat: fp32 NCHW [B,4,h,w] (each sample is fed `rep` times: CFG) -> eps fp32 [rep*B*h*w, 4]."""
        g, cfg, W = self.ugraph, self.ucfg, self.w
        b0, cin, h, w_ = lat.shape
        B = b0 * rep
        self._rep = rep
        if self.groups > 1 and b0 % self.groups:
            raise ValueError(f"set_lora_groups({self.groups}) needs a batch that is a multiple of the group count, got {b0}")
        eps_n = cfg.norm_eps
        launches0 = self.lib.idb_launch_count()
        c0 = cfg.block_out_channels[0]
        x = self.arena.alloc((B * h * w_, c0), self.tdt)
        L.check(self.lib.idb_conv_in(lat.data_ptr(), W["conv_in.w"].data_ptr(), W["conv_in.b"].data_ptr(), x.data_ptr(),
                                     b0, rep, cin, h, w_, c0, 1.0, None, None, self.dt, _stream()), "idb_conv_in")
        skips: List[Tuple[torch.Tensor, int]] = [(x, c0)]
        self._pinned.add(x.data_ptr())
        self._tap("conv_in", x)
        ch = c0
        for blk in g.down:
            for j, r in enumerate(blk["resnets"]):
                # the next consumer of y is a non-concatenated GroupNorm (Transformer2DModel.norm or the next resnet's norm1)
                # unless the block's downsample conv comes first
                last = j == len(blk["resnets"]) - 1
                y = self._resnet(r.name, x, r.cin, None, 0, r.cout, B, h, w_, sbias, eps_n,
                                 out_stats=bool(blk["attns"]) or not (last and blk["down"]))
                self._free(x)
                x, ch = self._tap(r.name, y), r.cout
                if blk["attns"]:
                    y = self._transformer(blk["attns"][j], x, B, h, w_, kv[blk["attns"][j].name], n_ctx, out_stats=not (last and blk["down"]))
                    self._free(x)
                    x = self._tap(blk["attns"][j].name, y)
                skips.append((x, ch))
                self._pinned.add(x.data_ptr())
            if blk["down"]:
                y = self.gemm([(x, ch, 9, h, w_, 0)], W[blk["down"] + ".w"], ch, B, h // 2, w_ // 2,
                              bias=W[blk["down"] + ".b"], stride=2, gn_stats=cfg.norm_num_groups)   # next: a resnet's norm1
                h, w_ = h // 2, w_ // 2
                x = self._tap(blk["down"], y)
                skips.append((x, ch))
                self._pinned.add(x.data_ptr())
        m = g.mid
        y = self._resnet(m["resnets"][0].name, x, ch, None, 0, ch, B, h, w_, sbias, eps_n, out_stats=True)
        x = self._tap(m["resnets"][0].name, y)          # previous x is the last skip: stays pinned
        y = self._transformer(m["attn"], x, B, h, w_, kv[m["attn"].name], n_ctx, out_stats=True)
        self._free(x)
        x = self._tap(m["attn"].name, y)
        y = self._resnet(m["resnets"][1].name, x, ch, None, 0, ch, B, h, w_, sbias, eps_n)
        self._free(x)
        x = self._tap(m["resnets"][1].name, y)
        for bi, blk in enumerate(g.up):
            for j, r in enumerate(blk["resnets"]):
                sk, sc = skips.pop()
                assert sc == r.skip_channels and ch + sc == r.cin
                y = self._resnet(r.name, x, ch, sk, sc, r.cout, B, h, w_, sbias, eps_n, out_stats=bool(blk["attns"]))
                self._pinned.discard(sk.data_ptr())
                self.arena.free(sk)
                self._free(x)
                x, ch = self._tap(r.name, y), r.cout
                if blk["attns"]:
                    final = bi == len(g.up) - 1 and j == len(blk["resnets"]) - 1 and not blk["up"]     # next: conv_norm_out
                    y = self._transformer(blk["attns"][j], x, B, h, w_, kv[blk["attns"][j].name], n_ctx, out_stats=final)
                    self._free(x)
                    x = self._tap(blk["attns"][j].name, y)
            if blk["up"]:
                y = self.gemm([(x, ch, 9, h, w_, 1)], W[blk["up"] + ".w"], ch, B, 2 * h, 2 * w_, bias=W[blk["up"] + ".b"])
                self._free(x)
                x = self._tap(blk["up"], y)
                h, w_ = 2 * h, 2 * w_
        assert not skips
        n = self.groupnorm(x, ch, None, 0, B, h * w_, W["conv_norm_out.g"], W["conv_norm_out.b"], eps_n, True)
        self._free(x)
        eps = self.gemm([(n, ch, 9, h, w_, 0)], W["conv_out.w"], cfg.out_channels, B, h, w_, bias=W["conv_out.b"],
                        out_f32=True, out=eps_out)
        self.arena.free(n)
        self.last_forward_launches = int(self.lib.idb_launch_count() - launches0)     # kernels of ONE (CFG) UNet forward
        return eps

    def unet_forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        """API form (train_ID-Booth.py:1040-1046): sample [B,4,h,w], timestep scalar or [B],
        encoder_hidden_states [B,L,cross_dim]; returns eps NCHW fp32."""
        self.arena.reset()
        self._pinned.clear()
        B, _, h, w_ = sample.shape
        lat = sample.to(device=self.device, dtype=torch.float32).contiguous()
        ts = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
        ts = (ts.expand(B) if ts.numel() == 1 else ts).contiguous().to(self.device)
        tp = self.time_tables(ts)
        ctx = self.cast(encoder_hidden_states.to(self.device).float().reshape(-1, encoder_hidden_states.shape[-1]))
        n_ctx = encoder_hidden_states.shape[1]
        self._rep = 1                                  # API form: LoRA groups (if any) partition the GIVEN batch contiguously
        kv = self.cross_kv(ctx, B, n_ctx)
        eps = self.unet_nhwc(lat, 1, (tp, 0, self.tproj_total), kv, n_ctx)
        out = torch.empty((B, self.ucfg.out_channels, h, w_), dtype=torch.float32, device=self.device)
        L.check(self.lib.idb_f32_nhwc_to_nchw(eps.data_ptr(), out.data_ptr(), B, h * w_, self.ucfg.out_channels, _stream()),
                "idb_f32_nhwc_to_nchw")
        self.arena.free(eps)
        return out

    def cast(self, x_f32: torch.Tensor) -> torch.Tensor:
        x_f32 = x_f32.contiguous()
        out = torch.empty(x_f32.shape, dtype=self.tdt, device=self.device)
        L.check(self.lib.idb_cast_f32(x_f32.data_ptr(), out.data_ptr(), x_f32.numel(), self.dt, _stream()), "idb_cast_f32")
        return out

    # ------------------------------------------------------------------------------------
    # sampling loop (StableDiffusionPipeline.__call__ steps 3-5)
    # ------------------------------------------------------------------------------------
    def _sample_body(self, ctx_f32, noise, coefs, tp, lat, eps, steps, rep, n_ctx, vpred, trace=None, hist=None) -> None:
        """One whole sampling loop on the current stream; every buffer is passed in, so the same
        code runs eagerly or under HIP-graph capture."""
        B = noise.shape[1]
        h, w_ = lat.shape[2], lat.shape[3]
        self.arena.reset()
        self._pinned.clear()
        self._rep = rep
        lat.copy_(noise[0])                                  # init_noise_sigma = 1
        if hist is not None:
            hist.zero_()                                     # multistep solver: x0 history (coefficient 0 on the first step)
        ctx = self.arena.alloc(tuple(ctx_f32.shape), self.tdt)
        L.check(self.lib.idb_cast_f32(ctx_f32.data_ptr(), ctx.data_ptr(), ctx_f32.numel(), self.dt, _stream()), "idb_cast_f32")
        kv = {}
        cd = self.ucfg.cross_attention_dim
        for a in self._attn_specs:
            kv[a.name] = self.linear(ctx, self._Wl(f"{a.name}.kv2.w"), 2 * a.channels, cd)
            self._pinned.add(kv[a.name].data_ptr())
        self.arena.free(ctx)
        for i in range(steps):
            self.unet_nhwc(lat, rep, (tp, i * self.tproj_total, 0), kv, n_ctx, eps_out=eps)
            if hist is None:                                 # DDPM: third operand = this step's variance noise
                third, x0_out = noise[i + 1].data_ptr(), None
            else:                                            # DPM-Solver++ 2M: third operand = previous x0 prediction; keep this one
                third, x0_out = hist[(i + 1) & 1].data_ptr(), hist[i & 1].data_ptr()
            L.check(self.lib.idb_cfg_ddpm_step(eps.data_ptr(), lat.data_ptr(), third, coefs[i].data_ptr(),
                                               x0_out, B, lat.shape[1], h * w_, int(rep == 2), int(vpred), _stream()),
                    "idb_cfg_ddpm_step")
            if trace is not None:
                trace.append((eps.clone(), lat.clone()))

    def sample(self, prompt_embeds: torch.Tensor, negative_prompt_embeds: Optional[torch.Tensor], noise: torch.Tensor,
               timesteps: Sequence[int], coefs: torch.Tensor, vpred: bool = False, use_graph: bool = False,
               trace: Optional[list] = None, multistep: bool = False) -> torch.Tensor:
        """noise: [steps+1, B, 4, h, w] fp32 on device (noise[0] = initial latents); coefs: [steps, 6] fp32 on
        device (sqrt_a, sqrt_b, c_x0, c_x, sigma, guidance_scale).  CFG is on iff negative_prompt_embeds is
        given.  Returns final latents [B,4,h,w] fp32.  ``trace`` collects (eps [2B*hw,4], latents) per step.
        multistep (DPM-Solver++ 2M, train_ID-Booth.py:155): noise is [1, B, 4, h, w] (initial latents only) and the fifth
        coefficient multiplies the previous step's x0 prediction instead of fresh noise."""
        steps = len(timesteps)
        B, lc, h, w_ = noise.shape[1:]
        cfg_on = negative_prompt_embeds is not None
        rep = 2 if cfg_on else 1
        n_ctx = prompt_embeds.shape[1]
        ctx_f32 = torch.cat([negative_prompt_embeds, prompt_embeds]) if cfg_on else prompt_embeds   # uncond FIRST
        ctx_f32 = ctx_f32.to(self.device).float().reshape(rep * B * n_ctx, -1).contiguous()
        noise = noise.to(self.device).float().contiguous()
        coefs = coefs.to(self.device).float().contiguous()
        ts = torch.tensor([float(t) for t in timesteps], dtype=torch.float32, device=self.device)
        tp = self.time_tables(ts)
        if not use_graph or trace is not None:
            lat = torch.empty((B, lc, h, w_), dtype=torch.float32, device=self.device)
            eps = torch.empty((rep * B * h * w_, self.ucfg.out_channels), dtype=torch.float32, device=self.device)
            hist = torch.empty((2, B, lc, h, w_), dtype=torch.float32, device=self.device) if multistep else None
            self._sample_body(ctx_f32, noise, coefs, tp, lat, eps, steps, rep, n_ctx, vpred, trace, hist)
            return lat
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        key = ("sample", B, lc, h, w_, steps, n_ctx, cfg_on, vpred, self.dtype_name, multistep, self.groups)
        ent = self._graphs.get(key)
        if ent is None:
            ent = {"ctx": ctx_f32.clone(), "noise": noise.clone(), "coefs": coefs.clone(), "tp": tp.clone(),
                   "lat": torch.empty((B, lc, h, w_), dtype=torch.float32, device=self.device),
                   "eps": torch.empty((rep * B * h * w_, self.ucfg.out_channels), dtype=torch.float32, device=self.device)}
            # every buffer the captured graph references lives in `ent` (the multistep x0 history too: a local here would go
            # back to the caching allocator while replays keep writing through its address)
            ent["hist"] = torch.empty((2, B, lc, h, w_), dtype=torch.float32, device=self.device) if multistep else None
            args = (ent["ctx"], ent["noise"], ent["coefs"], ent["tp"], ent["lat"], ent["eps"], steps, rep, n_ctx, vpred, None,
                    ent["hist"])
            self._sample_body(*args)                     # eager warm-up: allocates every arena block, sets func attributes
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._sample_body(*args)
            ent["graph"] = graph
            self._graphs[key] = ent
        ent["ctx"].copy_(ctx_f32)
        ent["noise"].copy_(noise)
        ent["coefs"].copy_(coefs)
        ent["tp"].copy_(tp)
        ent["graph"].replay()
        return ent["lat"].clone()

    # ------------------------------------------------------------------------------------
    # VAE decoder (AutoencoderKL.decode -> Decoder.forward)
    # ------------------------------------------------------------------------------------
    def _vae_attention(self, x, batch, h, w_, c, a: str = "decoder.mid_block.attentions.0") -> torch.Tensor:
        W = self.w
        hw = h * w_
        G = self.vcfg.norm_num_groups
        xn = self.groupnorm(x, c, None, 0, batch, hw, W[f"{a}.gn.g"], W[f"{a}.gn.b"], self.vcfg.norm_eps, False, G)
        q = self.linear(xn, W[f"{a}.q.w"], c, c, bias=W[f"{a}.q.b"])
        k = self.linear(xn, W[f"{a}.k.w"], c, c, bias=W[f"{a}.k.b"])
        o = self.arena.alloc((batch * hw, c), self.tdt)
        esz = 2
        for b in range(batch):
            xb = xn[b * hw:(b + 1) * hw]
            # V^T[c][token] = Wv x^T (bias folded in after P V: rows of P sum to 1)
            vt = self.gemm([(W[f"{a}.v.w"], c, 1, 1, 1, 0)], xb, hw, c, 1, 1)
            s = self.gemm([(q[b * hw:(b + 1) * hw], c, 1, 1, 1, 0)], k[b * hw:(b + 1) * hw], hw, hw, 1, 1,
                          out_scale=float(c) ** -0.5)
            L.check(self.lib.idb_softmax_rows(s.data_ptr(), hw, hw, self.dt, _stream()), "idb_softmax_rows")
            self.gemm([(s, hw, 1, 1, 1, 0)], vt, c, hw, 1, 1, bias=W[f"{a}.v.b"], out=o[b * hw:(b + 1) * hw])
            self.arena.free(s)
            self.arena.free(vt)
        self.arena.free(q)
        self.arena.free(k)
        self.arena.free(xn)
        out = self.linear(o, W[f"{a}.o.w"], c, c, bias=W[f"{a}.o.b"], residual=x)
        self.arena.free(o)
        return out

    def vae_decode_nhwc(self, z: torch.Tensor, in_scale: float) -> torch.Tensor:
        """z fp32 NCHW [B,4,h,w] (multiplied by in_scale on load) -> decoded image fp32 [B*H*W, 3]."""
        W, g, cfg = self.w, self.vgraph, self.vcfg
        B, lc, h, w_ = z.shape
        G, eps_n = cfg.norm_num_groups, cfg.norm_eps
        cm = g.mid_channels
        x = self.arena.alloc((B * h * w_, cm), self.tdt)
        L.check(self.lib.idb_conv_in(z.data_ptr(), W["v.conv_in.w"].data_ptr(), W["v.conv_in.b"].data_ptr(), x.data_ptr(), B, 1,
                                     lc, h, w_, cm, float(in_scale), W["v.pq.w"].data_ptr(), W["v.pq.b"].data_ptr(), self.dt,
                                     _stream()), "idb_conv_in")
        y = self._resnet("decoder.mid_block.resnets.0", x, cm, None, 0, cm, B, h, w_, None, eps_n, G)
        self.arena.free(x)
        x = self._vae_attention(y, B, h, w_, cm)
        self.arena.free(y)
        y = self._resnet("decoder.mid_block.resnets.1", x, cm, None, 0, cm, B, h, w_, None, eps_n, G)
        self.arena.free(x)
        x, ch = y, cm
        for blk in g.up:
            for name, cin, cout in blk["resnets"]:
                y = self._resnet(name, x, cin, None, 0, cout, B, h, w_, None, eps_n, G)
                self.arena.free(x)
                x, ch = y, cout
            if blk["up"]:
                y = self.gemm([(x, ch, 9, h, w_, 1)], W[blk["up"] + ".w"], ch, B, 2 * h, 2 * w_, bias=W[blk["up"] + ".b"])
                self.arena.free(x)
                x = y
                h, w_ = 2 * h, 2 * w_
        n = self.groupnorm(x, ch, None, 0, B, h * w_, W["v.norm_out.g"], W["v.norm_out.b"], eps_n, True, G)
        self.arena.free(x)
        img = self.gemm([(n, ch, 9, h, w_, 0)], W["v.conv_out.w"], cfg.out_channels, B, h, w_, bias=W["v.conv_out.b"],
                        out_f32=True)
        self.arena.free(n)
        return img

    # ------------------------------------------------------------------------------------
    # VAE encoder (AutoencoderKL.encode -> Encoder.forward + quant_conv; train_ID-Booth.py:1001)
    # ------------------------------------------------------------------------------------
    def vae_encode_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """x fp32 NCHW [B,3,H,W] in [-1,1] -> moments fp32 [B*h*w, 2*latent_channels] (mean channels, then log-variance)."""
        if not getattr(self, "has_vae_encoder", False):
            raise RuntimeError("VAE encoder weights are not loaded (pack_vae_encoder)")
        W, cfg = self.w, self.vcfg
        B, ic, h, w_ = x.shape
        G, eps_n = cfg.norm_num_groups, cfg.norm_eps
        c0 = cfg.block_out_channels[0]
        cur = self.arena.alloc((B * h * w_, c0), self.tdt)
        L.check(self.lib.idb_conv_in(x.data_ptr(), W["ve.conv_in.w"].data_ptr(), W["ve.conv_in.b"].data_ptr(), cur.data_ptr(), B, 1,
                                     ic, h, w_, c0, 1.0, None, None, self.dt, _stream()), "idb_conv_in")
        ch = c0
        for blk in S.vae_encoder_blocks(cfg):
            for name, cin, cout in blk["resnets"]:
                y = self._resnet(name, cur, cin, None, 0, cout, B, h, w_, None, eps_n, G)
                self.arena.free(cur)
                cur, ch = y, cout
            if blk["down"]:
                if h % 2 or w_ % 2:
                    raise ValueError("VAE encoder needs image sides that are multiples of 8")
                y = self.gemm([(cur, ch, 9, h, w_, 0)], W[blk["down"] + ".w"], ch, B, h // 2, w_ // 2, bias=W[blk["down"] + ".b"],
                              stride=2, pad_mode=1)
                self.arena.free(cur)
                cur = y
                h, w_ = h // 2, w_ // 2
        y = self._resnet("encoder.mid_block.resnets.0", cur, ch, None, 0, ch, B, h, w_, None, eps_n, G)
        self.arena.free(cur)
        cur = self._vae_attention(y, B, h, w_, ch, "encoder.mid_block.attentions.0")
        self.arena.free(y)
        y = self._resnet("encoder.mid_block.resnets.1", cur, ch, None, 0, ch, B, h, w_, None, eps_n, G)
        self.arena.free(cur)
        n = self.groupnorm(y, ch, None, 0, B, h * w_, W["ve.norm_out.g"], W["ve.norm_out.b"], eps_n, True, G)
        self.arena.free(y)
        mom = self.gemm([(n, ch, 9, h, w_, 0)], W["ve.conv_out.w"], 2 * cfg.latent_channels, B, h, w_, bias=W["ve.conv_out.b"],
                        out_f32=True)
        self.arena.free(n)
        return mom

    def vae_encode(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None, scale: float = 1.0, chunk: int = 4):
        """``vae.encode(x).latent_dist``: returns (latents, mean, logvar), NCHW fp32.  latents = (mean + std * noise) * scale
        with noise [B,C,h,w] (``.sample()``) or mean * scale when noise is None (``.mode()``)."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, H, Wd = x.shape
        down = 2 ** (len(self.vcfg.block_out_channels) - 1)
        if H % down or Wd % down:
            raise ValueError(f"image sides must be multiples of {down}")
        h, w_, lc = H // down, Wd // down, self.vcfg.latent_channels
        lat = torch.empty((B, lc, h, w_), dtype=torch.float32, device=self.device)
        mean, logvar = torch.empty_like(lat), torch.empty_like(lat)
        nz = None if noise is None else noise.to(device=self.device, dtype=torch.float32).contiguous()
        if nz is not None and tuple(nz.shape) != tuple(lat.shape):
            raise ValueError(f"noise must have shape {tuple(lat.shape)}")
        for b0 in range(0, B, chunk):
            self.arena.reset()
            self._pinned.clear()
            xb = x[b0:b0 + chunk].contiguous()
            mom = self.vae_encode_nhwc(xb)
            L.check(self.lib.idb_vae_sample(mom.data_ptr(), None if nz is None else nz[b0:b0 + chunk].data_ptr(), float(scale),
                                            lat[b0:b0 + chunk].data_ptr(), mean[b0:b0 + chunk].data_ptr(),
                                            logvar[b0:b0 + chunk].data_ptr(), xb.shape[0], lc, h * w_, _stream()), "idb_vae_sample")
            self.arena.free(mom)
        return lat, mean, logvar

    def vae_decode(self, z: torch.Tensor, in_scale: float = 1.0, chunk: int = 4):
        """Returns (raw NCHW fp32 [B,3,H,W])  — the ``vae.decode(z).sample`` API form."""
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, h, w_ = z.shape
        up = 2 ** (len(self.vcfg.block_out_channels) - 1)
        H, Wd = h * up, w_ * up
        oc = self.vcfg.out_channels
        out = torch.empty((B, oc, H, Wd), dtype=torch.float32, device=self.device)
        for b0 in range(0, B, chunk):
            self.arena.reset()
            self._pinned.clear()
            zb = z[b0:b0 + chunk].contiguous()
            img = self.vae_decode_nhwc(zb, in_scale)
            L.check(self.lib.idb_f32_nhwc_to_nchw(img.data_ptr(), out[b0:b0 + chunk].data_ptr(), zb.shape[0], H * Wd, oc,
                                                  _stream()), "idb_f32_nhwc_to_nchw")
            self.arena.free(img)
        return out

    def decode_images(self, latents: torch.Tensor, chunk: int = 4, want_u8: bool = True):
        """latents (scaled) -> (img01 fp32 NHWC [B,H,W,3], uint8 NHWC) — pipeline steps 6 and 8 plus the
        save_image quantisation (inference_ID-Booth.py:139-144)."""
        latents = latents.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, h, w_ = latents.shape
        up = 2 ** (len(self.vcfg.block_out_channels) - 1)
        H, Wd = h * up, w_ * up
        oc = self.vcfg.out_channels
        img01 = torch.empty((B, H, Wd, oc), dtype=torch.float32, device=self.device)
        u8 = torch.empty((B, H, Wd, oc), dtype=torch.uint8, device=self.device) if want_u8 else None
        for b0 in range(0, B, chunk):
            self.arena.reset()
            self._pinned.clear()
            zb = latents[b0:b0 + chunk].contiguous()
            img = self.vae_decode_nhwc(zb, 1.0 / self.vcfg.scaling_factor)
            L.check(self.lib.idb_postprocess(img.data_ptr(), img01[b0:b0 + chunk].data_ptr(),
                                             _ptr(u8[b0:b0 + chunk]) if want_u8 else None, img.numel(), _stream()),
                    "idb_postprocess")
            self.arena.free(img)
        return img01, u8


def ddpm_step_device(model_output: torch.Tensor, sample: torch.Tensor, noise: Optional[torch.Tensor], coef5,
                     vpred: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """DDPMScheduler.step on GPU tensors (NCHW fp32) through idb_cfg_ddpm_step (no CFG)."""
    lib = L.load()
    dev = sample.device
    B, Cc, h, w_ = sample.shape
    mo = model_output.to(dtype=torch.float32).permute(0, 2, 3, 1).contiguous()       # kernel reads eps as [B][HW][C]
    lat = sample.to(dtype=torch.float32).contiguous().clone()
    x0 = torch.empty_like(lat)
    coef = torch.tensor(list(coef5) + [0.0], dtype=torch.float32, device=dev)
    nz = None if noise is None else noise.to(device=dev, dtype=torch.float32).contiguous()
    L.check(lib.idb_cfg_ddpm_step(mo.data_ptr(), lat.data_ptr(), _ptr(nz), coef.data_ptr(), x0.data_ptr(), B, Cc, h * w_, 0,
                                  int(vpred), _stream()), "idb_cfg_ddpm_step")
    return lat, x0

/*
 * idb_kernels.h — C ABI of libidb_kernels.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * ID-Booth sampling path (SD-2.1 UNet forward, CFG + DDPM step, VAE decode).
 *
 * The reference has no FFI for this path: it is a Python object protocol
 * (diffusers.StableDiffusionPipeline, /root/reference/inference_ID-Booth.py:103-108,138), and every
 * arithmetic step is dispatched by torch/cuDNN/cuBLAS inside un-vendored diffusers.  Each entry point
 * below therefore cites the upstream op it replaces and the reference call
 * site that reaches it.  INTEGRATION.md shows the ctypes binding a maintainer of the reference adds.
 *
 * Conventions:
 *   - plain C types only; device pointers are void*; `stream` is a hipStream_t passed as void*;
 *   - every function returns 0 (IDB_OK) or a negative idb_status; idb_last_error() gives the text
 *     (thread-local);
 *   - no ownership transfer: the caller allocates inputs, outputs and workspaces;
 *   - kernels are asynchronous on `stream` and never synchronise (graph-capturable);
 *   - activations are NHWC ("channels-last") in the operand dtype (bf16 or f16), so a [B,H,W,C]
 *     feature map and a [B, H*W, C] token matrix are the same bytes;
 *   - weights are [N][K] row-major in the operand dtype (torch Linear convention); conv weights are
 *     repacked to [Cout][tap][Cin] by idb_pack_conv_weight.
 */
#ifndef IDB_KERNELS_H
#define IDB_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { IDB_OK = 0, IDB_EINVAL = -1, IDB_EUNSUPPORTED = -2, IDB_EHIP = -3 } idb_status;
typedef enum { IDB_BF16 = 0, IDB_F16 = 1, IDB_F32 = 2 } idb_dtype;

#define IDB_MAX_SRC 4

int idb_version(void);
/* Number of kernels this library has launched in this process so far (every entry point counts each of its launches, e.g. a
 * split-K idb_gemm counts the GEMM and its reduce launch).  For launch-count accounting (bench.py, tests); monotonic. */
uint64_t idb_launch_count(void);
const char* idb_last_error(void);
/* 0 iff `device` is a gfx950 part (the only target this library is built for). */
int idb_device_check(int device);

/* ------------------------------------------------------------------------------------------
 * K1/K2/K3 — implicit GEMM:  out[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue)
 *
 * Replaces: nn.Conv2d 3x3 (stride 1/2, pad 1), nearest-2x upsample + conv, 1x1 conv_shortcut,
 * nn.Linear, GEGLU — i.e. diffusers ResnetBlock2D / Downsample2D / Upsample2D / Attention.to_* /
 * FeedForward as dispatched from UNet2DConditionModel.forward (inference_ID-Booth.py:138,
 * train_ID-Booth.py:1040-1046) and Decoder.forward (train_ID-Booth.py:410-412).
 *
 * A is never materialised: row m = output pixel (b, oy, ox); the K axis is the concatenation of up
 * to IDB_MAX_SRC sources, each contributing taps*channels columns ordered [tap][channel]:
 *   taps = 9 : 3x3 window with zero padding 1 read from an NHWC tensor [batch][in_h][in_w][channels]
 *              (upsample = 1: the tensor is logically nearest-2x upsampled first);
 *   taps = 1 : the pixel itself (1x1 conv, or a plain [M][K] matrix with in_h = in_w = 1).
 * Several sources give skip-concatenation and the fused 1x1 shortcut without a concat pass.
 * channels must be a multiple of 64 for every source.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* ptr;
    int32_t channels;
    int32_t taps;      /* 9 or 1 */
    int32_t in_h, in_w;
    int32_t upsample;  /* 0 or 1 */
} idb_gemm_src;

typedef struct {
    int32_t dtype;            /* IDB_BF16 or IDB_F16: operand dtype of A, W, residual */
    int32_t batch, out_h, out_w;   /* M = batch*out_h*out_w */
    int32_t stride;           /* 1 or 2, applies to taps=9 sources */
    int32_t n;                /* rows of W (before GEGLU halving) */
    int32_t nsrc;
    idb_gemm_src src[IDB_MAX_SRC];
    const void* w;            /* [n][K] */
    const float* bias;        /* [n] or NULL */
    const float* sample_bias; /* [batch][sample_bias_ld] added per sample (temb projection) or NULL */
    int32_t sample_bias_ld;   /* 0 = the same row for every sample */
    const void* residual;     /* [M][out_ld] operand dtype, or NULL */
    int32_t geglu;            /* 1: W rows interleaved by idb_pack_matrix(geglu=1); out has n/2 cols */
    void* out;
    int32_t out_dtype;        /* IDB_BF16/IDB_F16 (== dtype) or IDB_F32 */
    int32_t out_ld;           /* elements between output rows */
    int32_t split_k;          /* 0 = library heuristic, >=1 explicit */
    int32_t tile;             /* 0 = heuristic; else a tile config id as idb_gemm_plan reports it (tests and measurements) */
    float out_scale;          /* multiplies the accumulator before bias (0 => 1.0) */
    int32_t flags;            /* profiling/testing only — bit 0: skip the split-K reduce launch (`out` not written); bit 1: skip the epilogue stores; bit 2: force the direct (non-LDS-staged) epilogue; bit 3: force the two-launch split-K reduce; bit 4: in-kernel split-K reduce (default: separate reduce launch, which measured faster); bit 8: let the persistent variant fold a LayerNorm (ln_stats; default: IDB_EUNSUPPORTED there, the separate idb_layernorm measured no slower) */
    int32_t act;              /* 0 none, 1 exact GELU applied to (acc*scale + bias) (CLIP MLP fc1); not with residual/GEGLU */
    uint32_t* counters;       /* optional: >= counters_len zeroed uint32 on the device, private to the stream; used only with
                                 flags bit 4: a split-K launch then reduces inside the GEMM (the last-arriving workgroup of a
                                 tile sums the slabs in fixed order and runs the epilogue; counters are left zero) */
    int32_t counters_len;
    float* gn_partials;       /* optional out: per-(sample, 64-row block, group) {sum, sum of squares} of the ROUNDED output, fp32
                                 [batch][out_h*out_w/64][gn_groups][2] — the first pass of the nn.GroupNorm that consumes `out`
                                 (idb_groupnorm's partials_in), produced by the split-K reduce launch when there is one, by the GEMM's own
                                 LDS-staged epilogue when its column tiles hold whole groups (idb_gemm_emits_gn_partials), by an extra
                                 statistics launch otherwise.  Needs out_h*out_w % 64 == 0, out_h*out_w <= 4096, n % gn_groups == 0,
                                 operand-dtype output, no GEGLU */
    int32_t gn_groups;
    /* LayerNorm folded into the GEMM (BasicTransformerBlock norm1/2/3 -> to_q/k/v, attn2.to_q, ff.net.0): `src` holds the RAW rows
     * x [M][K = ln_c]; `w` = W scaled by the LayerNorm gamma along K; out = rstd_m (x W'^T - mean_m u) + v, with
     * u[n] = sum_k W'[n][k] (of the ROUNDED operand values) and v[n] = sum_k beta[k] W[n][k] + the layer's bias[n] (fp32 [n],
     * 16-byte aligned; `bias` and `sample_bias` must be NULL: the epilogue keeps two column vectors per fragment in flight), and the
     * per-row statistics taken from ln_stats = the row_stats_out of the GEMM that produced x: fp32 [M][ln_tiles][2] partial
     * {sum, sum of squares} per column tile.  Exact in fp32 (x W'^T - mean u = (x - mean) W'^T term by term).  row_stats_out needs
     * a plan that runs the LDS-staged epilogue (no split-K, no persistent variant, operand-dtype output), ln_stats that or the
     * persistent variant: idb_gemm returns IDB_EUNSUPPORTED otherwise and the caller keeps idb_layernorm; idb_gemm_row_stats_tiles
     * (producer side, 0 = unsupported) and idb_gemm_folds_layernorm (consumer side) tell beforehand. */
    float* row_stats_out;     /* optional out: [M][idb_gemm_row_stats_tiles(d)][2] */
    const float* ln_stats;    /* optional in */
    int32_t ln_tiles;
    const float* ln_u;
    const float* ln_v;
    float ln_eps;
    int32_t pad_mode;         /* taps=9 sources — 0: zero padding 1 on every side (nn.Conv2d padding=1); 1: padding on the
                                 bottom/right only, i.e. F.pad(x, (0,1,0,1)) + padding=0, the stride-2 Downsample2D of the VAE
                                 encoder (diffusers downsampling.py; AutoencoderKL.encode at train_ID-Booth.py:1001) */
    int32_t w_layout;         /* 0: `w` is [n][K] rows (K contiguous); 1: K-tiled 16-row blocks written by idb_tile_weight,
                                 [ceil(n/16)][K/64][16 rows][64 k]: the rows of one K-step of a column tile are whole 2 KiB runs and
                                 a workgroup's K loop reads each of its row blocks as ONE contiguous stream (HBM pages, TLB reach)
                                 instead of 128-byte pieces at a K*2-byte stride */
    /* Grouped weights — a mixed-identity batch in ONE launch (BASELINE configs[2] = 8 identities x 8 prompts; per-row LoRA adapters of
     * peft's lora.Linear, train_ID-Booth.py:672-678, in their MERGED form): `w` holds w_groups weight matrices of identical shape,
     * w_group_stride bytes apart (each [n][K] rows or K-tiled as w_layout says); output row m uses matrix (m / w_group_rows) % w_groups.
     * w_group_rows must be a multiple of the plan's tile height (idb_gemm returns IDB_EUNSUPPORTED otherwise: the caller then runs one
     * launch per group).  With a folded LayerNorm, ln_u / ln_v hold w_groups vectors of n floats each.  0 or 1: one matrix. */
    int32_t w_groups;
    int32_t w_group_rows;
    int64_t w_group_stride;
    /* GroupNorm(+SiLU) fused into the GEMM (north_star "conv3x3 + GroupNorm+SiLU fused"; diffusers ResnetBlock2D norm1+conv1,
     * norm2+conv2(+conv_shortcut), Transformer2DModel norm+proj_in): the first gn_in_nsrc sources hold the RAW tensors; their channel
     * concatenation is normalised with nn.GroupNorm(gn_in_groups, eps) from gn_in_partials (the statistics idb_groupnorm takes as
     * partials_in: [batch][gn_in_chunks][gn_in_groups][2], gn_in_chunks <= 64), gamma / beta over those channels, then SiLU if
     * gn_in_silu — inside the kernel, with idb_groupnorm's arithmetic (the MFMA waves read the operand bits idb_groupnorm would have
     * written); zero padding stays zero.  Two forms, both for plans that run one workgroup per CU (small grids: the batch-1 UNet):
     * 3x3 sources (every source normalised, 64- / 128-row tiles inside one sample) go through the patch-resident conv, whose patch
     * loaders normalise each halo patch once per 64-channel chunk (tile ids 10x); 1x1 sources (proj_in) through dedicated normalizer
     * waves on the landed LDS stage (tile ids 5x-7x).  idb_gemm_fuses_groupnorm tells beforehand, idb_gemm returns IDB_EUNSUPPORTED
     * otherwise and the caller runs idb_groupnorm + idb_gemm.  Needs stride 1 and sources on the output's spatial grid.  NULL: off. */
    const float* gn_in_partials;
    int32_t gn_in_chunks, gn_in_groups, gn_in_nsrc, gn_in_silu;
    float gn_in_eps;
    const float* gn_in_gamma;
    const float* gn_in_beta;
} idb_gemm_desc;

size_t idb_gemm_workspace_bytes(const idb_gemm_desc* d);
/* What idb_gemm would launch for `d`: tile config id (shape + 10 * variant; shapes 1: 128x160, 2: 128x128, 3: 64x160,
 * 4: 64x64 (8 waves), 5: 128x32, 6-9: 64x160 / 64x128 / 128x160 / 128x128 with 8 waves; variant 0/1/2: 2-/3-/4-stage LDS ring,
 * 3: register-staged, 4: persistent, 5-7: loader waves, 8: loader waves with 256-row tiles (shapes 8/9), 9: patch-resident 3x3 conv
 * on 256-row tiles, 10: patch-resident 3x3 conv on the shape's own 64-/128-row tile), split-K factor and workgroup count.
 * Host-only, no GPU call. */
int idb_gemm_plan(const idb_gemm_desc* d, int32_t* tile, int32_t* split_k, int32_t* blocks);
/* Column tiles of the plan idb_gemm would run for `d` if that plan can emit row statistics (LDS-staged epilogue), else 0. */
int32_t idb_gemm_row_stats_tiles(const idb_gemm_desc* d);
/* Would idb_gemm produce gn_partials for `groups` groups WITHOUT an extra statistics launch?  0: no (it would launch one — the caller's
 * two-pass idb_groupnorm is as good); 1: from its split-K reduce launch; 2: from its own LDS-staged epilogue (no split-K, 160-wide
 * tiles holding whole groups). */
int32_t idb_gemm_emits_gn_partials(const idb_gemm_desc* d, int32_t groups);
/* 1 if the plan idb_gemm would run for `d` (bias / sample_bias NULL, one 1x1 source) can apply a folded LayerNorm (ln_*). */
int32_t idb_gemm_folds_layernorm(const idb_gemm_desc* d);
/* 1 if idb_gemm would run `d` with its fused GroupNorm (gn_in_* set) in ONE launch. */
int32_t idb_gemm_fuses_groupnorm(const idb_gemm_desc* d);
int idb_gemm(const idb_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Weight packing (run once at load; SURVEY.md §8b "idb_pack_*"). src is fp32 in torch layout. */
/* [Cout][Cin][kh][kw] fp32 -> [Cout][kh*kw][Cin] operand dtype. */
int idb_pack_conv_weight(const float* src, void* dst, int32_t cout, int32_t cin, int32_t ktaps,
                         int32_t dtype, void* stream);
/* [rows][cols] fp32 -> operand dtype, optional GEGLU row interleave (value/gate in 16-row groups). */
int idb_pack_matrix(const float* src, void* dst, int64_t rows, int64_t cols, int32_t geglu,
                    int32_t dtype, void* stream);
/* [n][k] operand dtype (k % 64 == 0) -> the K-tiled layout idb_gemm_desc.w_layout = 1 reads: [ceil(n/16)][k/64][16][64], rows
 * past n zero-filled; dst holds idb_tiled_weight_bytes(n, k) bytes and must not overlap src. */
size_t idb_tiled_weight_bytes(int64_t n, int64_t k);
int idb_tile_weight(const void* src, void* dst, int64_t n, int64_t k, int32_t dtype, void* stream);
/* dst[rows][cols] = W + scale * B[rows][r] * A[r][cols], fp32 in, operand dtype out: merged LoRA
 * (peft lora.Linear with merged weights; inference_ID-Booth.py:107). */
int idb_lora_merge(const float* w, const float* lora_a, const float* lora_b, void* dst, int64_t rows,
                   int64_t cols, int32_t rank, float scale, int32_t dtype, void* stream);
/* The same with every column k multiplied by col_scale[k] before the one rounding (the gamma of a folded LayerNorm);
 * lora_a == NULL (rank 0): dst = round(W * col_scale). */
int idb_lora_merge_scaled(const float* w, const float* lora_a, const float* lora_b, void* dst, int64_t rows, int64_t cols,
                          int32_t rank, float scale, const float* col_scale, int32_t dtype, void* stream);
/* idb_pack_matrix with every column k multiplied by col_scale[k] before the one rounding (GEGLU projection behind a folded LayerNorm). */
int idb_pack_matrix_scaled(const float* src, void* dst, int64_t rows, int64_t cols, int32_t geglu, const float* col_scale,
                           int32_t dtype, void* stream);
/* The two column vectors of a LayerNorm folded into a projection (idb_gemm_desc.ln_u / ln_v), for output rows r = 0..rows-1
 * (r -> source row sr through the GEGLU interleave when geglu = 1):
 *   u[r] = sum_k float(w_folded[r][k])              w_folded: the ROUNDED gamma-scaled operand-dtype matrix the GEMM multiplies
 *   v[r] = sum_k (W[sr][k] + scale * (B A)[sr][k]) * beta[k] + (bias ? bias[r] : 0)      fp32, W / A / B as idb_lora_merge
 * lora_a == NULL or rank == 0: no adapter.  rank <= 16. */
int idb_ln_fold_vectors(const float* w, const float* lora_a, const float* lora_b, int32_t rank, float scale, const void* w_folded,
                        const float* beta, const float* bias, float* u, float* v, int64_t rows, int64_t cols, int32_t geglu,
                        int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1 (norm part) / K6 — GroupNorm and LayerNorm.
 * Replaces nn.GroupNorm(32, C) (+ SiLU) of ResnetBlock2D / Transformer2DModel.norm / conv_norm_out
 * and nn.LayerNorm(C) x3 of BasicTransformerBlock.
 * GroupNorm input may be the channel-concatenation of two NHWC tensors (skip connections); the
 * output is one dense [B][HW][C0+C1] tensor.  Statistics are fp32, deterministic (fixed summation
 * order; no float atomics).
 * sync: optional array of sync_len int32 counters that is ZERO on first use (the kernel leaves it
 * zero): when given, and the whole grid is resident on the chip at once (small batches), the two
 * passes run as ONE launch whose workgroups hand their partial sums over through these counters;
 * NULL selects the two-launch form.  One array may serve every call on the same stream.
 * partials_in: optional statistics already produced by idb_gemm (idb_gemm_desc.gn_partials) for x0 (then x1 must be NULL);
 * partials_chunks = hw / 64.  Only the normalise pass is launched.
 * ------------------------------------------------------------------------------------------ */
size_t idb_groupnorm_workspace_bytes(int32_t batch, int32_t hw, int32_t groups);
int idb_groupnorm(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw,
                  int32_t groups, float eps, const float* gamma, const float* beta, int32_t silu,
                  void* out, int32_t dtype, void* workspace, size_t workspace_bytes, int32_t* sync,
                  int32_t sync_len, const float* partials_in, int32_t partials_chunks, void* stream);
int idb_layernorm(const void* x, void* out, int64_t rows, int32_t c, float eps, const float* gamma,
                  const float* beta, int32_t dtype, void* stream);
/* First GroupNorm pass only: partial {sum, sum of squares} per (sample, pixel chunk, group) of the channel concatenation x0 | x1,
 * fp32 [batch][*chunks][groups][2] in `partials` (capacity: idb_groupnorm_workspace_bytes) — for callers that
 * want the statistics of a tensor whose producer did not emit them (skip concatenations, conv_in).  *chunks receives the pixel-chunk count (<= 64). */
int idb_groupnorm_stats(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw, int32_t groups,
                        float* partials, size_t partials_bytes, int32_t* chunks, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4/K5 — fused attention, head_dim 64: out = softmax(Q K^T * scale) V, online softmax in fp32.
 * Replaces F.scaled_dot_product_attention in AttnProcessor2_0 (self- and cross-attention).
 * q: [batch][n_q][heads*64] with row stride q_ld; k, v: [batch][n_kv_alloc][...] with row stride
 * kv_ld (n_kv_alloc rows allocated per batch entry, only the first n_kv are attended).
 * ------------------------------------------------------------------------------------------ */
int idb_attention(const void* q, int32_t q_ld, const void* k, const void* v, int32_t kv_ld,
                  void* out, int32_t out_ld, int32_t batch, int32_t heads, int32_t n_q, int32_t n_kv,
                  int32_t n_kv_alloc, float scale, int32_t causal, int32_t dtype, void* stream);
/* causal != 0: query i attends keys 0..i only (CLIPTextModel's causal mask, n_q == n_kv). */

/* Token + position embedding gather of CLIPTextModel: out[b][t][:] = tok[ids[b][t]][:] + pos[t][:]
 * (fp32 tables, operand-dtype output).  ids are int64 on the device; out-of-range ids are an error the
 * caller must exclude (checked on the host by the engine). */
int idb_embed_tokens(const int64_t* ids, const float* tok, const float* pos, void* out, int32_t batch,
                     int32_t n_tokens, int32_t dim, int32_t dtype, void* stream);
/* In-place row softmax over [rows][cols] (VAE mid-block single-head attention, d = 512). */
int idb_softmax_rows(void* x, int64_t rows, int32_t cols, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7 — time embedding path in fp32: y = act_in(x) W^T + b for tiny M (30 timesteps).
 * Replaces Timesteps/TimestepEmbedding and ResnetBlock2D.time_emb_proj.
 * ------------------------------------------------------------------------------------------ */
int idb_timestep_sinusoid(const float* timesteps, float* out, int32_t n, int32_t dim, void* stream);
int idb_linear_f32(const float* x, const float* w, const float* bias, float* y, int32_t m, int32_t n,
                   int32_t k, int32_t silu_in, void* stream);

/* ------------------------------------------------------------------------------------------
 * conv_in (Cin = 4): fp32 NCHW latents -> operand-dtype NHWC features; `rep` replicates the batch
 * (CFG feeds the same latents twice).  Replaces UNet conv_in / VAE post_quant_conv + conv_in.
 * ------------------------------------------------------------------------------------------ */
int idb_conv_in(const float* x_nchw, const float* w, const float* bias, void* out, int32_t batch,
                int32_t rep, int32_t cin, int32_t h, int32_t w_, int32_t cout, float in_scale,
                const float* pre_w, const float* pre_b, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K8 — classifier-free guidance + DDPMScheduler.step, fp32 (inference_ID-Booth.py:104,138;
 * train_ID-Booth.py:1081).  eps is [2B][HW][C] fp32 (uncond first); latents/noise are NCHW fp32.
 * coef points at 6 floats on the DEVICE: {sqrt_alpha_bar_t, sqrt_beta_bar_t, c_x0, c_x, sigma,
 * guidance_scale}; prediction_type 0 = epsilon, 1 = v_prediction.  x0_out may be NULL.
 * ------------------------------------------------------------------------------------------ */
int idb_cfg_ddpm_step(const float* eps, float* latents, const float* noise, const float* coef,
                      float* x0_out, int32_t batch, int32_t channels, int32_t hw, int32_t cfg,
                      int32_t prediction_type, void* stream);

/* K10 — VaeImageProcessor.postprocess + save_image quantisation:
 * x [B][HW][C] fp32 -> img01 = clamp(x/2+0.5,0,1) (fp32 NHWC, may be NULL) and u8 = floor(255*img01+0.5). */
int idb_postprocess(const float* x, float* img01, uint8_t* u8, int64_t count, void* stream);

/* [B][HW][C] operand dtype -> [B][C][HW] fp32 (API boundary: unet(...)[0], vae.decode(...).sample). */
int idb_nhwc_to_nchw_f32(const void* x, float* out, int32_t batch, int32_t hw, int32_t c, int32_t dtype,
                         void* stream);
int idb_f32_nhwc_to_nchw(const float* x, float* out, int32_t batch, int32_t hw, int32_t c, void* stream);
/* [rows][cols] fp32 -> operand dtype (prompt embeddings). */
int idb_cast_f32(const float* x, void* out, int64_t count, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * AutoencoderKL.encode tail (train_ID-Booth.py:1001-1002): DiagonalGaussianDistribution.sample()/.mode() times
 * vae.config.scaling_factor.  moments: fp32 [batch][hw][2*channels] (mean channels first, then log-variance), the
 * output of the encoder's conv_out with quant_conv folded in; noise: NCHW fp32 [batch][channels][hw] or NULL (mode);
 * latents, and optional mean / logvar (clamped to [-30, 20]) outputs: NCHW fp32.
 * ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------
 * Face align-and-crop warp: cv2.warpAffine(img, M, (out_w, out_h), borderValue) for 8-bit images with
 * OpenCV's defaults (INTER_LINEAR, BORDER_CONSTANT) — utils/detect_align_crop_data.py:180.  src: uint8 [batch][h][w][channels];
 * m_inv: [batch][6] doubles, the INVERSE of each 2x3 matrix (cv::invertAffineTransform, computed on the host in double);
 * coordinates in 10-bit fixed point with a 5-bit sub-pixel fraction, 15-bit integer tap weights: integer-exact.
 * ------------------------------------------------------------------------------------------ */
int idb_warp_affine_u8(const uint8_t* src, int32_t batch, int32_t h, int32_t w, int32_t channels, const double* m_inv,
                       uint8_t* dst, int32_t out_h, int32_t out_w, int32_t border_value, void* stream);

/* ------------------------------------------------------------------------------------------
 * MTCNN face detector (facenet_pytorch P-Net / R-Net / O-Net as used at
 * utils/detect_align_crop_data.py:18-20,99).  fp32 NCHW activations; the cascade's control logic (image pyramid, box generation,
 * NMS, regression) is host code in faceposegenerator_amd/mtcnn.py.
 *   idb_crop_resize_area_u8: out[k] = (F.interpolate(src[image_k, y0:y1, x0:x1], (out_h, out_w), mode="area") - sub) * mul for
 *       boxes [n][5] = {image, y0, y1, x0, x1} (int32 on the device, exclusive ends, clipped by the caller); uint8 NHWC in,
 *       fp32 [n][channels][out_h][out_w] out (upstream imresample + the (x - 127.5) / 128 normalisation)
 *   idb_conv2d_f32: nn.Conv2d(cin, cout, (kh, kw)) (stride 1, no padding) + optional nn.PReLU(cout); also the dense layers
 *       (a kernel as large as the map)
 *   idb_maxpool2d_f32: nn.MaxPool2d(k, stride, ceil_mode=True) over `planes` = batch * channels maps
 *   idb_softmax_pairs_f32: softmax over a channel pair, p1 = softmax(x[b][0:2][i])[1]
 *   idb_nms_mask: the suppression bit matrix of greedy NMS (torchvision.ops.batched_nms as detect_face.py calls it, use_min = 0, and
 *       upstream's nms_numpy "Min" with +1 widths, use_min = 1 / plus_one = 1): boxes fp32 [n][4] SORTED by descending score, optional
 *       image index int32 [n] (boxes of different images never suppress each other); mask uint64 [n][(n+63)/64], bit j of row i set
 *       when box i would remove box j > i.  The greedy scan over the rows stays with the caller (host: a few hundred words).
 * ------------------------------------------------------------------------------------------ */
int idb_crop_resize_area_u8(const uint8_t* src, int32_t batch, int32_t h, int32_t w, int32_t channels, const int32_t* boxes, int32_t n,
                            float* out, int32_t out_h, int32_t out_w, float sub, float mul, void* stream);
int idb_conv2d_f32(const float* x, const float* w, const float* bias, const float* prelu, float* y, int32_t batch, int32_t cin, int32_t h,
                   int32_t w_, int32_t cout, int32_t kh, int32_t kw, void* stream);
int idb_maxpool2d_f32(const float* x, float* y, int32_t planes, int32_t h, int32_t w, int32_t k, int32_t stride, void* stream);
int idb_softmax_pairs_f32(const float* x, float* p1, int32_t batch, int32_t hw, void* stream);
int idb_nms_mask(const float* boxes, const int32_t* image, int32_t n, float thr, int32_t use_min, int32_t plus_one, uint64_t* mask,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) implicit GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 — BASELINE configs[4] "fp8 MFMA weight path" (the 768x768
 * v-prediction checkpoint selected at inference_ID-Booth.py:63-64,103), kernel level: out[m][n] = x_scale * w_scale[n] *
 * sum_k x8[m][k] * w8[n][k] (+ bias, per-sample bias, residual), both operands 8-bit, fp32 accumulation, operand-dtype output.
 * One source: NHWC fp8 [batch][in_h][in_w][channels] (taps 9: 3x3, padding 1; taps 1: the pixel itself), channels % 64 == 0.
 *   idb_quantize_fp8: out8 = e4m3(x * inv_scale), saturating at +-448 (x: bf16/f16, count % 8 == 0)
 *   idb_pack_weight_fp8: fp32 torch layout [cout][cin][ktaps] -> fp8 [cout][ktaps][cin rounded up to 128] (zero columns) with one
 *       scale per output channel, scales[n] = absmax(row n) / 448
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t out_dtype;            /* IDB_BF16 / IDB_F16: dtype of `out` and `residual` */
    int32_t batch, out_h, out_w, stride, n;
    const void* x;                /* fp8 e4m3 NHWC */
    int32_t channels, taps, in_h, in_w, upsample;
    float x_scale;                /* real value = x_scale * fp8 value (per tensor) */
    const void* w;                /* idb_pack_weight_fp8 layout */
    const float* w_scale;         /* [n] */
    const float* bias;
    const float* sample_bias;
    int32_t sample_bias_ld;
    const void* residual;
    void* out;
    int32_t out_ld;
    float* gn_partials;           /* optional out, as idb_gemm_desc.gn_partials: written by the kernel's own epilogue when the column tiles
                                     hold whole groups (n % 160 == 0, 160 % (n / gn_groups) == 0, out_h*out_w % 64 == 0, out_ld == n),
                                     else by an extra statistics launch */
    int32_t gn_groups;
} idb_gemm_fp8_desc;
int idb_quantize_fp8(const void* x, void* out, int64_t count, float inv_scale, int32_t dtype, void* stream);
/* idb_groupnorm (two-launch form) writing its output as fp8 e4m3 of y * out_inv_scale, saturating: the activation operand of
 * idb_gemm_fp8 straight from the normalisation — GroupNorm+SiLU outputs are bounded, so a fixed per-tensor scale serves. */
int idb_groupnorm_fp8(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw, int32_t groups, float eps,
                      const float* gamma, const float* beta, int32_t silu, void* out8, float out_inv_scale, int32_t dtype,
                      void* workspace, size_t workspace_bytes, const float* partials_in, int32_t partials_chunks, void* stream);
int idb_pack_weight_fp8(const float* src, void* dst, float* scales, int32_t cout, int32_t cin, int32_t ktaps, void* stream);
int idb_gemm_fp8(const idb_gemm_fp8_desc* d, void* stream);

int idb_vae_sample(const float* moments, const float* noise, float scale, float* latents, float* mean_out,
                   float* logvar_out, int32_t batch, int32_t channels, int32_t hw, void* stream);

#ifdef __cplusplus
}
#endif
#endif

// K1 (norm part) / K6 — GroupNorm(+SiLU) over NHWC, LayerNorm over rows, row softmax.  HBM-bound:
// 16-byte vector accesses, fp32 statistics, deterministic two-level reductions (no atomics).
//
// Upstream ops replaced: nn.GroupNorm(32, C, eps) [+ F.silu] in diffusers ResnetBlock2D /
// Transformer2DModel.norm / conv_norm_out / VAE Attention.group_norm; nn.LayerNorm(C) x3 in
// BasicTransformerBlock; the softmax inside the VAE mid-block attention.
#include "idb_common.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAXCOLS = 2;      // chunk columns per thread: C <= 2 * 256 * 8 = 4096 channels

__host__ __device__ inline int gn_nchunks(int batch, int hw) {
    int per = 2048 / (batch > 0 ? batch : 1);
    if (per < 1) per = 1;
    int byhw = (hw + 63) / 64;
    if (byhw < 1) byhw = 1;
    int n = per < byhw ? per : byhw;
    return n > 64 ? 64 : n;
}

struct GnGeom {
    int C, C0, C1, Ct, TPR, PR, HW, groups, cpg, nchunks;
};

template <typename T>
__device__ __forceinline__ void gn_load8(const T* x0, const T* x1, const GnGeom& g, long long pix, int cc, float* v) {
    typename Op<T>::v8 raw;
    const int c = cc * 8;
    if (c < g.C0) raw = *(const typename Op<T>::v8*)(x0 + pix * g.C0 + c);
    else raw = *(const typename Op<T>::v8*)(x1 + pix * g.C1 + (c - g.C0));
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = to_f32<T>(raw[e]);
}

// pass 1: per (sample, pixel-chunk) partial {sum, sumsq} per group
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* x0, const T* x1, GnGeom g, float* partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;            // [PR][C] sums then [PR][C] sumsq
    const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int tc = tid % g.TPR, tr = tid / g.TPR;
    const int len = (g.HW + g.nchunks - 1) / g.nchunks;
    const int p0 = chunk * len, p1 = min(p0 + len, g.HW);
    float s[GN_MAXCOLS][8], ss[GN_MAXCOLS][8];
#pragma unroll
    for (int k = 0; k < GN_MAXCOLS; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[k][e] = ss[k][e] = 0.f;
    if (tr < g.PR) {
        for (int p = p0 + tr; p < p1; p += g.PR) {
            const long long pix = (long long)b * g.HW + p;
#pragma unroll
            for (int k = 0; k < GN_MAXCOLS; ++k) {
                const int cc = tc + k * g.TPR;
                if (cc < g.Ct) {
                    float v[8];
                    gn_load8<T>(x0, x1, g, pix, cc, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s[k][e] += v[e];
                        ss[k][e] += v[e] * v[e];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < GN_MAXCOLS; ++k) {
            const int cc = tc + k * g.TPR;
            if (cc < g.Ct) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[tr * g.C + cc * 8 + e] = s[k][e];
                    red[(g.PR + tr) * g.C + cc * 8 + e] = ss[k][e];
                }
            }
        }
    }
    __syncthreads();
    for (int grp = tid; grp < g.groups; grp += GN_THREADS) {
        float a = 0.f, q = 0.f;
        for (int r = 0; r < g.PR; ++r)
            for (int c = grp * g.cpg; c < (grp + 1) * g.cpg; ++c) {
                a += red[r * g.C + c];
                q += red[(g.PR + r) * g.C + c];
            }
        float* dst = partial + (((long long)b * g.nchunks + chunk) * g.groups + grp) * 2;
        dst[0] = a;
        dst[1] = q;
    }
}

// pass 2: y = (x - mean) * rstd * gamma + beta  [-> SiLU]
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* x0, const T* x1, GnGeom g, const float* partial,
                                                              const float* gamma, const float* beta, float eps, int silu,
                                                              T* out, int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = (float*)smem_raw;             // [C] scale, [C] shift, [groups] mean, [groups] rstd
    float* sh = sc + g.C;
    float* gm = sh + g.C;
    float* gr = gm + g.groups;
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int grp = tid; grp < g.groups; grp += GN_THREADS) {
        double a = 0.0, q = 0.0;
        for (int ch = 0; ch < g.nchunks; ++ch) {
            const float* src = partial + (((long long)b * g.nchunks + ch) * g.groups + grp) * 2;
            a += (double)src[0];
            q += (double)src[1];
        }
        const double cnt = (double)g.HW * g.cpg;
        const double mean = a / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        gm[grp] = (float)mean;
        gr[grp] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = tid; c < g.C; c += GN_THREADS) {
        const int grp = c / g.cpg;
        const float k = gr[grp] * gamma[c];
        sc[c] = k;
        sh[c] = beta[c] - gm[grp] * k;
    }
    __syncthreads();
    const int tc = tid % g.TPR, tr = tid / g.TPR;
    if (tr >= g.PR) return;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, g.HW);
    float ks[GN_MAXCOLS][8], kh[GN_MAXCOLS][8];
#pragma unroll
    for (int k = 0; k < GN_MAXCOLS; ++k) {
        const int cc = tc + k * g.TPR;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ks[k][e] = cc < g.Ct ? sc[cc * 8 + e] : 0.f;
            kh[k][e] = cc < g.Ct ? sh[cc * 8 + e] : 0.f;
        }
    }
    for (int p = p0 + tr; p < p1; p += g.PR) {
        const long long pix = (long long)b * g.HW + p;
#pragma unroll
        for (int k = 0; k < GN_MAXCOLS; ++k) {
            const int cc = tc + k * g.TPR;
            if (cc < g.Ct) {
                float v[8];
                gn_load8<T>(x0, x1, g, pix, cc, v);
                typename Op<T>::v8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = v[e] * ks[k][e] + kh[k][e];
                    if (silu) y = silu_f(y);
                    o[e] = from_f32<T>(y);
                }
                *(typename Op<T>::v8*)(out + pix * g.C + cc * 8) = o;
            }
        }
    }
}

// LayerNorm: one wave per row, row held in registers (C <= 1536), exact two-pass variance.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, T* out, long long rows, int C, float eps,
                                                        const float* gamma, const float* beta) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Ct = C >> 3;
    float v[3][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cc = lane + i * 64;
        if (cc < Ct) {
            const typename Op<T>::v8 raw = *(const typename Op<T>::v8*)(x + row * C + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = to_f32<T>(raw[e]);
                sum += v[i][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cc = lane + i * 64;
        if (cc < Ct) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cc = lane + i * 64;
        if (cc < Ct) {
            typename Op<T>::v8 o;
            const f32x4 g0 = *(const f32x4*)(gamma + cc * 8), g1 = *(const f32x4*)(gamma + cc * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(beta + cc * 8), b1 = *(const f32x4*)(beta + cc * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = from_f32<T>((v[i][e] - mean) * rstd * g0[e] + b0[e]);
                o[e + 4] = from_f32<T>((v[i][e + 4] - mean) * rstd * g1[e] + b1[e]);
            }
            *(typename Op<T>::v8*)(out + row * C + cc * 8) = o;
        }
    }
}

// In-place row softmax, one block per row, three passes over an L2-resident row.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* x, int cols) {
    __shared__ float red[4];
    T* row = x + (long long)blockIdx.x * cols;
    const int tid = threadIdx.x, Ct = cols >> 3;
    float mx = -INFINITY;
    for (int cc = tid; cc < Ct; cc += 256) {
        const typename Op<T>::v8 raw = *(const typename Op<T>::v8*)(row + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, to_f32<T>(raw[e]));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int cc = tid; cc < Ct; cc += 256) {
        const typename Op<T>::v8 raw = *(const typename Op<T>::v8*)(row + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf(to_f32<T>(raw[e]) - mx);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int cc = tid; cc < Ct; cc += 256) {
        typename Op<T>::v8 raw = *(const typename Op<T>::v8*)(row + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[e] = from_f32<T>(__expf(to_f32<T>(raw[e]) - mx) * inv);
        *(typename Op<T>::v8*)(row + cc * 8) = raw;
    }
}

template <typename T>
int run_groupnorm(const void* x0, int c0, const void* x1, int c1, int batch, int hw, int groups, float eps,
                  const float* gamma, const float* beta, int silu, void* out, void* ws, hipStream_t st) {
    GnGeom g;
    g.C0 = c0;
    g.C1 = c1;
    g.C = c0 + c1;
    g.Ct = g.C / 8;
    g.TPR = g.Ct < GN_THREADS ? g.Ct : GN_THREADS;
    g.PR = GN_THREADS / g.TPR;
    g.HW = hw;
    g.groups = groups;
    g.cpg = g.C / groups;
    g.nchunks = gn_nchunks(batch, hw);
    const size_t lds1 = (size_t)2 * g.PR * g.C * sizeof(float);
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(g.nchunks, batch), dim3(GN_THREADS), lds1, st, (const T*)x0,
                       (const T*)x1, g, (float*)ws);
    IDB_CHECK_LAUNCH("idb_groupnorm(stats)");
    int target_blocks = 2048 / batch;
    if (target_blocks < 1) target_blocks = 1;
    int ppb = (hw + target_blocks - 1) / target_blocks;
    if (ppb < g.PR * 4) ppb = g.PR * 4;
    const int nblk = (hw + ppb - 1) / ppb;
    const size_t lds2 = (size_t)(2 * g.C + 2 * groups) * sizeof(float);
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nblk, batch), dim3(GN_THREADS), lds2, st, (const T*)x0, (const T*)x1,
                       g, (const float*)ws, gamma, beta, eps, silu, (T*)out, ppb);
    IDB_CHECK_LAUNCH("idb_groupnorm(apply)");
    return IDB_OK;
}

}  // namespace

extern "C" size_t idb_groupnorm_workspace_bytes(int32_t batch, int32_t hw, int32_t groups) {
    if (batch <= 0 || hw <= 0 || groups <= 0) return 0;
    return (size_t)batch * gn_nchunks(batch, hw) * groups * 2 * sizeof(float);
}

extern "C" int idb_groupnorm(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw,
                             int32_t groups, float eps, const float* gamma, const float* beta, int32_t silu, void* out,
                             int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    IDB_REQUIRE(idb_is_operand_dtype(dtype), "idb_groupnorm: dtype must be bf16/f16");
    IDB_REQUIRE(x0 && out && gamma && beta && idb_aligned16(x0) && idb_aligned16(out), "idb_groupnorm: null/unaligned pointer");
    IDB_REQUIRE(batch > 0 && hw > 0 && groups > 0 && c0 > 0 && c1 >= 0, "idb_groupnorm: bad dims");
    IDB_REQUIRE((c1 == 0) == (x1 == nullptr), "idb_groupnorm: x1/c1 mismatch");
    IDB_REQUIRE(c1 == 0 || idb_aligned16(x1), "idb_groupnorm: x1 unaligned");
    const int C = c0 + c1;
    IDB_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && C % groups == 0, "idb_groupnorm: channels %d+%d / groups %d unsupported", c0, c1, groups);
    IDB_REQUIRE(C / 8 <= GN_MAXCOLS * GN_THREADS, "idb_groupnorm: too many channels (%d)", C);
    IDB_REQUIRE((size_t)(2 * (GN_THREADS / (C / 8 < GN_THREADS ? C / 8 : GN_THREADS)) * C) * 4 <= 64 * 1024,
                "idb_groupnorm: LDS budget exceeded for C=%d", C);
    const size_t need = idb_groupnorm_workspace_bytes(batch, hw, groups);
    IDB_REQUIRE(workspace && workspace_bytes >= need, "idb_groupnorm: workspace too small (%zu < %zu)", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    return dtype == IDB_BF16
               ? run_groupnorm<__bf16>(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, silu, out, workspace, st)
               : run_groupnorm<_Float16>(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, silu, out, workspace, st);
}

extern "C" int idb_layernorm(const void* x, void* out, int64_t rows, int32_t c, float eps, const float* gamma,
                             const float* beta, int32_t dtype, void* stream) {
    IDB_REQUIRE(idb_is_operand_dtype(dtype), "idb_layernorm: dtype must be bf16/f16");
    IDB_REQUIRE(x && out && gamma && beta && idb_aligned16(x) && idb_aligned16(out) && idb_aligned16(gamma) && idb_aligned16(beta),
                "idb_layernorm: null/unaligned pointer");
    IDB_REQUIRE(rows > 0 && c > 0 && c % 8 == 0 && c <= 1536, "idb_layernorm: C=%d unsupported (multiple of 8, <= 1536)", c);
    const long long blocks = (rows + 3) / 4;
    IDB_REQUIRE(blocks < (1LL << 31), "idb_layernorm: too many rows");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == IDB_BF16)
        hipLaunchKernelGGL((layernorm_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)x, (__bf16*)out,
                           (long long)rows, c, eps, gamma, beta);
    else
        hipLaunchKernelGGL((layernorm_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)x,
                           (_Float16*)out, (long long)rows, c, eps, gamma, beta);
    IDB_CHECK_LAUNCH("idb_layernorm");
    return IDB_OK;
}

extern "C" int idb_softmax_rows(void* x, int64_t rows, int32_t cols, int32_t dtype, void* stream) {
    IDB_REQUIRE(idb_is_operand_dtype(dtype), "idb_softmax_rows: dtype must be bf16/f16");
    IDB_REQUIRE(x && idb_aligned16(x) && rows > 0 && rows < (1LL << 31) && cols > 0 && cols % 8 == 0, "idb_softmax_rows: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == IDB_BF16)
        hipLaunchKernelGGL((softmax_rows_kernel<__bf16>), dim3((unsigned)rows), dim3(256), 0, st, (__bf16*)x, cols);
    else
        hipLaunchKernelGGL((softmax_rows_kernel<_Float16>), dim3((unsigned)rows), dim3(256), 0, st, (_Float16*)x, cols);
    IDB_CHECK_LAUNCH("idb_softmax_rows");
    return IDB_OK;
}

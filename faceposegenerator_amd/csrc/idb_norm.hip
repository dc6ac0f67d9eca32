// K1 (norm part) / K6 — GroupNorm(+SiLU) over NHWC, LayerNorm over rows, row softmax.  HBM-bound:
// 16-byte vector accesses, fp32 statistics, deterministic two-level reductions (no atomics).
//
// Upstream ops replaced: nn.GroupNorm(32, C, eps) [+ F.silu] in diffusers ResnetBlock2D /
// Transformer2DModel.norm / conv_norm_out / VAE Attention.group_norm; nn.LayerNorm(C) x3 in
// BasicTransformerBlock; the softmax inside the VAE mid-block attention.
#include "idb_common.h"
#include <stdlib.h>

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_GSLOT = 4;        // groups one 8-channel chunk can touch (cpg >= 2)
constexpr int GN_MAXCHUNKS = 64;

// Geometry shared by the two passes.  Channels are cut into slices of SW channels (a multiple of both the
// group width and the 8-channel vector), so a workgroup owns whole groups: grid = (pixel chunks, slices, B).
struct GnGeom {
    int C, C0, C1, HW, groups, cpg;
    int SW, cols, PR, nslices, gps, nchunks, chunk_len;
};

inline int gn_gcd(int a, int b) { return b ? gn_gcd(b, a % b) : a; }

inline GnGeom gn_geometry(int c0, int c1, int batch, int hw, int groups) {
    GnGeom g;
    g.C0 = c0; g.C1 = c1; g.C = c0 + c1; g.HW = hw; g.groups = groups; g.cpg = g.C / groups;
    int sw = g.cpg / gn_gcd(g.cpg, 8) * 8;                       // lcm(cpg, 8)
    while (sw / 8 < 8 && g.C % (sw * 2) == 0) sw *= 2;            // at least 8 vector columns when possible
    {
        // large tensors: slices of whole 128-byte lines (lcm(cpg, 64) channels) when that fits a workgroup — an 80-channel slice is a
        // 160-byte piece of every pixel row, i.e. 2-3 partially used lines per piece, each line shared with the neighbouring slice's
        // workgroup.  B_eff 128 (tools/bench_norm.py): 64x64x320 293 -> 225 us (3.4 -> 4.5 TB/s incl. the statistics re-read),
        // 32x32x640 139 -> 122, 16x16x1280 66 -> 53; batch 64 15.08 -> 15.28 images/s.  At B_eff 2 the fewer, fatter workgroups lose
        // (14.5 -> 20.6 us) and up to batch 8 it is a wash: only from IDB_GN_ALIGN elements on (default 16 Mi = 32 MB; 0: never).
        static const long long env_align = [] { const char* e = getenv("IDB_GN_ALIGN"); return e ? atoll(e) : 16LL << 20; }();
        const int sa = g.cpg / gn_gcd(g.cpg, 64) * 64;
        if (env_align > 0 && (long long)batch * hw * g.C >= env_align && g.C0 % 64 == 0 && g.C1 % 64 == 0 && g.C % sa == 0 && sa / 8 <= GN_THREADS &&
            sa / g.cpg <= 64)
            sw = sa;
    }
    g.SW = sw; g.cols = sw / 8; g.PR = GN_THREADS / g.cols; g.nslices = g.C / sw; g.gps = sw / g.cpg;
    // enough pixel chunks that the grid has >= ~1024 workgroups, each with >= 2 pixels per thread row
    int want = (1024 + batch * g.nslices - 1) / (batch * g.nslices);
    int by_hw = (hw + 2 * g.PR - 1) / (2 * g.PR);
    int n = want < by_hw ? want : by_hw;
    if (n < 1) n = 1;
    if (n > GN_MAXCHUNKS) n = GN_MAXCHUNKS;
    g.nchunks = n;
    g.chunk_len = (hw + n - 1) / n;
    return g;
}

// Loads are issued unconditionally from clamped addresses and in batches (GN_BATCH per thread) so that a thread has
// several 16-byte loads in flight: a load inside a divergent branch or a data-dependent loop gets an s_waitcnt vmcnt(0)
// right behind it, which serialises the memory round trips — at batch 1 these kernels are latency-bound, not HBM-bound.
constexpr int GN_BATCH = 4;

template <typename T>
__device__ __forceinline__ typename Op<T>::v8 gn_load_raw(const T* x0, const T* x1, const GnGeom& g, long long pix, int c) {
    const T* src = c < g.C0 ? x0 + pix * g.C0 + c : x1 + pix * g.C1 + (c - g.C0);      // address select, no branch
    return *(const typename Op<T>::v8*)src;
}

// pass 1: partial {sum, sumsq} per (sample, pixel chunk, group); deterministic (fixed-order LDS + shuffle trees)
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* x0, const T* x1, GnGeom g, float* partial) {
    __shared__ float part[GN_THREADS][GN_GSLOT][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, slice = blockIdx.y, b = blockIdx.z;
    const int col = tid % g.cols, row = tid / g.cols;
    const int c = slice * g.SW + col * 8;                 // first channel of this thread's vector
    const int p0 = chunk * g.chunk_len, p1 = min(p0 + g.chunk_len, g.HW);
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    if (row < g.PR) {
        for (int pb = p0 + row; pb < p1; pb += GN_BATCH * g.PR) {
            typename Op<T>::v8 raw[GN_BATCH];
#pragma unroll
            for (int u = 0; u < GN_BATCH; ++u)
                raw[u] = gn_load_raw<T>(x0, x1, g, (long long)b * g.HW + min(pb + u * g.PR, p1 - 1), c);
#pragma unroll
            for (int u = 0; u < GN_BATCH; ++u) {
                if (pb + u * g.PR < p1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = to_f32<T>(raw[u][e]);
                        s[e] += v;
                        ss[e] += v * v;
                    }
                }
            }
        }
    }
    // fold the 8 channels into the (<= GN_GSLOT) groups they belong to
    const int g_first = c / g.cpg;
    float gs[GN_GSLOT], gq[GN_GSLOT];
#pragma unroll
    for (int k = 0; k < GN_GSLOT; ++k) gs[k] = gq[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = (c + e) / g.cpg - g_first;
#pragma unroll
        for (int kk = 0; kk < GN_GSLOT; ++kk)
            if (kk == k) {
                gs[kk] += s[e];
                gq[kk] += ss[e];
            }
    }
#pragma unroll
    for (int k = 0; k < GN_GSLOT; ++k) {
        part[tid][k][0] = gs[k];
        part[tid][k][1] = gq[k];
    }
    __syncthreads();
    const int nact = g.cols * g.PR;
    const int slice_g0 = slice * g.gps;
    for (int gl = wave; gl < g.gps; gl += GN_THREADS / 64) {
        const int ga = slice_g0 + gl;
        float a = 0.f, q = 0.f;
        for (int t = lane; t < nact; t += 64) {
            const int tc = slice * g.SW + (t % g.cols) * 8;
            const int k = ga - tc / g.cpg;
            if (k >= 0 && k < GN_GSLOT) {
                a += part[t][k][0];
                q += part[t][k][1];
            }
        }
        a = wave_sum(a);
        q = wave_sum(q);
        if (lane == 0) {
            float* dst = partial + (((long long)b * g.nchunks + chunk) * g.groups + ga) * 2;
            dst[0] = a;
            dst[1] = q;
        }
    }
}

// pass 2: y = (x - mean) * rstd * gamma + beta  [-> SiLU]; grid = (pixel blocks, slices, B)
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* x0, const T* x1, GnGeom g, const float* partial,
                                                              const float* gamma, const float* beta, float eps, int silu,
                                                              T* out, int pix_per_block, float out8 = 0.f) {
    __shared__ float gmean[64], grstd[64];
    const int tid = threadIdx.x;
    const int slice = blockIdx.y, b = blockIdx.z;
    const int col = tid % g.cols, row = tid / g.cols;
    const int c = slice * g.SW + col * 8;
    const bool active = row < g.PR;
    // affine parameters of this thread's 8 channels: requested first so that they travel with the partial sums
    const int cl = active ? c : 0;
    const f32x4 ga0 = *(const f32x4*)(gamma + cl), ga1 = *(const f32x4*)(gamma + cl + 4);
    const f32x4 be0 = *(const f32x4*)(beta + cl), be1 = *(const f32x4*)(beta + cl + 4);
    // group statistics of this slice: 8 lanes per group sum the pixel-chunk partials (<= 8 each, all loads in flight
    // at once), then a shuffle tree
    {
        const int gl = tid >> 3, sub = tid & 7;
        for (int g0 = 0; g0 < g.gps; g0 += GN_THREADS / 8) {
            const int gi = g0 + gl;
            const int ga = slice * g.gps + min(gi, g.gps - 1);
            f32x2 pv[GN_MAXCHUNKS / 8];
#pragma unroll
            for (int u = 0; u < GN_MAXCHUNKS / 8; ++u) {
                const int ch = min(sub + 8 * u, g.nchunks - 1);
                pv[u] = *(const f32x2*)(partial + (((long long)b * g.nchunks + ch) * g.groups + ga) * 2);
            }
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int u = 0; u < GN_MAXCHUNKS / 8; ++u) {
                if (gi < g.gps && sub + 8 * u < g.nchunks) {
                    a += pv[u][0];
                    q += pv[u][1];
                }
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (gi < g.gps && sub == 0) {
                const double cnt = (double)g.HW * g.cpg;
                const double mean = (double)a / cnt;
                double var = (double)q / cnt - mean * mean;
                if (var < 0.0) var = 0.0;
                gmean[gi] = (float)mean;
                grstd[gi] = (float)(1.0 / sqrt(var + (double)eps));
            }
        }
    }
    __syncthreads();
    if (!active) return;
    float ks[8], kh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int gi = (c + e) / g.cpg - slice * g.gps;
        const float k = grstd[gi] * (e < 4 ? ga0[e & 3] : ga1[e & 3]);
        ks[e] = k;
        kh[e] = (e < 4 ? be0[e & 3] : be1[e & 3]) - gmean[gi] * k;
    }
    const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, g.HW);
    for (int pb = p0 + row; pb < p1; pb += GN_BATCH * g.PR) {
        typename Op<T>::v8 raw[GN_BATCH];
#pragma unroll
        for (int u = 0; u < GN_BATCH; ++u)
            raw[u] = gn_load_raw<T>(x0, x1, g, (long long)b * g.HW + min(pb + u * g.PR, p1 - 1), c);
#pragma unroll
        for (int u = 0; u < GN_BATCH; ++u) {
            const int p = pb + u * g.PR;
            if (p < p1) {
                typename Op<T>::v8 o;
                float yv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = to_f32<T>(raw[u][e]) * ks[e] + kh[e];
                    if (silu) y = silu_f(y);
                    yv[e] = y;
                    o[e] = from_f32<T>(y);
                }
                if (out8 > 0.f) {                       // fp8 e4m3 output (idb_groupnorm_fp8): y * out8, saturating, 8 bytes per thread
#pragma unroll
                    for (int e = 0; e < 8; ++e) yv[e] = fminf(fmaxf(yv[e] * out8, -448.f), 448.f);
                    unsigned lo = __builtin_amdgcn_cvt_pk_fp8_f32(yv[0], yv[1], 0u, false);
                    lo = __builtin_amdgcn_cvt_pk_fp8_f32(yv[2], yv[3], lo, true);
                    unsigned hi = __builtin_amdgcn_cvt_pk_fp8_f32(yv[4], yv[5], 0u, false);
                    hi = __builtin_amdgcn_cvt_pk_fp8_f32(yv[6], yv[7], hi, true);
                    *(u32x2*)((unsigned char*)out + ((long long)b * g.HW + p) * g.C + c) = (u32x2){lo, hi};
                } else {
                    *(typename Op<T>::v8*)(out + ((long long)b * g.HW + p) * g.C + c) = o;
                }
            }
        }
    }
}

// Single-launch GroupNorm for grids that are co-resident on the chip (the whole batch-1 UNet): each workgroup keeps its
// (pixel chunk x channel slice) in registers, publishes its partial {sum, sumsq} per group, waits until the other pixel
// chunks of its (sample, slice) have published theirs, then normalises from registers: ONE read of the tensor, one launch
// (the two-pass form costs 5.4 + 6.2 us per layer at batch 1, mostly launch ramp and memory round trips).
// Hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"): partials are stored with agent-scope relaxed
// atomic stores (sc1), every storing wave drains vmcnt, workgroup barrier, ONE lane adds to the agent-scope arrival
// counter; the same lane polls the counter with agent-scope relaxed loads, joins a workgroup barrier, and every wave then
// reads the partials with agent-scope relaxed loads (sc1: never served from this CU's L1).  Summation order is fixed
// (chunk index), so results do not depend on arrival order.  The last workgroup to leave resets both counters, so the
// counter array stays zero between launches.  The poll is bounded: the host only takes this path for grids of at most
// GN_SYNC_MAXBLOCKS workgroups (all resident at once), and a poll that still runs out gives up rather than hanging.
constexpr int GN_SYNC_MAXIT = 4;
constexpr int GN_SYNC_MAXBLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_sync_kernel(const T* x0, const T* x1, GnGeom g, float* partial, int* counters,
                                                             const float* gamma, const float* beta, float eps, int silu, T* out) {
    __shared__ float part[GN_THREADS][GN_GSLOT][2];
    __shared__ float gmean[64], grstd[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, slice = blockIdx.y, b = blockIdx.z;
    const int col = tid % g.cols, row = tid / g.cols;
    const int c = slice * g.SW + col * 8;
    const int p0 = chunk * g.chunk_len, p1 = min(p0 + g.chunk_len, g.HW);
    const bool active = row < g.PR;
    typename Op<T>::v8 keep[GN_SYNC_MAXIT];
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    const int cl = active ? c : slice * g.SW;
#pragma unroll
    for (int it = 0; it < GN_SYNC_MAXIT; ++it)
        keep[it] = gn_load_raw<T>(x0, x1, g, (long long)b * g.HW + min(p0 + row + it * g.PR, p1 - 1), cl);
#pragma unroll
    for (int it = 0; it < GN_SYNC_MAXIT; ++it) {
        if (active && p0 + row + it * g.PR < p1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = to_f32<T>(keep[it][e]);
                s[e] += v;
                ss[e] += v * v;
            }
        }
    }
    const int g_first = c / g.cpg;
    float gs[GN_GSLOT], gq[GN_GSLOT];
#pragma unroll
    for (int k = 0; k < GN_GSLOT; ++k) gs[k] = gq[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = (c + e) / g.cpg - g_first;
#pragma unroll
        for (int kk = 0; kk < GN_GSLOT; ++kk)
            if (kk == k) {
                gs[kk] += s[e];
                gq[kk] += ss[e];
            }
    }
#pragma unroll
    for (int k = 0; k < GN_GSLOT; ++k) {
        part[tid][k][0] = gs[k];
        part[tid][k][1] = gq[k];
    }
    __syncthreads();
    const int nact = g.cols * g.PR;
    const int slice_g0 = slice * g.gps;
    for (int gl = wave; gl < g.gps; gl += GN_THREADS / 64) {
        const int ga = slice_g0 + gl;
        float a = 0.f, q = 0.f;
        for (int t = lane; t < nact; t += 64) {
            const int tc = slice * g.SW + (t % g.cols) * 8;
            const int k = ga - tc / g.cpg;
            if (k >= 0 && k < GN_GSLOT) {
                a += part[t][k][0];
                q += part[t][k][1];
            }
        }
        a = wave_sum(a);
        q = wave_sum(q);
        if (lane == 0) {
            float* dst = partial + (((long long)b * g.nchunks + chunk) * g.groups + ga) * 2;
            __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    int* cnt = counters + ((long long)b * g.nslices + slice) * 2;
    if (g.nchunks > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.nchunks && ++spins < (1 << 22))
                __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    {
        const int gl = tid >> 3, sub = tid & 7;
        for (int g0 = 0; g0 < g.gps; g0 += GN_THREADS / 8) {
            const int gi = g0 + gl;
            float a = 0.f, q = 0.f;
            if (gi < g.gps) {
                const int ga = slice_g0 + gi;
                for (int ch = sub; ch < g.nchunks; ch += 8) {
                    float* src = partial + (((long long)b * g.nchunks + ch) * g.groups + ga) * 2;
                    a += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (gi < g.gps && sub == 0) {
                const double cnt_el = (double)g.HW * g.cpg;
                const double mean = (double)a / cnt_el;
                double var = (double)q / cnt_el - mean * mean;
                if (var < 0.0) var = 0.0;
                gmean[gi] = (float)mean;
                grstd[gi] = (float)(1.0 / sqrt(var + (double)eps));
            }
        }
    }
    __syncthreads();
    if (g.nchunks > 1 && tid == 0) {
        // departures: the partials of this launch have been read by this workgroup; the last one to leave re-arms the pair
        if (__hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.nchunks - 1) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!active) return;
    float ks[8], kh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int gi = (c + e) / g.cpg - slice_g0;
        const float k = grstd[gi] * gamma[c + e];
        ks[e] = k;
        kh[e] = beta[c + e] - gmean[gi] * k;
    }
#pragma unroll
    for (int it = 0; it < GN_SYNC_MAXIT; ++it) {
        const int p = p0 + row + it * g.PR;
        if (p < p1) {
            typename Op<T>::v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = to_f32<T>(keep[it][e]) * ks[e] + kh[e];
                if (silu) y = silu_f(y);
                o[e] = from_f32<T>(y);
            }
            *(typename Op<T>::v8*)(out + ((long long)b * g.HW + p) * g.C + c) = o;
        }
    }
}

// LayerNorm: one wave per row, row held in registers (C <= 1536 = NCH x 64 lanes x 8), exact two-pass variance.  The row
// and the affine parameters are requested together, from clamped addresses, so that the kernel is ONE memory round trip.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, T* out, long long rows, int C, float eps,
                                                        const float* gamma, const float* beta) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Ct = C >> 3;
    typename Op<T>::v8 raw[NCH];
    f32x4 g0[NCH], g1[NCH], b0[NCH], b1[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = min(lane + i * 64, Ct - 1);
        raw[i] = *(const typename Op<T>::v8*)(x + row * C + cc * 8);
        g0[i] = *(const f32x4*)(gamma + cc * 8);
        g1[i] = *(const f32x4*)(gamma + cc * 8 + 4);
        b0[i] = *(const f32x4*)(beta + cc * 8);
        b1[i] = *(const f32x4*)(beta + cc * 8 + 4);
    }
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool in = lane + i * 64 < Ct;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] = in ? to_f32<T>(raw[i][e]) : 0.f;
            sum += v[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (lane + i * 64 < Ct) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = lane + i * 64;
        if (cc < Ct) {
            typename Op<T>::v8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = from_f32<T>((v[i][e] - mean) * rstd * g0[i][e] + b0[i][e]);
                o[e + 4] = from_f32<T>((v[i][e + 4] - mean) * rstd * g1[i][e] + b1[i][e]);
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
                  const float* gamma, const float* beta, int silu, void* out, void* ws, int* sync, int sync_len, const float* pin,
                  int pin_chunks, hipStream_t st, float out8 = 0.f) {
    GnGeom g = gn_geometry(c0, c1, batch, hw, groups);
    if (pin) {                                   // statistics came with the tensor (idb_gemm_desc.gn_partials): normalise only
        g.nchunks = pin_chunks;
        g.chunk_len = 64;
        int want = (2048 + batch * g.nslices - 1) / (batch * g.nslices);
        int ppb = (hw + want - 1) / want;
        if (ppb < g.PR * 2) ppb = g.PR * 2;
        const int nblk = (hw + ppb - 1) / ppb;
        hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nblk, g.nslices, batch), dim3(GN_THREADS), 0, st, (const T*)x0, (const T*)x1, g,
                           pin, gamma, beta, eps, silu, (T*)out, ppb, out8);
        IDB_CHECK_LAUNCH("idb_groupnorm(apply)");
        return IDB_OK;
    }
    {
        // single launch when the caller passes hand-off counters and every workgroup of the grid is resident at once (see
        // gn_sync_kernel).  Measured in the sampling loop (batch 1, A/B on one box): 6.05 images/s against 6.28 for the
        // two-launch form — the hand-off is four dependent agent-scope round trips (publish, arrive, poll, read partials),
        // dearer on this chip than a second launch (1.8 us floor inside a graph); the engine therefore passes no counters
        // unless IDB_GN_SYNC=1.  A one-workgroup-per-(sample, slice) form that keeps feature maps up to 32x32 in registers
        // (one launch, one read) measured equal to the two-launch form (6.285 vs 6.284: 32-64 workgroups cannot pull their
        // slices faster than 512 can pull them twice) and was dropped.
        GnGeom f = g;
        f.nchunks = (hw + GN_SYNC_MAXIT * f.PR - 1) / (GN_SYNC_MAXIT * f.PR);
        f.chunk_len = (hw + f.nchunks - 1) / f.nchunks;
        const long long blocks = (long long)f.nchunks * f.nslices * batch;
        if (sync && f.nchunks <= GN_MAXCHUNKS && blocks <= GN_SYNC_MAXBLOCKS && 2LL * batch * f.nslices <= sync_len &&
            (f.chunk_len + f.PR - 1) / f.PR <= GN_SYNC_MAXIT) {
            hipLaunchKernelGGL((gn_sync_kernel<T>), dim3(f.nchunks, f.nslices, batch), dim3(GN_THREADS), 0, st, (const T*)x0,
                               (const T*)x1, f, (float*)ws, sync, gamma, beta, eps, silu, (T*)out);
            IDB_CHECK_LAUNCH("idb_groupnorm(sync)");
            return IDB_OK;
        }
    }
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(g.nchunks, g.nslices, batch), dim3(GN_THREADS), 0, st, (const T*)x0,
                       (const T*)x1, g, (float*)ws);
    IDB_CHECK_LAUNCH("idb_groupnorm(stats)");
    int want = (2048 + batch * g.nslices - 1) / (batch * g.nslices);
    int ppb = (hw + want - 1) / want;
    if (ppb < g.PR * 2) ppb = g.PR * 2;
    const int nblk = (hw + ppb - 1) / ppb;
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nblk, g.nslices, batch), dim3(GN_THREADS), 0, st, (const T*)x0,
                       (const T*)x1, g, (const float*)ws, gamma, beta, eps, silu, (T*)out, ppb, out8);
    IDB_CHECK_LAUNCH("idb_groupnorm(apply)");
    return IDB_OK;
}

}  // namespace

int idb_launch_gn_stats64(const void* x, int c, int batch, int hw, int groups, float* partial, int dtype, hipStream_t st) {
    IDB_REQUIRE(x && partial && hw % 64 == 0 && hw / 64 <= GN_MAXCHUNKS && c % 8 == 0 && c % groups == 0 && c / groups >= 2,
                "idb_gemm: gn_partials unsupported for hw=%d c=%d groups=%d", hw, c, groups);
    GnGeom g = gn_geometry(c, 0, batch, hw, groups);
    IDB_REQUIRE(g.cols <= GN_THREADS && g.gps <= 64, "idb_gemm: gn_partials unsupported geometry");
    g.nchunks = hw / 64;
    g.chunk_len = 64;
    if (dtype == IDB_BF16)
        hipLaunchKernelGGL((gn_stats_kernel<__bf16>), dim3(g.nchunks, g.nslices, batch), dim3(GN_THREADS), 0, st, (const __bf16*)x,
                           (const __bf16*)nullptr, g, partial);
    else
        hipLaunchKernelGGL((gn_stats_kernel<_Float16>), dim3(g.nchunks, g.nslices, batch), dim3(GN_THREADS), 0, st, (const _Float16*)x,
                           (const _Float16*)nullptr, g, partial);
    IDB_CHECK_LAUNCH("idb_gemm(gn stats)");
    return IDB_OK;
}

extern "C" int idb_groupnorm_stats(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw, int32_t groups,
                                   float* partials, size_t partials_bytes, int32_t* chunks, int32_t dtype, void* stream) {
    IDB_REQUIRE(x0 && partials && chunks && idb_aligned16(x0) && idb_aligned16(partials) && (!x1 || idb_aligned16(x1)), "idb_groupnorm_stats: null or unaligned pointer");
    IDB_REQUIRE(idb_is_operand_dtype(dtype) && batch > 0 && hw > 0 && groups > 0, "idb_groupnorm_stats: bad arguments");
    IDB_REQUIRE(c0 > 0 && c0 % 8 == 0 && c1 % 8 == 0 && (x1 != nullptr) == (c1 > 0) && (c0 + c1) % groups == 0 && (c0 + c1) / groups >= 2,
                "idb_groupnorm_stats: c0=%d c1=%d groups=%d unsupported", c0, c1, groups);
    const GnGeom g = gn_geometry(c0, c1, batch, hw, groups);
    IDB_REQUIRE(g.cols <= GN_THREADS && g.gps <= 64, "idb_groupnorm_stats: unsupported geometry");
    IDB_REQUIRE(partials_bytes >= (size_t)batch * g.nchunks * groups * 2 * sizeof(float), "idb_groupnorm_stats: partials buffer too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == IDB_BF16)
        hipLaunchKernelGGL((gn_stats_kernel<__bf16>), dim3(g.nchunks, g.nslices, batch), dim3(GN_THREADS), 0, st, (const __bf16*)x0, (const __bf16*)x1, g, partials);
    else
        hipLaunchKernelGGL((gn_stats_kernel<_Float16>), dim3(g.nchunks, g.nslices, batch), dim3(GN_THREADS), 0, st, (const _Float16*)x0, (const _Float16*)x1, g, partials);
    IDB_CHECK_LAUNCH("idb_groupnorm_stats");
    *chunks = g.nchunks;
    return IDB_OK;
}

extern "C" size_t idb_groupnorm_workspace_bytes(int32_t batch, int32_t hw, int32_t groups) {
    if (batch <= 0 || hw <= 0 || groups <= 0) return 0;
    return (size_t)batch * GN_MAXCHUNKS * groups * 2 * sizeof(float);
}

extern "C" int idb_groupnorm(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw,
                             int32_t groups, float eps, const float* gamma, const float* beta, int32_t silu, void* out,
                             int32_t dtype, void* workspace, size_t workspace_bytes, int32_t* sync, int32_t sync_len,
                             const float* partials_in, int32_t partials_chunks, void* stream) {
    IDB_REQUIRE(idb_is_operand_dtype(dtype), "idb_groupnorm: dtype must be bf16/f16");
    IDB_REQUIRE(x0 && out && gamma && beta && idb_aligned16(x0) && idb_aligned16(out) && idb_aligned16(gamma) && idb_aligned16(beta),
                "idb_groupnorm: null/unaligned pointer");
    IDB_REQUIRE(batch > 0 && hw > 0 && groups > 0 && c0 > 0 && c1 >= 0, "idb_groupnorm: bad dims");
    IDB_REQUIRE((c1 == 0) == (x1 == nullptr), "idb_groupnorm: x1/c1 mismatch");
    IDB_REQUIRE(c1 == 0 || idb_aligned16(x1), "idb_groupnorm: x1 unaligned");
    const int C = c0 + c1;
    IDB_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && C % groups == 0, "idb_groupnorm: channels %d+%d / groups %d unsupported", c0, c1, groups);
    IDB_REQUIRE(C / groups >= 2, "idb_groupnorm: needs at least 2 channels per group");
    {
        const GnGeom g = gn_geometry(c0, c1, batch, hw, groups);
        IDB_REQUIRE(g.cols <= GN_THREADS && g.gps <= 64 && batch <= 65535 && g.nslices <= 65535,
                    "idb_groupnorm: unsupported geometry C=%d groups=%d (slice %d channels, %d groups/slice)", C, groups, g.SW, g.gps);
    }
    const size_t need = idb_groupnorm_workspace_bytes(batch, hw, groups);
    IDB_REQUIRE(workspace && workspace_bytes >= need, "idb_groupnorm: workspace too small (%zu < %zu)", workspace_bytes, need);
    IDB_REQUIRE(!sync || (sync_len > 0 && ((uintptr_t)sync & 3) == 0), "idb_groupnorm: bad sync counter array");
    IDB_REQUIRE(!partials_in || (c1 == 0 && hw % 64 == 0 && partials_chunks == hw / 64 && partials_chunks <= GN_MAXCHUNKS &&
                                 ((uintptr_t)partials_in & 7) == 0),
                "idb_groupnorm: partials_in needs one dense input, hw %% 64 == 0, partials_chunks == hw / 64 <= %d", GN_MAXCHUNKS);
    hipStream_t st = (hipStream_t)stream;
    return dtype == IDB_BF16
               ? run_groupnorm<__bf16>(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, silu, out, workspace, sync, sync_len, partials_in,
                                       partials_chunks, st)
               : run_groupnorm<_Float16>(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, silu, out, workspace, sync, sync_len, partials_in,
                                         partials_chunks, st);
}

extern "C" int idb_groupnorm_fp8(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t batch, int32_t hw, int32_t groups, float eps,
                                 const float* gamma, const float* beta, int32_t silu, void* out8, float out_inv_scale, int32_t dtype,
                                 void* workspace, size_t workspace_bytes, const float* partials_in, int32_t partials_chunks, void* stream) {
    IDB_REQUIRE(out_inv_scale > 0.f && out8 && (((uintptr_t)out8) & 7) == 0, "idb_groupnorm_fp8: out_inv_scale must be > 0, out 8-byte aligned");
    IDB_REQUIRE(idb_is_operand_dtype(dtype) && x0 && gamma && beta && idb_aligned16(x0) && idb_aligned16(gamma) && idb_aligned16(beta), "idb_groupnorm_fp8: bad pointers / dtype");
    IDB_REQUIRE(batch > 0 && hw > 0 && groups > 0 && c0 > 0 && c1 >= 0 && (c1 == 0) == (x1 == nullptr) && (c1 == 0 || idb_aligned16(x1)), "idb_groupnorm_fp8: bad dims");
    const int C = c0 + c1;
    IDB_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && C % groups == 0 && C / groups >= 2, "idb_groupnorm_fp8: channels %d+%d / groups %d unsupported", c0, c1, groups);
    {
        const GnGeom g = gn_geometry(c0, c1, batch, hw, groups);
        IDB_REQUIRE(g.cols <= GN_THREADS && g.gps <= 64 && batch <= 65535 && g.nslices <= 65535, "idb_groupnorm_fp8: unsupported geometry");
    }
    IDB_REQUIRE(workspace && workspace_bytes >= idb_groupnorm_workspace_bytes(batch, hw, groups), "idb_groupnorm_fp8: workspace too small");
    IDB_REQUIRE(!partials_in || (c1 == 0 && hw % 64 == 0 && partials_chunks == hw / 64 && partials_chunks <= GN_MAXCHUNKS), "idb_groupnorm_fp8: bad partials_in");
    hipStream_t st = (hipStream_t)stream;
    return dtype == IDB_BF16 ? run_groupnorm<__bf16>(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, silu, out8, workspace, nullptr, 0, partials_in,
                                                     partials_chunks, st, out_inv_scale)
                             : run_groupnorm<_Float16>(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, silu, out8, workspace, nullptr, 0, partials_in,
                                                       partials_chunks, st, out_inv_scale);
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
    const int nch = (c / 8 + 63) / 64;
#define IDB_LN_LAUNCH(TT, N) \
    hipLaunchKernelGGL((layernorm_kernel<TT, N>), dim3((unsigned)blocks), dim3(256), 0, st, (const TT*)x, (TT*)out, (long long)rows, c, eps, gamma, beta)
    if (dtype == IDB_BF16) {
        if (nch == 1) IDB_LN_LAUNCH(__bf16, 1); else if (nch == 2) IDB_LN_LAUNCH(__bf16, 2); else IDB_LN_LAUNCH(__bf16, 3);
    } else {
        if (nch == 1) IDB_LN_LAUNCH(_Float16, 1); else if (nch == 2) IDB_LN_LAUNCH(_Float16, 2); else IDB_LN_LAUNCH(_Float16, 3);
    }
#undef IDB_LN_LAUNCH
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

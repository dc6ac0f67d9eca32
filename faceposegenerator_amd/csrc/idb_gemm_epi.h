// Shared by the implicit-GEMM translation units (idb_gemm.hip, idb_hconv.hip): kernel parameter block, the epilogues
// (LDS-staged coalesced / direct / split-K slabs) and the split-K reduce launches.  Moved out of idb_gemm.hip unchanged.
#pragma once
#include "idb_common.h"
#include <stdlib.h>

struct GemmSrcK {
    const char* ptr;
    unsigned bytes;   // tensor size in bytes = buffer num_records (< 2^31, checked on the host)
    int C, taps, H, W, up;
};

struct GemmParams {
    GemmSrcK src[IDB_MAX_SRC];
    int M, N, HW, OW, stride, pad;
    unsigned w_row_bytes, w_bytes;   // w_row_bytes: bytes between consecutive weight rows inside a 16-row block
    unsigned w_blk_bytes, w_kstep;   // bytes between 16-row blocks / between K-steps of one row (idb_gemm_desc.w_layout; idb_gemm kernels only)
    int ktiles, kt_per_split, splitk;
    int slab_swc;   // 0: split-K slabs are [z][M][N] row-major; else [z][M/64][N/swc][64][swc] blocks = the windows of idb_splitk_reduce_gn_kernel
    int xcd_mode;   // 0: tiles dealt to XCDs in runs, K-split on grid z; 1/2: one K-slice per XCD (group), see idb_gemm_kernel
    const char* w;
    const float* bias;
    const float* sbias;
    int sbias_ld;
    const char* res;
    void* out;
    int out_ld, out_f32, geglu;
    float scale;
    float* partial;
    int tiles_n;
    int dbg_skip_store, lds_epi, act, dbg_loop;   // dbg_* are read only by the profiling build (IDB_PROFILING, `make prof`)
    unsigned* counters;   // non-null: in-kernel split-K reduce
    unsigned out_bytes;   // persistent variant: size of the output tensor (buffer range check drops masked stores)
    // LayerNorm folded into this GEMM (idb_gemm_desc.ln_*): A holds the RAW rows x, W its gamma-scaled weights; the LDS-staged
    // epilogue turns acc = x W'^T into rstd (acc - mean u) + v from the per-row partial sums the producer of x emitted
    float* rowstat_out;   // producer side: [M][tiles_n][2] partial {sum, sum of squares} of the rounded output rows
    const float* ln_stats;
    const float* ln_u;
    const float* ln_v;
    int ln_nt, ln_c;
    float ln_eps;
    // first GroupNorm pass of the output from the LDS-staged epilogue (no split-K: no reduce launch to ride on): non-null only when
    // the tile's column range holds whole groups (host: idb_epilogue_emits_gn)
    float* gn_part;
    int gn_groups;
    // grouped weights (idb_gemm_desc.w_groups): tile rows [m0, m0 + BM) use matrix (m0 / w_group_rows) % w_groups
    int w_groups, w_group_rows;
    long long w_group_stride;
    // GroupNorm(+SiLU) of the first gn_nsrc sources applied by normalizer waves inside the kernel (idb_gemm_kernel_gn): statistics as
    // idb_groupnorm's partials_in over the channel concatenation of those sources, gamma / beta over the same channels
    const float* gn_in_part;
    const float* gn_in_gamma;
    const float* gn_in_beta;
    int gn_in_chunks, gn_in_groups, gn_in_nsrc, gn_in_silu, gn_in_c;      // gn_in_c: channels of the normalised concatenation
    float gn_in_eps;
};

// byte offset of the weight matrix (and element offset of its folded-LayerNorm vectors) a tile uses
__device__ __forceinline__ int idb_weight_group(const GemmParams& p, int m0) { return p.w_groups > 1 ? (m0 / p.w_group_rows) % p.w_groups : 0; }

// voffset of a lane that must read zeros: beyond num_records of every descriptor (all < 2^31), and
// voffset + soffset cannot wrap, whichever of the two the hardware range check looks at.
[[maybe_unused]] constexpr unsigned IDB_OOB = 0x80000000u;
[[maybe_unused]] constexpr int IDB_RSRC_FLAGS = 0x00020000;

// Measurement-only branches (skip the epilogue stores, alias them, run the K loop without MFMAs or without loads) exist only in
// the separately compiled profiling library libidb_kernels_prof.so (`make prof`, -DIDB_PROFILING); the shipped kernels carry none.
#ifdef IDB_PROFILING
#define IDB_DBG(x) (x)
#else
#define IDB_DBG(x) 0
#endif

// LDS-staged epilogue (operand-dtype outputs): the accumulator tile goes through LDS so that global traffic is
// whole 16-byte-per-lane row segments (full 128-B lines) instead of 8-byte pieces at a row stride — measured on the
// K = C projection GEMMs, the scattered stores alone cost as much as the whole K loop.
//   phase R (residual only): coalesced copy of the residual tile into LDS
//   phase W: each lane adds bias / per-sample bias / residual (fp32, ONE rounding) and writes its 4-channel pieces in place
//   phase S: coalesced LDS -> global stores
// Folded LayerNorm (GemmParams::ln_stats), part 1 — called from the kernel PROLOGUE behind the first LDS-DMA issues: THREADS / BM
// threads per tile row add their share of the row's ln_nt partial {sum, sum of squares}; the loads travel with the first operand
// tiles instead of standing exposed in every tile's epilogue (measured +33 % on the GEGLU projections there).  The two partial
// sums ride through the K loop in registers: a table in LDS beside the ring was measured to cost the 128x160 2-stage tile its
// second workgroup per CU (+37-51 % on the QKV projections).
template <int BM, int THREADS>
__device__ __forceinline__ float2 idb_ln_row_partials(const GemmParams& p, int m0, int tid) {
    constexpr int TPR = THREADS / BM;
    const int row = tid / TPR, sub = tid % TPR;
    const int m = min(m0 + row, p.M - 1);
    const float* ps = p.ln_stats + (long long)m * p.ln_nt * 2;
    float a = 0.f, q = 0.f;
    for (int t0 = 0; t0 < p.ln_nt; t0 += 4 * TPR) {
        float2 pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = *(const float2*)(ps + 2 * min(t0 + sub + k * TPR, p.ln_nt - 1));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = t0 + sub + k * TPR < p.ln_nt;
            a += in ? pv[k].x : 0.f;
            q += in ? pv[k].y : 0.f;
        }
    }
    return make_float2(a, q);
}

// Part 2 — at the head of the epilogue: shuffle tree over the row's threads, {mean, rstd} into `rowtab` (LDS, BM entries, in the
// ring the K loop is done with); the caller's barrier publishes it.
template <int BM, int THREADS>
__device__ __forceinline__ void idb_ln_row_table(const GemmParams& p, float2* rowtab, int tid, float2 part) {
    constexpr int TPR = THREADS / BM;
    float a = part.x, q = part.y;
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (tid % TPR == 0) {
        const float inv_c = 1.0f / (float)p.ln_c;
        const float mean = a * inv_c;
        // E[x^2] - mean^2 in double: the two terms can agree to 4-5 digits when |mean| >> std
        double var = (double)q * (double)inv_c - (double)mean * (double)mean;
        if (var < 0.0) var = 0.0;
        rowtab[tid / TPR] = make_float2(mean, __builtin_amdgcn_rsqf((float)var + p.ln_eps));
    }
}

// Column vectors of a lane's NF accumulator fragments, loaded ONCE and unconditionally (an absent term reads a valid dummy address
// and is dropped by a value select at the use): the loads are independent and all in flight together.  Loads behind
// `if (p.bias)` / `if (sb)` inside the (i, j) loop were issued and awaited one by one — 2-4 exposed L2 round trips per fragment,
// measured +20-70 % on the K = C projections once the folded-LayerNorm vectors joined them.
// Two vectors per fragment (more in flight costs the 8-wave tiles their second workgroup per CU in VGPRs):
//   ca = bias, or ln_v of a folded LayerNorm (the caller adds the layer's bias into ln_v; idb_gemm rejects bias + ln_stats)
//   cb = the per-sample bias when the whole tile lies in one sample (else per-row loads in the epilogue), or ln_u
// The values stay RAW (no arithmetic here), so the wait lands at the first use.
template <int NF, int BM>
__device__ __forceinline__ void idb_load_colvecs(const GemmParams& p, int m0, int n0, int wn, int fg, f32x4 (&ca)[NF], f32x4 (&cb)[NF]) {
    const bool ln = p.ln_stats != nullptr;
    const int m_last = min(m0 + BM, p.M) - 1;
    const bool sb_tile = p.sbias && (m0 / p.HW == m_last / p.HW);
    const float* dummy = (const float*)p.w;
    const long long go = (long long)idb_weight_group(p, m0) * p.N;           // grouped weights: one (u, v) pair per group
    const float* pa = ln ? p.ln_v + go : p.bias;
    const float* pb = ln ? p.ln_u + go : (sb_tile ? p.sbias + (long long)(m0 / p.HW) * p.sbias_ld : nullptr);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int nc = min(n0 + (wn * NF + j) * 16 + fg * 4, p.N - 4);
        ca[j] = *(const f32x4*)(pa ? pa + nc : dummy);
        cb[j] = *(const f32x4*)(pb ? pb + nc : dummy);
    }
}

template <typename T, int MF, int NF, bool GEGLU, int WM = 2>
__device__ __forceinline__ void idb_lds_epilogue(const GemmParams& p, char* smem, f32x4 (&acc)[MF][NF], const f32x4 (&ca)[NF],
                                                 const f32x4 (&cb)[NF], int m0, int n0, int tid, int wm, int wn, int fr, int fg, bool ln = false,
                                                 float2 ln_part = {0.f, 0.f}) {
    using V8 = typename Op<T>::v8;
    using V4 = typename Op<T>::v4;
    constexpr int BM = 16 * MF * WM, BN = 32 * NF, THREADS = 128 * WM;
    constexpr int BNO = GEGLU ? BN / 2 : BN;          // output columns of this tile
    constexpr int OLD = BNO * 2 + 16;                  // LDS row stride (bytes), 16-B aligned, de-phased banks
    constexpr int CPR = BNO / 8;                       // 16-byte chunks per row
    constexpr int NCHUNK = BM * CPR, ITER = (NCHUNK + THREADS - 1) / THREADS;
    const int No = GEGLU ? p.N / 2 : p.N;
    const int n0o = GEGLU ? n0 / 2 : n0;
    __syncthreads();                                   // every wave is done reading the last K tile
    float2* rowtab = nullptr;
    if (ln) {                                          // folded LayerNorm: {mean, rstd} per tile row, behind the staging area
        rowtab = (float2*)(smem + ((BM * OLD + 15) & ~15));
        idb_ln_row_table<BM, THREADS>(p, rowtab, tid, ln_part);
        if (!p.res) __syncthreads();
    }
    if (p.res) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q = it * THREADS + tid;
            const int row = q / CPR, c = q - row * CPR;
            const int m = m0 + row, n = n0o + c * 8;
            if (q < NCHUNK && m < p.M && n < No)
                *(V8*)(smem + row * OLD + c * 16) = *(const V8*)((const T*)p.res + (long long)m * p.out_ld + n);
        }
        __syncthreads();
    }
    // ca / cb: idb_load_colvecs (raw values; which terms exist is decided here)
    const int m_last = min(m0 + BM, p.M) - 1;
    const bool sb_tile = p.sbias && (m0 / p.HW == m_last / p.HW);      // else (HW < BM): per-row loads below
    const bool has_a = rowtab || p.bias, has_b = rowtab || sb_tile;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int row = (wm * MF + i) * 16 + fr;
        const int mc = min(m0 + row, p.M - 1);
        const float* sb = (p.sbias && !sb_tile) ? p.sbias + (long long)(mc / p.HW) * p.sbias_ld : nullptr;
        const float2 mr = rowtab ? rowtab[row] : make_float2(0.f, 1.f);
        // folded LayerNorm: rstd (scale acc - mean u) + v = rs acc + kb u + v;  otherwise scale acc + bias + per-sample bias
        const float rs = mr.y * p.scale, kb = rowtab ? -mr.x * mr.y : 1.f;
        if constexpr (GEGLU) {
#pragma unroll
            for (int j = 0; j < NF; j += 2) {
                const int nv = n0 + (wn * NF + j) * 16 + fg * 4;
                const int col = (wn * NF + j) * 8 + fg * 4;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = rs * acc[i][j][e], gt = rs * acc[i][j + 1][e];
                    if (nv + 16 < p.N) {
                        v += (has_b ? kb * cb[j][e] : 0.f) + (has_a ? ca[j][e] : 0.f);
                        gt += (has_b ? kb * cb[j + 1][e] : 0.f) + (has_a ? ca[j + 1][e] : 0.f);
                    }
                    o[e] = v * gelu_erf_f(gt);
                }
                V4 pk = {from_f32<T>(o[0]), from_f32<T>(o[1]), from_f32<T>(o[2]), from_f32<T>(o[3])};
                *(V4*)(smem + row * OLD + col * 2) = pk;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int col = (wn * NF + j) * 16 + fg * 4;
                const int n = n0 + col;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * acc[i][j][e];
                if (n < p.N) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += (has_b ? kb * cb[j][e] : 0.f) + (has_a ? ca[j][e] : 0.f);
                    if (sb) {
                        const f32x4 b4 = *(const f32x4*)(sb + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += b4[e];
                    }
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gelu_erf_f(o[e]);
                }
                if (p.res) {
                    const V4 r4 = *(const V4*)(smem + row * OLD + col * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += to_f32<T>(r4[e]);
                }
                V4 pk = {from_f32<T>(o[0]), from_f32<T>(o[1]), from_f32<T>(o[2]), from_f32<T>(o[3])};
                *(V4*)(smem + row * OLD + col * 2) = pk;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int q = it * THREADS + tid;
        const int row = q / CPR, c = q - row * CPR;
        const int m = m0 + row, n = n0o + c * 8;
        const int mo = IDB_DBG(p.dbg_skip_store) == 2 ? (m & 127) : m;      // profiling: every tile writes the same L2-resident rows
        if (q < NCHUNK && m < p.M && n < No) *(V8*)((T*)p.out + (long long)mo * p.out_ld + n) = *(const V8*)(smem + row * OLD + c * 16);
    }
    if (p.rowstat_out) {                                   // behind the stores: its LDS reads overlap their flight
        // per-row partial {sum, sum of squares} of the ROUNDED tile row (what a LayerNorm of the output would read), one entry per
        // column tile: the consumer GEMM (idb_gemm_desc.ln_stats) adds the tiles_n entries of a row.  TPR threads per row, fixed
        // order + shuffle tree: deterministic
        constexpr int TPR = THREADS / BM;
        const int row = tid / TPR, sub = tid % TPR;
        const char* rp = smem + row * OLD;
        float a = 0.f, q = 0.f;
        for (int c = sub; c < CPR; c += TPR) {
            if (n0o + c * 8 < No) {
                const V8 v = *(const V8*)(rp + c * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = to_f32<T>(v[e]);
                    a += f;
                    q += f * f;
                }
            }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {
            a += __shfl_xor(a, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (sub == 0 && m0 + row < p.M) *(float2*)(p.rowstat_out + ((long long)(m0 + row) * p.tiles_n + n0 / BN) * 2) = make_float2(a, q);
    }
    if constexpr (!GEGLU) {
        if (p.gn_part) {
            // first GroupNorm pass of the ROUNDED output (idb_gemm_desc.gn_partials: [batch][HW / 64][groups] {sum, sum of squares}),
            // behind the stores like the row statistics: the tile holds BM / 64 row chunks x BN / cpg whole groups = 4-32 items of
            // 64 x cpg values; TPI lanes of ONE wave per item, each summing rows sub, sub + TPI, ... as dword pairs (cpg is even),
            // fixed order + shuffle tree: deterministic, no atomics.  The host guarantees M % 64 == 0 and N % BN == 0.
            const int cpg = p.N / p.gn_groups, gpt = BN / cpg, items = (BM / 64) * gpt;
            int tpi = THREADS / items;
            if (tpi > 64) tpi = 64;
            const int item = tid / tpi, sub = tid - item * tpi;
            if (item < items) {
                const int c = item / gpt, g = item - c * gpt;
                const char* base = smem + (c * 64) * OLD + g * cpg * 2;
                const bool live = m0 + c * 64 < p.M;
                float a = 0.f, q = 0.f;
                for (int r = sub; r < 64 && live; r += tpi) {
                    const unsigned* rp = (const unsigned*)(base + r * OLD);
                    for (int k = 0; k < cpg / 2; ++k) {
                        const unsigned w2 = rp[k];
                        const float f0 = to_f32<T>(__builtin_bit_cast(T, (unsigned short)(w2 & 0xffffu)));
                        const float f1 = to_f32<T>(__builtin_bit_cast(T, (unsigned short)(w2 >> 16)));
                        a += f0 + f1;
                        q += f0 * f0 + f1 * f1;
                    }
                }
                for (int o = 1; o < tpi; o <<= 1) {
                    a += __shfl_xor(a, o, 64);
                    q += __shfl_xor(q, o, 64);
                }
                if (sub == 0 && live) {
                    const int mrow = m0 + c * 64, b = mrow / p.HW, chunk = (mrow - b * p.HW) >> 6;
                    float* dst = p.gn_part + (((long long)b * (p.HW >> 6) + chunk) * p.gn_groups + n0 / cpg + g) * 2;
                    dst[0] = a;
                    dst[1] = q;
                }
            }
        }
    }
}

// Everything after the K loop: LDS-staged coalesced epilogue for operand-dtype outputs, direct epilogue for fp32 outputs /
// split-K slabs / odd widths.  PRE: ca / cb were loaded by the kernel before its K loop (idb_load_colvecs).
template <typename T, int MF, int NF, int WM = 2, bool PRE = false>
__device__ __forceinline__ void idb_gemm_epilogue(const GemmParams& p, char* smem, f32x4 (&acc)[MF][NF], f32x4 (&ca)[NF], f32x4 (&cb)[NF], int m0,
                                                  int n0, int tid, int wm, int wn, int fr, int fg, int kz, bool ln = false,
                                                  float2 ln_part = {0.f, 0.f}) {
    if (IDB_DBG(p.dbg_skip_store) == 1) {            // profiling experiment: keep the accumulators live, write nothing
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) ((float*)p.out)[0] = keep;
        return;
    }
    if (p.splitk > 1) {
        // ---- split-K: every workgroup stores its fp32 slab (16-B per lane)
        const bool vec4 = (p.N & 3) == 0;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + (wm * MF + i) * 16 + fr;
            if (m >= p.M) continue;
            float* dst = p.partial + ((long long)kz * p.M + m) * p.N;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int n = n0 + (wn * NF + j) * 16 + fg * 4;
                if (p.slab_swc) {
                    // blocked slab: the reduce + GroupNorm-statistics kernel then reads each of its 64 x swc windows as ONE
                    // contiguous run (row-major windows of 40-64 channels are 160-256-byte pieces at a row stride)
                    if (n < p.N) {
                        const int sl = n / p.slab_swc;
                        float* blk = p.partial + ((((long long)kz * (p.M >> 6) + (m >> 6)) * (p.N / p.slab_swc) + sl) * 64 + (m & 63)) * p.slab_swc;
                        *(f32x4*)(blk + (n - sl * p.slab_swc)) = acc[i][j];
                    }
                } else if (vec4 && n + 3 < p.N) {
                    *(f32x4*)(dst + n) = acc[i][j];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) dst[n + e] = acc[i][j][e];
                }
            }
        }
        if (!p.counters) return;                      // two-launch mode: idb_splitk_reduce_kernel finishes the job
        // ---- publish the slab (agent-scope release), draw a ticket; the last arriver of this tile reduces.
        // Placement-independent protocol (cdna_hip_programming.md, Guideline 16 / "In-launch split-K reduction").
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = (unsigned*)smem;             // the K loop is done with LDS (barrier above)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned* cnt = p.counters + (m0 / (16 * MF * WM)) * p.tiles_n + n0 / (32 * NF);
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = t == (unsigned)p.splitk - 1u;
            if (last) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the counter zero for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last ? 1u : 0u;
        }
        __syncthreads();
        const bool is_last = *flag != 0u;
        __syncthreads();                              // flag consumed before the epilogue reuses LDS
        if (!is_last) return;
        // ---- deterministic reduction: every slab, own one included, in ascending split order (bit-identical to the
        // two-launch path whoever arrives last)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + (wm * MF + i) * 16 + fr;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int n = n0 + (wn * NF + j) * 16 + fg * 4;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                if (m < p.M) {
                    for (int z = 0; z < p.splitk; ++z) {
                        const float* src = p.partial + ((long long)z * p.M + m) * p.N + n;
                        if (vec4 && n + 3 < p.N) {
                            const f32x4 v = *(const f32x4*)src;
#pragma unroll
                            for (int e = 0; e < 4; ++e) sum[e] += v[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.N) sum[e] += src[e];
                        }
                    }
                }
                acc[i][j] = sum;
            }
        }
    }
    if (p.lds_epi) {
        if constexpr (!PRE) idb_load_colvecs<NF, 16 * MF * WM>(p, m0, n0, wn, fg, ca, cb);
        if (p.geglu) {
            if constexpr ((NF & 1) == 0) idb_lds_epilogue<T, MF, NF, true, WM>(p, smem, acc, ca, cb, m0, n0, tid, wm, wn, fr, fg, ln, ln_part);
        } else {
            idb_lds_epilogue<T, MF, NF, false, WM>(p, smem, acc, ca, cb, m0, n0, tid, wm, wn, fr, fg, ln, ln_part);
        }
        return;
    }
    // ---- direct epilogue (fp32 outputs, split-K slabs, odd widths): lane holds out[m][n .. n+3], m = tile row (lane&15), n = 4*(lane>>4) + reg
    const bool vec_ok = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + (wm * MF + i) * 16 + fr;
        if (m >= p.M) continue;
        const float* sb = p.sbias ? p.sbias + (long long)(m / p.HW) * p.sbias_ld : nullptr;
        if (p.geglu) {
            if constexpr ((NF & 1) == 0) {
#pragma unroll
                for (int j = 0; j < NF; j += 2) {
                    const int nv = n0 + (wn * NF + j) * 16 + fg * 4;   // packed row of the value part
                    if (nv >= p.N) continue;   // N % 32 == 0: a value/gate pair is in range or not as a whole
                    const int oc = (n0 + (wn * NF + j) * 16) / 2 + fg * 4;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][e] * p.scale, gt = acc[i][j + 1][e] * p.scale;
                        if (p.bias) {
                            v += p.bias[nv + e];
                            gt += p.bias[nv + 16 + e];
                        }
                        o[e] = v * gelu_erf_f(gt);
                    }
                    T* dst = (T*)p.out + (long long)m * p.out_ld + oc;
                    typename Op<T>::v4 pk = {from_f32<T>(o[0]), from_f32<T>(o[1]), from_f32<T>(o[2]), from_f32<T>(o[3])};
                    *(typename Op<T>::v4*)dst = pk;
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = n0 + (wn * NF + j) * 16 + fg * 4;
            if (n >= p.N) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e] * p.scale;
            if (vec_ok && n + 3 < p.N) {
                if (p.bias) {
                    const f32x4 b4 = *(const f32x4*)(p.bias + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += b4[e];
                }
                if (sb) {
                    const f32x4 b4 = *(const f32x4*)(sb + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += b4[e];
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gelu_erf_f(o[e]);
                }
                if (p.res) {
                    const typename Op<T>::v4 r4 = *(const typename Op<T>::v4*)((const T*)p.res + (long long)m * p.out_ld + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += to_f32<T>(r4[e]);
                }
                if (p.out_f32) {
                    *(f32x4*)((float*)p.out + (long long)m * p.out_ld + n) = (f32x4){o[0], o[1], o[2], o[3]};
                } else {
                    typename Op<T>::v4 pk = {from_f32<T>(o[0]), from_f32<T>(o[1]), from_f32<T>(o[2]), from_f32<T>(o[3])};
                    *(typename Op<T>::v4*)((T*)p.out + (long long)m * p.out_ld + n) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float v = o[e];
                    if (p.bias) v += p.bias[n + e];
                    if (sb) v += sb[n + e];
                    if (p.act == 1) v = gelu_erf_f(v);
                    if (p.res) v += to_f32<T>(((const T*)p.res)[(long long)m * p.out_ld + n + e]);
                    if (p.out_f32) ((float*)p.out)[(long long)m * p.out_ld + n + e] = v;
                    else ((T*)p.out)[(long long)m * p.out_ld + n + e] = from_f32<T>(v);
                }
            }
        }
    }
}

// the form every kernel uses: column vectors loaded at the head of the epilogue.  (PRE = loading them before the K loop of the
// 64-row tiles, older than every LDS-DMA, was measured: batch 1 6.653 -> 6.630 images/s, batch 8 13.85 -> 13.68 — the 16-40 more
// live VGPRs and the later first DMA cost more than the one L2 round trip saved.)
template <typename T, int MF, int NF, int WM = 2>
__device__ __forceinline__ void idb_gemm_epilogue(const GemmParams& p, char* smem, f32x4 (&acc)[MF][NF], int m0, int n0, int tid, int wm, int wn,
                                                  int fr, int fg, int kz, bool ln = false, float2 ln_part = {0.f, 0.f}) {
    f32x4 ca[NF], cb[NF];
    idb_gemm_epilogue<T, MF, NF, WM, false>(p, smem, acc, ca, cb, m0, n0, tid, wm, wn, fr, fg, kz, ln, ln_part);
}


// Split-K tail: sum the fp32 slabs and apply the same epilogue (bias, per-sample bias, residual).
// VEC = 4: one thread owns out[m][n .. n+3]; every load of the thread — residual, biases, then the slabs in batches of 8 —
// is issued unconditionally (clamped slab index, masked in registers) so that the memory round trips overlap: a load inside
// a data-dependent loop or branch gets an s_waitcnt vmcnt(0) right behind it, and with 8-30 slabs the first version of this
// kernel spent its time in that many dependent round trips.  Slabs are added in ascending split order (deterministic, and
// bit-identical to the in-kernel reduce).  VEC = 1 is the scalar fallback for odd widths / unaligned residuals.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void idb_splitk_reduce_kernel(const float* __restrict__ partial, int splitk,
                                                                int M, int N, int HW, float scale,
                                                                const float* bias, const float* sbias, int sbias_ld,
                                                                const T* res, void* out, int out_ld, int out_f32) {
    const long long total = (long long)M * N / VEC;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long e0 = idx * VEC;
    const int m = (int)(e0 / N), n = (int)(e0 - (long long)m * N);
    if constexpr (VEC == 4) {
        using V4 = typename Op<T>::v4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        V4 r4;
        f32x4 bi = zero, sbv = zero;
        if (res) r4 = *(const V4*)(res + (long long)m * out_ld + n);          // uniform conditions: no divergence, and the
        if (bias) bi = *(const f32x4*)(bias + n);                                // waits for these sit behind the slab loads
        if (sbias) sbv = *(const f32x4*)(sbias + (long long)(m / HW) * sbias_ld + n);
        const float* src = partial + (long long)m * N + n;
        const long long slab = (long long)M * N;
        f32x4 v = zero;
        for (int z0 = 0; z0 < splitk; z0 += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(src + (long long)min(z0 + u, splitk - 1) * slab);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = z0 + u < splitk;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += in ? t[u][e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] *= scale;
            if (bias) v[e] += bi[e];
            if (sbias) v[e] += sbv[e];
            if (res) v[e] += to_f32<T>(r4[e]);
        }
        if (out_f32) {
            *(f32x4*)((float*)out + (long long)m * out_ld + n) = v;
        } else {
            V4 o = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
            *(V4*)((T*)out + (long long)m * out_ld + n) = o;
        }
    } else {
        float v = 0.f;
        for (int z = 0; z < splitk; ++z) v += partial[((long long)z * M + m) * N + n];
        v *= scale;
        if (bias) v += bias[n];
        if (sbias) v += sbias[(long long)(m / HW) * sbias_ld + n];
        if (res) v += to_f32<T>(res[(long long)m * out_ld + n]);
        if (out_f32) ((float*)out)[(long long)m * out_ld + n] = v;
        else ((T*)out)[(long long)m * out_ld + n] = from_f32<T>(v);
    }
}

// Split-K tail that also emits the first GroupNorm pass of its output (idb_gemm_desc.gn_partials): one workgroup owns 64
// rows x SWC channels (SWC a multiple of the group width and of 4, SWC <= 64) with ONE thread per (row, 4 channels), so that —
// as in idb_splitk_reduce_kernel<T,4> — every load of a thread is in flight at once.  It sums the slabs, applies the
// epilogue, stores the rounded result, and reduces {x, x^2} of the ROUNDED values per group through LDS (fixed order:
// deterministic).  Saves the statistics launch of the GroupNorm that follows a split-K convolution: 9.2 us against
// 5.5 + 5.4 us (batch 1: +0.7 %).  Measured and dropped: a serial 6-rows-per-thread form (slower than the two launches) and a
// 2-rows-per-thread form with 80-128-channel slices (longer contiguous runs but half the workgroups: no gain).
template <typename T>
__global__ __launch_bounds__(1024) void idb_splitk_reduce_gn_kernel(const float* __restrict__ partial, int splitk, int M, int N,
                                                                    int HW, float scale, const float* bias, const float* sbias,
                                                                    int sbias_ld, const T* res, T* out, int out_ld, float* gn_part,
                                                                    int groups, int swc) {
    using V4 = typename Op<T>::v4;
    __shared__ float part[1024][2][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = swc >> 2;                                  // blockDim.x == 64 * cols
    const int col = tid % cols, row = tid / cols;
    const int cpg = N / groups;
    const int n = blockIdx.y * swc + col * 4;
    const int m_blk = blockIdx.x * 64, m = m_blk + row;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    V4 r4;
    f32x4 bi = zero, sbv = zero;
    if (res) r4 = *(const V4*)(res + (long long)m * out_ld + n);
    if (bias) bi = *(const f32x4*)(bias + n);
    if (sbias) sbv = *(const f32x4*)(sbias + (long long)(m / HW) * sbias_ld + n);
    // slabs in the blocked layout GemmParams::slab_swc describes: this workgroup's window is one contiguous run per slab
    const float* src = partial + (((long long)blockIdx.x * gridDim.y + blockIdx.y) * 64 + row) * swc + col * 4;
    const long long slab = (long long)M * N;
    f32x4 v = zero;
    for (int z0 = 0; z0 < splitk; z0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(src + (long long)min(z0 + u, splitk - 1) * slab);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = z0 + u < splitk;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += in ? t[u][e] : 0.f;
        }
    }
    const int g_first = n / cpg;
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    V4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] * scale;
        if (bias) x += bi[e];
        if (sbias) x += sbv[e];
        if (res) x += to_f32<T>(r4[e]);
        o[e] = from_f32<T>(x);
        const float xr = to_f32<T>(o[e]);
        const int k = ((n + e) / cpg - g_first) & 1;            // four channels touch at most two groups (cpg >= 2)
        gs[k] += xr;
        gq[k] += xr * xr;
    }
    *(V4*)(out + (long long)m * out_ld + n) = o;
    part[tid][0][0] = gs[0]; part[tid][0][1] = gq[0];
    part[tid][1][0] = gs[1]; part[tid][1][1] = gq[1];
    __syncthreads();
    const int gps = swc / cpg, nact = cols * 64, nwaves = cols;   // 64 * cols threads = cols waves
    const int slice_g0 = blockIdx.y * gps;
    const int nch = HW >> 6;
    const int b = m_blk / HW, chunk = (m_blk - b * HW) >> 6;
    for (int gl = wave; gl < gps; gl += nwaves) {
        const int ga = slice_g0 + gl;
        float a = 0.f, q = 0.f;
        for (int t = lane; t < nact; t += 64) {
            const int tc = blockIdx.y * swc + (t % cols) * 4;
            const int k = ga - tc / cpg;
            if (k == 0 || k == 1) {
                a += part[t][k][0];
                q += part[t][k][1];
            }
        }
        a = wave_sum(a);
        q = wave_sum(q);
        if (lane == 0) {
            float* dst = gn_part + (((long long)b * nch + chunk) * groups + ga) * 2;
            dst[0] = a;
            dst[1] = q;
        }
    }
}


// ---- host side: what follows the GEMM launch (shared by idb_gemm and idb_hconv) -----------------------------------------
// slice width of idb_splitk_reduce_gn_kernel: the largest multiple of lcm(group width, 4) that is <= 64 and divides n (one thread
// per row and 4 channels: 64 * swc / 4 <= 1024 threads); 0 if there is none
inline int gn_reduce_slice(int n, int groups) {
    const int cpg = n / groups;
    int base = cpg;
    while (base % 4) base += cpg;
    int swc = 0;
    for (int c = base; c <= 64; c += base)
        if (n % c == 0) swc = c;
    return swc;
}


// the reduce launch may use 16-byte accesses
inline bool idb_reduce_vec_ok(const GemmParams& p, int n) {
    return n % 4 == 0 && p.out_ld % 4 == 0 && idb_aligned16(p.partial) && (!p.bias || idb_aligned16(p.bias)) &&
           (!p.sbias || (idb_aligned16(p.sbias) && p.sbias_ld % 4 == 0)) && (!p.res || ((uintptr_t)p.res & 7) == 0) &&
           ((uintptr_t)p.out & 15) == 0;
}

// Can the LDS-staged epilogue of a bm x bn tile emit the first GroupNorm pass of an [M][n] output itself (GemmParams::gn_part)?
// Whole groups per column tile, whole 64-row chunks per row tile, an even group width (dword pairs), at most 64 lanes per item.
inline bool idb_epilogue_emits_gn(int bm, int bn, int threads, long long M, int n, int groups) {
    if (groups <= 0 || n % groups || n % bn || bm % 64 || M % 64) return false;
    const int cpg = n / groups;
    if (cpg % 2 || bn % cpg) return false;
    const int items = (bm / 64) * (bn / cpg);
    return items <= threads && threads % items == 0 && ((threads / items) & (threads / items - 1)) == 0;
}

// Split-K reduce launch (+ the GroupNorm statistics of the output when asked for: from the reduce launch when the slabs were
// written in its window order, from the GEMM's own LDS-staged epilogue when there is no split (p.gn_part), by an extra statistics
// launch otherwise).
template <typename T>
int idb_finish_splitk(const GemmParams& p, int M, int n, int batch, int splitk, float* gn_partials, int gn_groups, int dtype, hipStream_t st) {
    if (gn_partials && !p.gn_part && (splitk == 1 || p.counters))
        return idb_launch_gn_stats64(p.out, n, batch, p.HW, gn_groups, gn_partials, dtype, st);   // no reduce launch to ride on
    if (splitk == 1 || p.counters) return IDB_OK;
    if (p.slab_swc) {
        const int swc = p.slab_swc;       // chosen together with the blocked slab layout the GEMM has just written
        hipLaunchKernelGGL((idb_splitk_reduce_gn_kernel<T>), dim3(M / 64, n / swc), dim3(16 * swc), 0, st, p.partial, splitk, M, n, p.HW,
                           p.scale, p.bias, p.sbias, p.sbias_ld, (const T*)p.res, (T*)p.out, p.out_ld, gn_partials, gn_groups, swc);
        IDB_CHECK_LAUNCH("idb_splitk_reduce_gn");
        return IDB_OK;
    }
    const int vec = idb_reduce_vec_ok(p, n) ? 4 : 1;
    const long long total = (long long)M * n / vec;
    const int blocks = (int)((total + 255) / 256);
    if (vec == 4)
        hipLaunchKernelGGL((idb_splitk_reduce_kernel<T, 4>), dim3(blocks), dim3(256), 0, st, p.partial, splitk, M, n, p.HW, p.scale, p.bias,
                           p.sbias, p.sbias_ld, (const T*)p.res, p.out, p.out_ld, p.out_f32);
    else
        hipLaunchKernelGGL((idb_splitk_reduce_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, p.partial, splitk, M, n, p.HW, p.scale, p.bias,
                           p.sbias, p.sbias_ld, (const T*)p.res, p.out, p.out_ld, p.out_f32);
    IDB_CHECK_LAUNCH("idb_splitk_reduce");
    if (gn_partials) return idb_launch_gn_stats64(p.out, n, batch, p.HW, gn_groups, gn_partials, dtype, st);
    return IDB_OK;
}

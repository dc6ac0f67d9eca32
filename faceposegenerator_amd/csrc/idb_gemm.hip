// K1/K2/K3 — implicit GEMM on MFMA for gfx950: conv3x3 (stride 1/2, nearest-2x upsample), 1x1 conv,
// Linear, with skip-concat / fused-shortcut K segments and bias / temb / residual / GEGLU epilogues.
//
// Upstream ops replaced (see include/idb_kernels.h): diffusers ResnetBlock2D.conv1/conv2/conv_shortcut,
// Downsample2D.conv, Upsample2D (interpolate + conv), Attention.to_q/k/v/to_out, FeedForward (GEGLU),
// Transformer2DModel.proj_in/proj_out — reached from /root/reference/inference_ID-Booth.py:138.
//
// Design (MI355X):
//   * NHWC activations: one K-step (64 channels of one tap) of an im2col row is ONE contiguous
//     128-byte line, so both operands stream HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA,
//     16 B/lane, no VGPR staging); zero padding = out-of-range buffer offsets (hardware zero fill).
//   * LDS tile rows are 128 B; chunk c of row r is stored at chunk position c ^ (r & 7) (swizzle
//     applied on the SOURCE address, LDS-DMA destinations are lane-linear) so every ds_read_b128
//     fragment read is bank-conflict free.
//   * v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate.  The weight fragment is the MFMA A operand
//     and the activation fragment the B operand, so each lane ends up with 4 CONSECUTIVE output
//     channels of one pixel -> 8-byte (bf16) / 16-byte (fp32) epilogue accesses.
//   * LDS ring of 2 stages (two workgroups per CU hide each other's latency) or 3 stages (a lone
//     workgroup per CU: two K-steps in flight), 8 waves per workgroup by default, next stage's DMA
//     issued before the current stage's MFMAs, counted vmcnt, one barrier per K-step; XCD-aware
//     bijective block remaps (runs of tiles per XCD, or one K-slice per XCD for split-K grids).
//   * grids smaller than the chip (the whole batch-1 UNet) use split-K sized to one workgroup per
//     CU, fp32 slabs + a reduce/epilogue launch (deterministic, no atomics) that can also emit the
//     first GroupNorm pass of its output.
#include "idb_common.h"
#include <stdlib.h>

#include "idb_gemm_epi.h"

// __launch_bounds__(256, 2): at most 256 VGPRs so that TWO workgroups share a CU — the second workgroup's MFMAs are what
// hides this one's LDS-DMA issue, waits and epilogue (one workgroup per CU measured 0.70 vs 1.12 PFLOP/s on the conv shape).
template <typename T, int MF, int NF, int NS, int WM = 2>   // WM wave rows x 2 wave columns; tile = (16*MF*WM) x (32*NF)
__global__ __launch_bounds__(128 * WM, 2) void idb_gemm_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource types are device-only)
    using V8 = typename Op<T>::v8;
    constexpr int BM = 16 * MF * WM, BN = 32 * NF;
    constexpr int THREADS = 128 * WM, RS = 16 * WM;          // staging: RS tile rows per wave-instruction sweep of the workgroup
    constexpr int NJ = (BN + RS - 1) / RS;                   // weight-row sweeps; the last may be partial (160 rows / 64): its surplus
    constexpr int STAGE = (BM + NJ * RS) * 128;              // rows are LDS padding filled with zeros (out-of-range voffset), so every
                                                             // wave issues the same number of loads and the counted vmcnt stays exact
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    // XCD-aware bijective remaps (workgroups are dealt round-robin over the 8 XCDs in linear-id order, so ids b and b+8 share
    // an XCD and its L2; the 8 L2s are not coherent and do not share lines).
    //  mode 0: each XCD gets a contiguous run of tiles, so neighbours re-use the same activation rows from that L2; the K
    //          split, if any, is the grid's z.
    //  mode 1 (split-K, S % 8 == 0) / mode 2 (S == 4): each XCD owns ONE K-slice (mode 2: half the tiles of one) of EVERY
    //          tile, so every weight and activation byte crosses the fabric once instead of once per XCD — on the batch-1
    //          weight-streaming layers (M = 512: 4 row tiles on 4 XCD pairs) mode 0 fetched the weights 4-8 times
    //          (rocprofv3 FETCH_SIZE: 99-113 MB per launch against 28-40 MB of operands).
    int wg, kz;
    if (p.xcd_mode == 0) {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        kz = blockIdx.z;
    } else {
        const int X = gridDim.x;
        const int lin = blockIdx.x + X * blockIdx.z;
        const int xcd = lin & 7, j = lin >> 3;
        if (p.xcd_mode == 1) {
            kz = xcd + 8 * (j / X);
            wg = j % X;
        } else {
            kz = xcd >> 1;
            wg = (xcd & 1) * (X >> 1) + j;
        }
    }
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kt0 = (int)(((long long)kz * p.ktiles) / p.splitk);          // balanced partition: slice sizes differ by at most one
    const int kt1 = (int)(((long long)(kz + 1) * p.ktiles) / p.splitk);
    const int nk = kt1 - kt0;

    // ---- per-thread staging coordinates: thread loads chunk position (tid&7) of rows (tid>>3)+32i;
    // the 16-byte chunk it fetches is (tid&7) ^ (row&7): the swizzle lives on the source address.
    const int lrow = tid >> 3;
    const unsigned cg16 = ((tid & 7) ^ (lrow & 7)) * 16;
    int a_b[MF], a_oy[MF], a_ox[MF];
    bool a_ok[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + i * RS + lrow;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (p.HW == 1) {                     // plain [M][K] matrix: no pixel decode (two integer divisions per row)
            a_b[i] = mm;
            a_oy[i] = a_ox[i] = 0;
        } else {
            a_b[i] = mm / p.HW;
            const int rem = mm - a_b[i] * p.HW;
            a_oy[i] = rem / p.OW;
            a_ox[i] = rem - a_oy[i] * p.OW;
        }
    }
    // weights: one descriptor, per-row voffset fixed for the whole K loop, K position in the SGPR soffset
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + idb_weight_group(p, m0) * p.w_group_stride), 0, p.w_bytes, IDB_RSRC_FLAGS);
    unsigned w_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + j * RS + lrow;
        w_voff[j] = (n < p.N && j * RS + lrow < BN) ? (unsigned)(n >> 4) * p.w_blk_bytes + (unsigned)(n & 15) * p.w_row_bytes + cg16 : IDB_OOB;
    }
    unsigned w_soff = (unsigned)kt0 * p.w_kstep;

    // ---- K-step state: source s, tap (0..8; a 1x1 source sits on the centre tap 4), channel offset c0.
    // Per-row voffsets (pixel address, zero padding -> out-of-range) are recomputed only when the tap or the
    // source changes (every C/64 K-steps); inside a tap the channel offset rides in the SGPR soffset, so a
    // K-step costs no address VALU at all.
    int s = 0, tap = 0, c0 = 0, cur_c = 64, tap_end = 9;
    {
        int rem = kt0;
        while (s < IDB_MAX_SRC - 1) {
            const int steps = p.src[s].taps * (p.src[s].C >> 6);
            if (rem < steps) break;
            rem -= steps;
            ++s;
        }
        const int cs = p.src[s].C >> 6;
        if (p.src[s].taps == 9) {
            tap = rem / cs;
            c0 = (rem - tap * cs) << 6;
        } else {
            tap = 4;
            c0 = rem << 6;
        }
    }
    __amdgpu_buffer_rsrc_t rs_a = rs_w;
    unsigned a_voff[MF];
    bool need_retap = true;
    auto retap = [&]() {
        const GemmSrcK S = p.src[s];
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
        cur_c = S.C;
        tap_end = S.taps == 9 ? 9 : 5;
        const int t3 = tap / 3;
        const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
        const int LH = S.H << S.up, LW = S.W << S.up;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;
            const int pix = (a_b[i] * S.H + (iy >> S.up)) * S.W + (ix >> S.up);
            a_voff[i] = ok ? (unsigned)pix * (unsigned)(S.C * 2) + cg16 : IDB_OOB;
        }
    };
    auto stage = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + BM * 128;
        if (need_retap) {
            retap();
            need_retap = false;
        }
        const unsigned a_soff = (unsigned)c0 * 2u;
#pragma unroll
        for (int i = 0; i < MF; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(sA + (i * THREADS + wave * 64) * 16), 16, a_voff[i], a_soff, 0, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + (j * THREADS + wave * 64) * 16), 16, w_voff[j], w_soff, 0, 0);
        w_soff += p.w_kstep;
        c0 += 64;
        if (c0 == cur_c) {
            c0 = 0;
            need_retap = true;
            if (++tap == tap_end) {
                if (s < IDB_MAX_SRC - 1) ++s;
                tap = p.src[s].taps == 9 ? 0 : 4;
            }
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- NS-deep LDS ring, one barrier per K-step.  At the top of iteration `it` tiles it .. it+NS-2 are in
    // flight; the counted vmcnt retires tile `it` (this wave's share), the barrier makes every wave's share
    // visible AND proves that all waves are done reading tile it-1, whose buffer the next DMA overwrites.
    constexpr int LOADS = MF + NJ;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) stage(st);
    // folded LayerNorm: this thread's share of its tile row's statistics, the loads in flight with the first operand tiles
    float2 ln_part = make_float2(0.f, 0.f);
    if (p.ln_stats) ln_part = idb_ln_row_partials<BM, THREADS>(p, m0, tid);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        if (NS > 2 && it + NS - 2 < nk)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * LOADS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (it + NS - 1 < nk && IDB_DBG(p.dbg_loop) != 2) stage(cur == 0 ? NS - 1 : cur - 1);
        if (IDB_DBG(p.dbg_loop) == 1) {
            cur = cur + 1 == NS ? 0 : cur + 1;
            continue;
        }
        const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
        const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ((ks * 4 + fg) ^ (fr & 7)) * 16;
            V8 af[MF], wf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const V8*)(sA + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *(const V8*)(sB + j * 16 * 128 + pos);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf[j], af[i], acc[i][j]);
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
    }

    idb_gemm_epilogue<T, MF, NF, WM>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, kz, p.ln_stats != nullptr, ln_part);
#endif
}


// ------------------------------------------------------------------------------------------------------------
// Loader-wave variant (round 3) for grids of at most ONE workgroup per CU (the whole batch-1 UNet).  In idb_gemm_kernel every
// wave issues its share of the next stage's LDS-DMA instructions and THEN reads fragments and issues MFMAs; with a lone
// workgroup per CU nothing hides the ~100 issue cycles of each DMA instruction (4-5 per wave and K-step) nor the counted wait in
// front of the barrier, and all eight waves go through those phases in lock step: a K-step takes 0.6-1.0 us against 0.13-0.27 us
// of MFMA time.  Here the workgroup has LW extra waves (one or two per SIMD) that do nothing but the address arithmetic, the
// DMA issue and the counted vmcnt wait; the 2*WM compute waves keep exactly the fragment mapping, accumulation order and epilogue
// of idb_gemm_kernel (results are bit-identical) but execute only barrier -> ds_read -> MFMA.  Both roles pass ONE s_barrier per
// K-step (no flags, no polling: every wave reaches the same nk barriers), the loaders leave after the loop — s_barrier counts only
// the surviving waves of a workgroup — and the compute waves run the shared epilogue among themselves.
//   ring: NS stages; at the top of K-step `it` the loaders wait until stage `it` has landed (counted vmcnt: stages it+1 ..
//   it+NS-2 stay in flight), the barrier publishes it and proves stage it-1 is read, then they issue stage it+NS-1 into that buffer.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int MF, int NF, int NS, int WM, int LW>
__global__ __launch_bounds__(128 * WM + 64 * LW) void idb_gemm_kernel_lw(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using V8 = typename Op<T>::v8;
    constexpr int BM = 16 * MF * WM, BN = 32 * NF;
    constexpr int CT = 128 * WM;                               // compute threads
    constexpr int LR = 8 * LW;                                 // tile rows per sweep of the loader waves
    constexpr int NA = BM / LR, NJ = (BN + LR - 1) / LR;       // sweeps per stage: A rows, weight rows (the last may be partial)
    static_assert(BM % LR == 0, "loader sweep must divide the row tile");
    constexpr int STAGE = (BM + NJ * LR) * 128;
    constexpr int LOADS = NA + NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 2 * WM;                        // wave-uniform (readfirstlane): a scalar branch, not an EXEC mask

    int wg, kz;
    if (p.xcd_mode == 0) {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        kz = blockIdx.z;
    } else {
        const int X = gridDim.x;
        const int lin = blockIdx.x + X * blockIdx.z;
        const int xcd = lin & 7, j = lin >> 3;
        if (p.xcd_mode == 1) {
            kz = xcd + 8 * (j / X);
            wg = j % X;
        } else {
            kz = xcd >> 1;
            wg = (xcd & 1) * (X >> 1) + j;
        }
    }
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)kz * p.ktiles) / p.splitk);
    const int kt1 = (int)(((long long)(kz + 1) * p.ktiles) / p.splitk);
    const int nk = kt1 - kt0;

    if (loader) {
        // ---------------- loader waves: addresses, LDS-DMA issue, counted waits ----------------
        const int lw = wave - 2 * WM;
        const int lt = tid - CT;                               // 0 .. 64*LW-1
        const int lrow = lt >> 3;                              // 0 .. LR-1
        const unsigned cg16 = ((lt & 7) ^ (lrow & 7)) * 16;    // swizzle on the source side (LR is a multiple of 8)
        int a_b[NA], a_oy[NA], a_ox[NA];
        bool a_ok[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + i * LR + lrow;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            if (p.HW == 1) {
                a_b[i] = mm;
                a_oy[i] = a_ox[i] = 0;
            } else {
                a_b[i] = mm / p.HW;
                const int rem = mm - a_b[i] * p.HW;
                a_oy[i] = rem / p.OW;
                a_ox[i] = rem - a_oy[i] * p.OW;
            }
        }
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + idb_weight_group(p, m0) * p.w_group_stride), 0, p.w_bytes, IDB_RSRC_FLAGS);
        unsigned w_voff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + j * LR + lrow;
            w_voff[j] = (n < p.N && j * LR + lrow < BN) ? (unsigned)(n >> 4) * p.w_blk_bytes + (unsigned)(n & 15) * p.w_row_bytes + cg16 : IDB_OOB;
        }
        unsigned w_soff = (unsigned)kt0 * p.w_kstep;
        int s = 0, tap = 0, c0 = 0, cur_c = 64, tap_end = 9;
        {
            int rem = kt0;
            while (s < IDB_MAX_SRC - 1) {
                const int steps = p.src[s].taps * (p.src[s].C >> 6);
                if (rem < steps) break;
                rem -= steps;
                ++s;
            }
            const int cs = p.src[s].C >> 6;
            if (p.src[s].taps == 9) {
                tap = rem / cs;
                c0 = (rem - tap * cs) << 6;
            } else {
                tap = 4;
                c0 = rem << 6;
            }
        }
        __amdgpu_buffer_rsrc_t rs_a = rs_w;
        unsigned a_voff[NA];
        bool need_retap = true;
        auto stage = [&](int buf) {
            char* sA = smem + buf * STAGE;
            char* sB = sA + BM * 128;
            if (need_retap) {
                const GemmSrcK S = p.src[s];
                rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
                cur_c = S.C;
                tap_end = S.taps == 9 ? 9 : 5;
                const int t3 = tap / 3;
                const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
                const int LH = S.H << S.up, LWd = S.W << S.up;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LWd;
                    const int pix = (a_b[i] * S.H + (iy >> S.up)) * S.W + (ix >> S.up);
                    a_voff[i] = ok ? (unsigned)pix * (unsigned)(S.C * 2) + cg16 : IDB_OOB;
                }
                need_retap = false;
            }
            const unsigned a_soff = (unsigned)c0 * 2u;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(sA + (i * 64 * LW + lw * 64) * 16), 16, a_voff[i], a_soff, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + (j * 64 * LW + lw * 64) * 16), 16, w_voff[j], w_soff, 0, 0);
            w_soff += p.w_kstep;
            c0 += 64;
            if (c0 == cur_c) {
                c0 = 0;
                need_retap = true;
                if (++tap == tap_end) {
                    if (s < IDB_MAX_SRC - 1) ++s;
                    tap = p.src[s].taps == 9 ? 0 : 4;
                }
            }
        };
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < nk) stage(st);
        int cur = 0;
        for (int it = 0; it < nk; ++it) {
            if (it + NS - 2 < nk)
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * LOADS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (it + NS - 1 < nk) stage(cur == 0 ? NS - 1 : cur - 1);
            cur = cur + 1 == NS ? 0 : cur + 1;
        }
        return;                                                // s_barrier counts the surviving waves only
    }

    // ---------------- compute waves: the fragment mapping and the epilogue of idb_gemm_kernel ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float2 ln_part = make_float2(0.f, 0.f);
    if (p.ln_stats) ln_part = idb_ln_row_partials<BM, CT>(p, m0, tid);
    if constexpr (MF >= 2) {
        // The K-step is rotated by half a step around the barrier: the fragments of the stage's SECOND 32-deep half are read before the
        // barrier that publishes the next stage and multiplied after it, while the next stage's first half is being read — every wave
        // leaves a barrier with 4*MF*NF/4 MFMAs of operands already in registers instead of with an LDS round trip in front of its
        // first MFMA (all waves of the workgroup pass the barrier together, so that round trip was exposed once per K-step).  The
        // protocol with the loaders is unchanged (barrier `it` publishes stage `it` and proves stage `it - 1` read: its reads are
        // drained by lgkmcnt(0) in front of the barrier) and so is the accumulation order: results stay bit-identical.
        const int pos0 = (fg ^ (fr & 7)) * 16, pos1 = pos0 ^ 64;
        const char* sA0 = smem + (wm * 16 * MF + fr) * 128;
        const char* sB0 = smem + BM * 128 + (wn * 16 * NF + fr) * 128;
        V8 af0[MF], wf0[NF], af1[MF], wf1[NF];
        asm volatile("s_barrier" ::: "memory");
    #pragma unroll
        for (int i = 0; i < MF; ++i) af0[i] = *(const V8*)(sA0 + i * 16 * 128 + pos0);
    #pragma unroll
        for (int j = 0; j < NF; ++j) wf0[j] = *(const V8*)(sB0 + j * 16 * 128 + pos0);
        int cur = 0;
        for (int it = 0; it + 1 < nk; ++it) {
            const char* sA = sA0 + cur * STAGE;
            const char* sB = sB0 + cur * STAGE;
    #pragma unroll
            for (int i = 0; i < MF; ++i) af1[i] = *(const V8*)(sA + i * 16 * 128 + pos1);
    #pragma unroll
            for (int j = 0; j < NF; ++j) wf1[j] = *(const V8*)(sB + j * 16 * 128 + pos1);
    #pragma unroll
            for (int i = 0; i < MF; ++i)
    #pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf0[j], af0[i], acc[i][j]);
            cur = cur + 1 == NS ? 0 : cur + 1;
            // lgkmcnt(0): this wave's fragment reads of stage `it` have returned before the loaders may overwrite its buffer.  As a
            // BUILTIN (0xC07F = lgkmcnt 0, vmcnt / expcnt untouched): the compiler's wait-count pass cannot see inside inline asm and
            // would otherwise make the first MFMAs behind the barrier wait for the reads issued behind it as well
            __builtin_amdgcn_s_waitcnt(0xC07F);
            asm volatile("s_barrier" ::: "memory");
            const char* nA = sA0 + cur * STAGE;
            const char* nB = sB0 + cur * STAGE;
    #pragma unroll
            for (int i = 0; i < MF; ++i) af0[i] = *(const V8*)(nA + i * 16 * 128 + pos0);
    #pragma unroll
            for (int j = 0; j < NF; ++j) wf0[j] = *(const V8*)(nB + j * 16 * 128 + pos0);
    #pragma unroll
            for (int i = 0; i < MF; ++i)
    #pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf1[j], af1[i], acc[i][j]);
        }
        {                                                          // the last K-step: no barrier behind its first half
            const char* sA = sA0 + cur * STAGE;
            const char* sB = sB0 + cur * STAGE;
    #pragma unroll
            for (int i = 0; i < MF; ++i) af1[i] = *(const V8*)(sA + i * 16 * 128 + pos1);
    #pragma unroll
            for (int j = 0; j < NF; ++j) wf1[j] = *(const V8*)(sB + j * 16 * 128 + pos1);
    #pragma unroll
            for (int i = 0; i < MF; ++i)
    #pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf0[j], af0[i], acc[i][j]);
    #pragma unroll
            for (int i = 0; i < MF; ++i)
    #pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf1[j], af1[i], acc[i][j]);
        }
    } else {
        // 64-row tiles (the short-K linears of the batch-1 UNet): plain order — barrier, both halves' reads, MFMAs (the rotated loop was
        // measured -0.3 % end to end at batch 1 with these tiles in it: nothing to hide behind 5-10 MFMAs per half)
        int cur = 0;
        for (int it = 0; it < nk; ++it) {
            // lgkmcnt(0): this wave's fragment reads of the previous stage have returned before the loaders may overwrite its buffer
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
            const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pos = ((ks * 4 + fg) ^ (fr & 7)) * 16;
                V8 af[MF], wf[NF];
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *(const V8*)(sA + i * 16 * 128 + pos);
#pragma unroll
                for (int j = 0; j < NF; ++j) wf[j] = *(const V8*)(sB + j * 16 * 128 + pos);
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf[j], af[i], acc[i][j]);
            }
            cur = cur + 1 == NS ? 0 : cur + 1;
        }
    }
    idb_gemm_epilogue<T, MF, NF, WM>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, kz, p.ln_stats != nullptr, ln_part);
#endif
}


// ------------------------------------------------------------------------------------------------------------
// Patch-resident 3x3 conv (round 3) for large grids: the 256-row loader-wave tile with the ACTIVATION operand fetched once per
// 64-channel chunk instead of once per tap.  A tile of 256 output pixels is whole image rows (W | 256), so for one channel chunk
// the nine taps read shifted windows of ONE halo patch — (rows + 2) x (W + 2) pixels x 128 B, <= 400 pixels = 50 KB — that sits in
// LDS while the K loop walks tap after tap over it: K order is chunk-major ([chunk][tap], the weight K-step index is
// tap * C/64 + chunk, an SGPR offset either way).  Per K-step the workgroup now moves patch/9 + BN*128 B into LDS instead of
// (256 + BN) * 128 B: 26 KB instead of 52 KB at BN = 160 — 195 instead of 97 FLOP per byte moved L2 -> LDS, and
// the same drop in LDS write traffic, which shares the LDS port with the fragment reads.
//   roles (wave index): [0, 8) MFMA — fragment mapping, epilogue and wave tile of idb_gemm_kernel_lw<T,4,NF,3,4,4>; [8, 10) weight
//   loaders (3-stage ring, counted vmcnt); [10, 12) patch loaders (double-buffered patch: chunk c+1 lands while the nine taps of
//   chunk c run; one vmcnt(0) per chunk).  ONE s_barrier per K-step for every role, no flags, no polling.
//   1x1 K segments (a ResnetBlock2D's conv_shortcut fused behind conv2) go through the same structure with a halo-less "patch" of the
//   256 tile pixels and one tap per chunk.  Zero padding = out-of-range buffer offsets, as everywhere.
// The accumulation order over K differs from the tap-major kernels, so results are equal to rounding, not bit-identical.
// ------------------------------------------------------------------------------------------------------------
// K-step g of the chunk-major K walk ([segment][chunk][tap]) -> segment s, chunk c, tap t, K-step base of that segment's weights
__device__ __forceinline__ void idb_patch_seek(const GemmParams& p, int g, int& s, int& c, int& t, int& base) {
    s = 0;
    base = 0;
    while (s < IDB_MAX_SRC - 1 && g >= p.src[s].taps * (p.src[s].C >> 6)) {
        const int steps = p.src[s].taps * (p.src[s].C >> 6);
        g -= steps;
        base += steps;
        ++s;
    }
    c = g / p.src[s].taps;
    t = g - c * p.src[s].taps;
}

// MF = 4: 256-row tiles, 3-stage weight ring (large grids).  MF = 1 / 2: 64- / 128-row tiles, 4-stage ring, split-K by whole chunks
// (the one-workgroup-per-CU plans of the batch-1 UNet): the patch is 3-4 image rows there, the saving is the 9x smaller activation
// stream of a K-step whose weight stream is unchanged.
// GN = true (north_star "conv3x3 + GroupNorm+SiLU fused"; ResnetBlock2D norm1+conv1 / norm2+conv2 via inference_ID-Booth.py:138): the
// patch loaders become FOUR waves that fetch the raw patch into registers, apply y = silu(x * k[c] + h[c]) — gn_apply_kernel's arithmetic
// on the same partial sums (k = rstd * gamma, h = beta - mean * k), so the MFMA waves read the operands idb_groupnorm would have written —
// and store it to the patch buffer, one slice between each pair of the chunk's nine barriers.  The patch is normalised ONCE per chunk (the
// tap-major idb_gemm_kernel_gn re-normalises every pixel for each of its nine taps and lost on every 3x3 conv); padding pixels stay zero.
// Every K segment is a normalised 3x3 source and a tile lies inside one sample (host: gemm_fuses_gn).
template <typename T, int MF, int NF, int NS, bool GN = false>
__global__ __launch_bounds__(GN ? 896 : 768) void idb_conv_patch_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using V8 = typename Op<T>::v8;
    constexpr int WM = 4;
    constexpr int BM = 64 * MF, BN = 32 * NF;
    constexpr int PL = GN ? 4 : 2;                             // patch-loader waves
    constexpr int PP_PAD = MF == 1 ? (GN ? 224 : 208) : MF == 2 ? (GN ? 288 : 272) : 400;   // patch pixels, padded to the patch loaders' sweep of 8 * PL pixels (host: conv_patch_ok)
    constexpr int PATCH = PP_PAD * 128;                        // bytes of one patch buffer
    constexpr int NPI = PP_PAD / (8 * PL);                     // 16-byte-per-lane loads per patch-loader wave and chunk
    constexpr int NJ = BN / 16;                                // weight-row sweeps of the two weight loaders (8 rows per instruction)
    constexpr int BSTAGE = BN * 128;
    static_assert(2 * PATCH + NS * BSTAGE + (GN ? 256 : 0) <= 160 * 1024 && (MF < 4 || !GN), "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem + 2 * PATCH;
    float2* gst = (float2*)(ring + NS * BSTAGE);               // GN: {mean, rstd} of the tile's sample, per group

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wg, kz;
    if (p.xcd_mode == 0) {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        kz = blockIdx.z;
    } else {                                                   // one K-slice per XCD (group): see idb_gemm_kernel
        const int X = gridDim.x;
        const int lin = blockIdx.x + X * blockIdx.z;
        const int xcd = lin & 7, j = lin >> 3;
        if (p.xcd_mode == 1) {
            kz = xcd + 8 * (j / X);
            wg = j % X;
        } else {
            kz = xcd >> 1;
            wg = (xcd & 1) * (X >> 1) + j;
        }
    }
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // this workgroup's K range in K-steps of the chunk-major walk (balanced; a split may start and end inside a chunk)
    const int g0 = (int)(((long long)kz * p.ktiles) / p.splitk), g1 = (int)(((long long)(kz + 1) * p.ktiles) / p.splitk);
    const int nk = g1 - g0;
    int s0, c0, t0, base0;
    idb_patch_seek(p, g0, s0, c0, t0, base0);
    // tile geometry: R image rows of width W; nimg whole images when the image is smaller than the tile
    const int W = p.OW, H = p.HW / W;
    const int nimg = p.HW >= BM ? 1 : BM / p.HW;
    const int RI = BM / W / nimg;                              // tile rows per image

    if constexpr (GN) {
        if (wave >= 10) {
            // ---------------- transforming patch loaders (registers -> normalise -> LDS) ----------------
            const int pw = wave - 10;
            const int q8 = lane >> 3;
            const int b0 = m0 / p.HW, y0 = (m0 - b0 * p.HW) / W;
            const int G = p.gn_in_groups, cpg = p.gn_in_c / G;
            const int co = (lane & 7) ^ q8;                    // this lane's channel octet inside every chunk (pixel q has q & 7 == q8)
            unsigned voff[NPI];
            bool okp[NPI];
            int seg = -1, cb = 0;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src[0].ptr, 0, p.src[0].bytes, IDB_RSRC_FLAGS);
            auto setup = [&](int s) {
                const GemmSrcK S = p.src[s];
                rs = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
                const int PW = W + 2, RIh = RI + 2;
                const int PP = RIh * PW;
#pragma unroll
                for (int i = 0; i < NPI; ++i) {
                    const int q = (i * PL + pw) * 8 + q8;
                    const int pr = q / PW, px = q - pr * PW;
                    const int y = y0 + pr - 1, x = px - 1;
                    okp[i] = q < PP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                    const int pix = (b0 * H + y) * W + x;
                    voff[i] = okp[i] ? (unsigned)pix * (unsigned)(S.C * 2) + (unsigned)(co * 16) : IDB_OOB;
                }
                cb = 0;
                for (int q = 0; q < s; ++q) cb += p.src[q].C;   // first channel of source s inside the normalised concatenation
                seg = s;
            };
            V8 raw[NPI];
            f32x4 g0v, g1v, b0v, b1v;
            float kk[8], hh[8];
            int ch0 = 0;
            auto fetch_gb = [&]() {                            // gamma / beta of this lane's 8 channels
                g0v = *(const f32x4*)(p.gn_in_gamma + ch0), g1v = *(const f32x4*)(p.gn_in_gamma + ch0 + 4);
                b0v = *(const f32x4*)(p.gn_in_beta + ch0), b1v = *(const f32x4*)(p.gn_in_beta + ch0 + 4);
            };
            auto fetch = [&](int s, int c) {                   // the raw patch: loads only
                if (s != seg) setup(s);
                ch0 = cb + c * 64 + co * 8;
                const unsigned soff = (unsigned)c * 128u;
#pragma unroll
                for (int i = 0; i < NPI; ++i) raw[i] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0));
            };
            auto coeffs = [&]() {                              // k = rstd * gamma, h = beta - mean * k (the first use of the loads above)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float2 st = gst[(ch0 + e) / cpg];
                    kk[e] = st.y * (e < 4 ? g0v[e] : g1v[e - 4]);
                    hh[e] = (e < 4 ? b0v[e] : b1v[e - 4]) - st.x * kk[e];
                }
            };
            auto put = [&](int i, int buf) {
                V8 o = raw[i];
                if (okp[i]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float y = to_f32<T>(raw[i][e]) * kk[e] + hh[e];
                        if (p.gn_in_silu) y = silu_f(y);
                        o[e] = from_f32<T>(y);
                    }
                }
                *(V8*)(smem + buf * PATCH + ((i * PL + pw) * 8 + q8) * 128 + (lane & 7) * 16) = o;
            };
            int s = s0, c = c0, t = t0, buf = 0, done = 0;
            fetch(s, c);                                       // the first patch is in flight while the statistics are added up
            {
                // {mean, rstd} per group of this tile's sample: 8 lanes per group add the pixel-chunk partials in gn_apply_kernel's order
                // (4 waves x 8 groups = one pass for GroupNorm(32))
                for (int g0_ = pw * 8; g0_ < G; g0_ += 8 * PL) {
                    const int g = min(g0_ + (lane >> 3), G - 1), sub = lane & 7;
                    f32x2 pv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int ch = min(sub + 8 * u, p.gn_in_chunks - 1);
                        pv[u] = *(const f32x2*)(p.gn_in_part + (((long long)b0 * p.gn_in_chunks + ch) * G + g) * 2);
                    }
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (sub + 8 * u < p.gn_in_chunks) {
                            a += pv[u][0];
                            q += pv[u][1];
                        }
                    }
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) {
                        a += __shfl_xor(a, o, 64);
                        q += __shfl_xor(q, o, 64);
                    }
                    if (g0_ + (lane >> 3) < G && sub == 0) {
                        const double cnt = (double)p.HW * cpg;
                        const double mean = (double)a / cnt;
                        double var = (double)q / cnt - mean * mean;
                        if (var < 0.0) var = 0.0;
                        gst[g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)p.gn_in_eps)));
                    }
                }
            }
            fetch_gb();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // barrier S: the statistics are in LDS
            coeffs();
#pragma unroll
            for (int i = 0; i < NPI; ++i) put(i, 0);
            while (done < nk) {
                const int n = 9 - t < nk - done ? 9 - t : nk - done;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // first K-step of the chunk: its normalised patch is in LDS
                done += n;
                int s2 = s, c2 = c + 1;
                if (c2 == (p.src[s].C >> 6)) { c2 = 0; if (s2 < IDB_MAX_SRC - 1) ++s2; }
                if (done < nk) {
                    fetch(s2, c2);
                    fetch_gb();
                    if (n == 9) {
                        // a whole chunk ahead: the loads fly during the chunk's first three K-steps, then one slice of the next patch is
                        // normalised and stored between each pair of the remaining barriers (the compiler's counted vmcnt: loads return in order)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if (k == 3) coeffs();
                            if (k >= 3) {
#pragma unroll
                                for (int i = (k - 3) * NPI / 5; i < (k - 2) * NPI / 5; ++i) put(i, buf ^ 1);
                            }
                            asm volatile("s_barrier" ::: "memory");
                        }
                    } else {                                   // a split boundary inside the chunk: everything at once
                        coeffs();
#pragma unroll
                        for (int i = 0; i < NPI; ++i) put(i, buf ^ 1);
                        for (int k = 1; k < n; ++k) asm volatile("s_barrier" ::: "memory");
                    }
                } else {
                    for (int k = 1; k < n; ++k) asm volatile("s_barrier" ::: "memory");
                }
                s = s2; c = c2; t = 0; buf ^= 1;
            }
            return;
        }
        if (wave < 8) asm volatile("s_barrier" ::: "memory");  // barrier S (MFMA waves; the weight loaders pass it behind their first stages)
    }
    if (!GN && wave >= 10) {
        // ---------------- patch loaders ----------------
        const int pw = wave - 10;
        const int q8 = lane >> 3;
        const int b0 = m0 / p.HW, y0 = (m0 - b0 * p.HW) / W;
        unsigned voff[NPI];
        int seg = -1, npi = 0;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src[0].ptr, 0, p.src[0].bytes, IDB_RSRC_FLAGS);
        auto setup = [&](int s) {
            const GemmSrcK S = p.src[s];
            rs = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
            const int h = S.taps == 9 ? 1 : 0;
            const int PW = W + 2 * h, RIh = RI + 2 * h;
            const int PP = nimg * RIh * PW;
            npi = (PP + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int q = (i * 2 + pw) * 8 + q8;
                const int pr = q / PW, px = q - pr * PW;
                const int img = pr / RIh, lr = pr - img * RIh;
                const int y = y0 + lr - h, x = px - h;
                const bool ok = q < PP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                const int pix = ((b0 + img) * H + y) * W + x;
                voff[i] = ok ? (unsigned)pix * (unsigned)(S.C * 2) + (unsigned)(((lane & 7) ^ q8) * 16) : IDB_OOB;
            }
            seg = s;
        };
        auto issue = [&](int s, int c, int buf) {
            if (s != seg) setup(s);
            char* dst = smem + buf * PATCH + pw * 1024;
            const unsigned soff = (unsigned)c * 128u;
#pragma unroll
            for (int i = 0; i < NPI; ++i)
                if (i < npi) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + i * 2048), 16, voff[i], soff, 0, 0);
        };
        int s = s0, c = c0, t = t0, buf = 0, done = 0;
        issue(s, c, 0);
        while (done < nk) {
            const int taps = p.src[s].taps, CS = p.src[s].C >> 6;
            const int n = taps - t < nk - done ? taps - t : nk - done;          // this split's K-steps inside the chunk
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // first of them: the chunk's patch has landed
            done += n;
            int s2 = s, c2 = c + 1;
            if (c2 == CS) { c2 = 0; if (s2 < IDB_MAX_SRC - 1) ++s2; }
            if (done < nk) issue(s2, c2, buf ^ 1);             // into the buffer of the PREVIOUS chunk: every read of it is behind the barrier
            for (int k = 1; k < n; ++k) asm volatile("s_barrier" ::: "memory");
            s = s2; c = c2; t = 0; buf ^= 1;
        }
        return;
    }
    if (wave >= 8) {
        // ---------------- weight loaders ----------------
        const int bw = wave - 8;
        const int lrow = lane >> 3;
        const unsigned cg16 = ((lane & 7) ^ lrow) * 16;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + idb_weight_group(p, m0) * p.w_group_stride), 0, p.w_bytes, IDB_RSRC_FLAGS);
        unsigned w_voff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + j * 16 + bw * 8 + lrow;
            w_voff[j] = n < p.N ? (unsigned)(n >> 4) * p.w_blk_bytes + (unsigned)(n & 15) * p.w_row_bytes + cg16 : IDB_OOB;
        }
        int s = s0, c = c0, tap = t0, base = base0;
        int taps = p.src[s].taps, CS = p.src[s].C >> 6;
        auto stage = [&](int buf) {
            char* sB = ring + buf * BSTAGE + bw * 1024;
            const unsigned soff = (unsigned)(base + tap * CS + c) * p.w_kstep;
#pragma unroll
            for (int j = 0; j < NJ; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + j * 2048), 16, w_voff[j], soff, 0, 0);
            if (++tap == taps) {
                tap = 0;
                if (++c == CS) {
                    c = 0;
                    base += taps * CS;
                    if (s < IDB_MAX_SRC - 1) ++s;
                    taps = p.src[s].taps;
                    CS = p.src[s].C >> 6;
                }
            }
        };
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < nk) stage(st);
        if constexpr (GN) asm volatile("s_barrier" ::: "memory");   // barrier S
        int cur = 0;
        for (int it = 0; it < nk; ++it) {
            if (it + NS - 2 < nk)
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * NJ) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (it + NS - 1 < nk) stage(cur == 0 ? NS - 1 : cur - 1);
            cur = cur + 1 == NS ? 0 : cur + 1;
        }
        return;
    }

    // ---------------- MFMA waves ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int rr[MF], xx[MF];                                        // tile row -> (patch row without halo, x)
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int ml = wm * 16 * MF + i * 16 + fr;
        const int r = ml / W;
        xx[i] = ml - r * W;
        rr[i] = r + 2 * (r / RI);                              // + 2 halo rows per image in front (9-tap segments only)
    }
    // K-step state (segment s, chunk c, tap t) and the half-step rotation around the barrier of idb_gemm_kernel_lw: the second 32-deep
    // half of a K-step is read before the barrier that publishes the next one and multiplied behind it
    int cur = 0, pb = 0, s = s0, c = c0, t = t0, ky = t0 / 3, kx = t0 - 3 * (t0 / 3);
    int taps = p.src[s].taps, CS = p.src[s].C >> 6, h = taps == 9 ? 1 : 0, PW = W + 2 * h;
    int pp0[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) pp0[i] = h ? rr[i] * PW + xx[i] : wm * 16 * MF + i * 16 + fr;
    const int posw0 = (fg ^ (fr & 7)) * 16;
    const char* sBw = ring + (wn * 16 * NF + fr) * 128;
    unsigned a_addr[MF];
    auto addr = [&]() {
        const int toff = ky * PW + kx;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int pp = pp0[i] + toff;
            a_addr[i] = (unsigned)(pb * PATCH) + (unsigned)pp * 128u + (unsigned)((fg ^ (pp & 7)) << 4);
        }
    };
    auto advance = [&]() {
        cur = cur + 1 == NS ? 0 : cur + 1;
        if (++kx == 3) { kx = 0; ++ky; }
        if (++t == taps) {
            t = 0; ky = 0; kx = 0;
            pb ^= 1;
            if (++c == CS) {
                c = 0;
                if (s < IDB_MAX_SRC - 1) ++s;
                taps = p.src[s].taps;
                CS = p.src[s].C >> 6;
                h = taps == 9 ? 1 : 0;
                PW = W + 2 * h;
#pragma unroll
                for (int i = 0; i < MF; ++i) pp0[i] = h ? rr[i] * PW + xx[i] : wm * 16 * MF + i * 16 + fr;
            }
        }
    };
    V8 af0[MF], wf0[NF], af1[MF], wf1[NF];
    asm volatile("s_barrier" ::: "memory");
    addr();
#pragma unroll
    for (int i = 0; i < MF; ++i) af0[i] = *(const V8*)(smem + a_addr[i]);
#pragma unroll
    for (int j = 0; j < NF; ++j) wf0[j] = *(const V8*)(sBw + j * 16 * 128 + posw0);
    for (int it = 0; it + 1 < nk; ++it) {
        const char* sB = sBw + cur * BSTAGE;
#pragma unroll
        for (int i = 0; i < MF; ++i) af1[i] = *(const V8*)(smem + (a_addr[i] ^ 64u));
#pragma unroll
        for (int j = 0; j < NF; ++j) wf1[j] = *(const V8*)(sB + j * 16 * 128 + (posw0 ^ 64));
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf0[j], af0[i], acc[i][j]);
        advance();
        addr();
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0), as a builtin (see idb_gemm_kernel_lw)
        asm volatile("s_barrier" ::: "memory");
        const char* nB = sBw + cur * BSTAGE;
#pragma unroll
        for (int i = 0; i < MF; ++i) af0[i] = *(const V8*)(smem + a_addr[i]);
#pragma unroll
        for (int j = 0; j < NF; ++j) wf0[j] = *(const V8*)(nB + j * 16 * 128 + posw0);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf1[j], af1[i], acc[i][j]);
    }
    {
        const char* sB = sBw + cur * BSTAGE;
#pragma unroll
        for (int i = 0; i < MF; ++i) af1[i] = *(const V8*)(smem + (a_addr[i] ^ 64u));
#pragma unroll
        for (int j = 0; j < NF; ++j) wf1[j] = *(const V8*)(sB + j * 16 * 128 + (posw0 ^ 64));
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf0[j], af0[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf1[j], af1[i], acc[i][j]);
    }
    idb_gemm_epilogue<T, MF, NF, WM>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, kz, false, make_float2(0.f, 0.f));
#endif
}


// ------------------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) -> conv / linear in ONE kernel (round 3; north_star "conv3x3 + GroupNorm+SiLU fused"; diffusers ResnetBlock2D
// norm1+conv1, norm2+conv2(+shortcut), Transformer2DModel norm+proj_in via inference_ID-Booth.py:138) on the loader-wave structure:
// a THIRD role, NV normalizer waves, transforms each landed A stage in place in LDS — y = silu(x * k[b][c] + h[b][c]), the exact
// arithmetic of gn_apply_kernel (k = rstd * gamma, h = beta - mean * k from the same partial sums, added in the same order, so the
// MFMA waves read bit-identical operands) — one K-step ahead of the MFMA waves.  The normalised tensor never exists in HBM and the
// gn_apply launch (60 per CFG forward, 6.4 us each at batch 1) disappears.  Round 2's idb_hconv did this transform on the MFMA
// waves themselves and lost; here it runs on waves that do nothing else, so the MFMA waves' stream is unchanged.
//   roles (wave index): [0, 2*WM) MFMA, [2*WM, 2*WM + 4) loaders, then NV normalizers.  Every wave passes the same barriers:
//   two while the normalizers build the {k, h} table in LDS (group statistics, then per channel), one that publishes the
//   normalised stage 0, then one per K-step.  Ring of NS stages: at K-step `it` the MFMA waves read stage it, the normalizers
//   transform stage it+1 (landed: the loaders waited for it before the barrier), stages it+2 .. are in flight.
//   Zero padding stays zero: rows whose tap falls outside the image are skipped (silu(h) != 0).  Sources beyond gn_in_nsrc (the
//   raw inputs of a fused 1x1 shortcut) pass untouched.  A tile lies inside one sample, or covers whole samples (host check).
// ------------------------------------------------------------------------------------------------------------
template <typename T, int MF, int NF, int NS, int WM, int NV>
__global__ __launch_bounds__(128 * WM + 256 + 64 * NV) void idb_gemm_kernel_gn(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using V8 = typename Op<T>::v8;
    constexpr int LW = 4;
    constexpr int BM = 16 * MF * WM, BN = 32 * NF;
    constexpr int CT = 128 * WM, LT = 64 * LW, VT = 64 * NV;
    constexpr int LR = 8 * LW;
    constexpr int NA = BM / LR, NJ = (BN + LR - 1) / LR;
    constexpr int STAGE = (BM + NJ * LR) * 128;
    constexpr int LOADS = NA + NJ;
    constexpr int NI = BM * 8 / VT;                            // 16-byte items of an A stage per normalizer lane
    static_assert(BM % LR == 0 && (BM * 8) % VT == 0, "tile / role geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tab = (float2*)(smem + NS * STAGE);                // [nsamp][gn_in_c] {k, h}
    const int hw_s = p.HW;
    const int nsamp = BM > hw_s ? BM / hw_s : 1;               // host: HW % BM == 0 or BM % HW == 0
    float2* gstat = tab + nsamp * p.gn_in_c;                   // [nsamp][groups] {mean, rstd}

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int wg, kz;
    if (p.xcd_mode == 0) {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        kz = blockIdx.z;
    } else {
        const int X = gridDim.x;
        const int lin = blockIdx.x + X * blockIdx.z;
        const int xcd = lin & 7, j = lin >> 3;
        if (p.xcd_mode == 1) {
            kz = xcd + 8 * (j / X);
            wg = j % X;
        } else {
            kz = xcd >> 1;
            wg = (xcd & 1) * (X >> 1) + j;
        }
    }
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)kz * p.ktiles) / p.splitk);
    const int kt1 = (int)(((long long)(kz + 1) * p.ktiles) / p.splitk);
    const int nk = kt1 - kt0;

    // K-step cursor shared by loaders and normalizers: source s, tap, channel offset c0
    int s = 0, tap = 0, c0 = 0;
    {
        int rem = kt0;
        while (s < IDB_MAX_SRC - 1) {
            const int steps = p.src[s].taps * (p.src[s].C >> 6);
            if (rem < steps) break;
            rem -= steps;
            ++s;
        }
        const int cs = p.src[s].C >> 6;
        if (p.src[s].taps == 9) {
            tap = rem / cs;
            c0 = (rem - tap * cs) << 6;
        } else {
            tap = 4;
            c0 = rem << 6;
        }
    }
    auto advance = [&]() {                                     // to the next K-step
        c0 += 64;
        if (c0 == p.src[s].C) {
            c0 = 0;
            if (++tap == (p.src[s].taps == 9 ? 9 : 5)) {
                if (s < IDB_MAX_SRC - 1) ++s;
                tap = p.src[s].taps == 9 ? 0 : 4;
            }
        }
    };

    if (wave >= 2 * WM + LW) {
        // ---------------- normalizer waves ----------------
        const int nt = tid - CT - LT;
        const int b_first = m0 / hw_s;
        const int G = p.gn_in_groups, cpg = p.gn_in_c / G;
        // phase 1: {mean, rstd} per (sample, group): 8 lanes per pair add the pixel-chunk partials in gn_apply_kernel's order
        for (int pr0 = 0; pr0 < nsamp * G; pr0 += VT / 8) {
            const int pr = pr0 + (nt >> 3), sub = nt & 7;
            const int prc = min(pr, nsamp * G - 1);
            const int sm = prc / G, g = prc - sm * G;
            const int b = min(b_first + sm, (p.M - 1) / hw_s);
            f32x2 pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = min(sub + 8 * u, p.gn_in_chunks - 1);
                pv[u] = *(const f32x2*)(p.gn_in_part + (((long long)b * p.gn_in_chunks + ch) * G + g) * 2);
            }
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (sub + 8 * u < p.gn_in_chunks) {
                    a += pv[u][0];
                    q += pv[u][1];
                }
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (pr < nsamp * G && sub == 0) {
                const double cnt = (double)hw_s * cpg;
                const double mean = (double)a / cnt;
                double var = (double)q / cnt - mean * mean;
                if (var < 0.0) var = 0.0;
                gstat[pr] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)p.gn_in_eps)));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                     // barrier A
        // phase 2: k = rstd * gamma, h = beta - mean * k per (sample, channel)
        for (int i = nt; i < nsamp * p.gn_in_c; i += VT) {
            const int sm = i / p.gn_in_c, c = i - sm * p.gn_in_c;
            const float2 st = gstat[sm * G + c / cpg];
            const float k = st.y * p.gn_in_gamma[c];
            tab[i] = make_float2(k, p.gn_in_beta[c] - st.x * k);
        }
        // rows of this lane's items: item j = row j * (VT / 8) + (nt >> 3), LDS chunk position nt & 7
        int r_b[NI], r_oy[NI], r_ox[NI];
        bool r_ok[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = j * (VT / 8) + (nt >> 3);
            const int m = m0 + row;
            r_ok[j] = m < p.M;
            const int mm = r_ok[j] ? m : 0;
            r_b[j] = mm / hw_s;
            const int rem = mm - r_b[j] * hw_s;
            r_oy[j] = rem / p.OW;
            r_ox[j] = rem - r_oy[j] * p.OW;
            r_b[j] -= b_first;
        }
        int cb = 0;                                            // first channel of source s inside the normalised concatenation
        for (int q = 0; q < s; ++q) cb += p.src[q].C;
        int s_seen = s;
        const int pos = nt & 7;
        auto transform = [&](int buf) {
            if (s != s_seen) {                                 // entered the next source
                cb += p.src[s_seen].C;
                s_seen = s;
            }
            if (s < p.gn_in_nsrc) {
                const GemmSrcK S = p.src[s];
                const int t3 = tap / 3;
                const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
                const int LH = S.H << S.up, LWd = S.W << S.up;
                char* sA = smem + buf * STAGE;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int row = j * (VT / 8) + (nt >> 3);
                    const int iy = r_oy[j] * p.stride + dy, ix = r_ox[j] * p.stride + dx;
                    const bool ok = r_ok[j] && (S.taps == 1 || ((unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LWd));
                    if (ok) {
                        const int c = cb + c0 + 8 * (pos ^ (row & 7));
                        const f32x4* tp = (const f32x4*)(tab + r_b[j] * p.gn_in_c + c);
                        const f32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3v = tp[3];      // {k,h} x 8 channels
                        V8* px = (V8*)(sA + row * 128 + pos * 16);
                        const V8 raw = *px;
                        const float ks[8] = {t0[0], t0[2], t1[0], t1[2], t2[0], t2[2], t3v[0], t3v[2]};
                        const float kh[8] = {t0[1], t0[3], t1[1], t1[3], t2[1], t2[3], t3v[1], t3v[3]};
                        V8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float y = to_f32<T>(raw[e]) * ks[e] + kh[e];
                            if (p.gn_in_silu) y = silu_f(y);
                            o[e] = from_f32<T>(y);
                        }
                        *px = o;
                    }
                }
            }
            advance();
        };
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                     // barrier B: table complete, stage 0 landed
        if (nk > 0) transform(0);
        int cur = 0;
        for (int it = 0; it < nk; ++it) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                 // barrier C_it
            const int nxt = cur + 1 == NS ? 0 : cur + 1;
            if (it + 1 < nk) transform(nxt);
            cur = nxt;
        }
        return;
    }

    if (wave >= 2 * WM) {
        // ---------------- loader waves ----------------
        const int lw = wave - 2 * WM;
        const int lt = tid - CT;
        const int lrow = lt >> 3;
        const unsigned cg16 = ((lt & 7) ^ (lrow & 7)) * 16;
        int a_b[NA], a_oy[NA], a_ox[NA];
        bool a_ok[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + i * LR + lrow;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            a_b[i] = mm / p.HW;
            const int rem = mm - a_b[i] * p.HW;
            a_oy[i] = rem / p.OW;
            a_ox[i] = rem - a_oy[i] * p.OW;
        }
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + idb_weight_group(p, m0) * p.w_group_stride), 0, p.w_bytes, IDB_RSRC_FLAGS);
        unsigned w_voff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + j * LR + lrow;
            w_voff[j] = (n < p.N && j * LR + lrow < BN) ? (unsigned)(n >> 4) * p.w_blk_bytes + (unsigned)(n & 15) * p.w_row_bytes + cg16 : IDB_OOB;
        }
        unsigned w_soff = (unsigned)kt0 * p.w_kstep;
        __amdgpu_buffer_rsrc_t rs_a = rs_w;
        unsigned a_voff[NA];
        int last_s = -1, last_tap = -1;
        auto stage = [&](int buf) {
            char* sA = smem + buf * STAGE;
            char* sB = sA + BM * 128;
            if (s != last_s || tap != last_tap) {
                const GemmSrcK S = p.src[s];
                rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
                const int t3 = tap / 3;
                const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
                const int LH = S.H << S.up, LWd = S.W << S.up;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LWd;
                    const int pix = (a_b[i] * S.H + (iy >> S.up)) * S.W + (ix >> S.up);
                    a_voff[i] = ok ? (unsigned)pix * (unsigned)(S.C * 2) + cg16 : IDB_OOB;
                }
                last_s = s;
                last_tap = tap;
            }
            const unsigned a_soff = (unsigned)c0 * 2u;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(sA + (i * 64 * LW + lw * 64) * 16), 16, a_voff[i], a_soff, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + (j * 64 * LW + lw * 64) * 16), 16, w_voff[j], w_soff, 0, 0);
            w_soff += p.w_kstep;
            advance();
        };
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < nk) stage(st);
        asm volatile("s_barrier" ::: "memory");                                             // barrier A
        if (NS - 1 <= nk)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * LOADS) : "memory");   // barrier B: stage 0 landed
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        int cur = 0;
        for (int it = 0; it < nk; ++it) {
            // stage it+1 landed (the normalizers transform it next); stages it+2 .. it+NS-2 stay in flight
            if (it + NS - 2 < nk)
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 3) * LOADS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (it + NS - 1 < nk) stage(cur == 0 ? NS - 1 : cur - 1);
            cur = cur + 1 == NS ? 0 : cur + 1;
        }
        return;
    }

    // ---------------- MFMA waves ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_barrier" ::: "memory");                                                 // barrier A
    asm volatile("s_barrier" ::: "memory");                                                 // barrier B
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                     // barrier C_it
        const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
        const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ((ks * 4 + fg) ^ (fr & 7)) * 16;
            // one weight fragment live at a time: with 4 MFMA waves the wave tile is 2x that of the 8-wave kernels and the workgroup's
            // 16 waves cap a wave at 128 VGPRs (80 of them accumulators at 128x160)
            V8 af[MF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const V8*)(sA + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const V8 wf = *(const V8*)(sB + j * 16 * 128 + pos);
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[i][j] = Op<T>::mfma16(wf, af[i], acc[i][j]);
            }
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    idb_gemm_epilogue<T, MF, NF, WM>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, kz, false, make_float2(0.f, 0.f));
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Register-staged variant: identical tiling, addressing and epilogue, but the operands go HBM/L2 -> VGPR
// (buffer_load_dwordx4, asynchronous until its first use) -> LDS (ds_write_b128 after the MFMAs of the current
// tile).  Measured on MI355X: one `buffer_load ... lds` costs ~85 issue cycles, nine of them per K-step are as long
// as the 40 MFMAs they feed and serialise with them inside a wave; a register load + ds_write_b128 pair costs ~20.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned idb_u32x4;

template <typename T, int MF, int NF>
__global__ __launch_bounds__(256, 2) void idb_gemm_kernel_rs(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using V8 = typename Op<T>::v8;
    constexpr int BM = 32 * MF, BN = 32 * NF, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kz = blockIdx.z;
    const int kt0 = (int)(((long long)kz * p.ktiles) / p.splitk);
    const int kt1 = (int)(((long long)(kz + 1) * p.ktiles) / p.splitk);
    const int nk = kt1 - kt0;

    // thread stages chunk (tid&7) of rows (tid>>3)+32i: global chunk c lands at LDS chunk position c ^ (row&7)
    const int lrow = tid >> 3;
    const unsigned c16 = (tid & 7) * 16;
    const int lds_off = lrow * 128 + (((tid & 7) ^ (lrow & 7)) * 16);
    int a_b[MF], a_oy[MF], a_ox[MF];
    bool a_ok[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + i * 32 + lrow;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (p.HW == 1) {
            a_b[i] = mm;
            a_oy[i] = a_ox[i] = 0;
        } else {
            a_b[i] = mm / p.HW;
            const int rem = mm - a_b[i] * p.HW;
            a_oy[i] = rem / p.OW;
            a_ox[i] = rem - a_oy[i] * p.OW;
        }
    }
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + idb_weight_group(p, m0) * p.w_group_stride), 0, p.w_bytes, IDB_RSRC_FLAGS);
    unsigned w_voff[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = n0 + j * 32 + lrow;
        w_voff[j] = n < p.N ? (unsigned)(n >> 4) * p.w_blk_bytes + (unsigned)(n & 15) * p.w_row_bytes + c16 : IDB_OOB;
    }
    unsigned w_soff = (unsigned)kt0 * p.w_kstep;

    int s = 0, tap = 0, c0 = 0, cur_c = 64, tap_end = 9;
    {
        int rem = kt0;
        while (s < IDB_MAX_SRC - 1) {
            const int steps = p.src[s].taps * (p.src[s].C >> 6);
            if (rem < steps) break;
            rem -= steps;
            ++s;
        }
        const int cs = p.src[s].C >> 6;
        if (p.src[s].taps == 9) {
            tap = rem / cs;
            c0 = (rem - tap * cs) << 6;
        } else {
            tap = 4;
            c0 = rem << 6;
        }
    }
    __amdgpu_buffer_rsrc_t rs_a = rs_w;
    unsigned a_voff[MF];
    bool need_retap = true;
    idb_u32x4 areg[MF], wreg[NF];
    auto gload = [&]() {
        if (need_retap) {
            const GemmSrcK S = p.src[s];
            rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
            cur_c = S.C;
            tap_end = S.taps == 9 ? 9 : 5;
            const int t3 = tap / 3;
            const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
            const int LH = S.H << S.up, LW = S.W << S.up;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;
                const int pix = (a_b[i] * S.H + (iy >> S.up)) * S.W + (ix >> S.up);
                a_voff[i] = ok ? (unsigned)pix * (unsigned)(S.C * 2) + c16 : IDB_OOB;
            }
            need_retap = false;
        }
        const unsigned a_soff = (unsigned)c0 * 2u;
#pragma unroll
        for (int i = 0; i < MF; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[i], a_soff, 0);
#pragma unroll
        for (int j = 0; j < NF; ++j) wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff[j], w_soff, 0);
        w_soff += p.w_kstep;
        c0 += 64;
        if (c0 == cur_c) {
            c0 = 0;
            need_retap = true;
            if (++tap == tap_end) {
                if (s < IDB_MAX_SRC - 1) ++s;
                tap = p.src[s].taps == 9 ? 0 : 4;
            }
        }
    };
    auto lstore = [&](int buf) {
        char* sA = smem + buf * STAGE + lds_off;
        char* sB = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < MF; ++i) *(idb_u32x4*)(sA + i * 32 * 128) = areg[i];
#pragma unroll
        for (int j = 0; j < NF; ++j) *(idb_u32x4*)(sB + j * 32 * 128) = wreg[j];
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        gload();
        lstore(0);
        __syncthreads();
    }
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < nk) gload();                      // next tile -> registers, in flight under the MFMAs below
        const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
        const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ((ks * 4 + fg) ^ (fr & 7)) * 16;
            V8 af[MF], wf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const V8*)(sA + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *(const V8*)(sB + j * 16 * 128 + pos);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf[j], af[i], acc[i][j]);
        }
        if (it + 1 < nk) lstore(cur ^ 1);               // the other buffer was last read one barrier ago
        __syncthreads();
    }
    idb_gemm_epilogue<T, MF, NF>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, kz);
#endif
}

// Epilogue of the persistent variant: same arithmetic as the direct epilogue, but EVERY lane issues exactly one
// buffer_store_dwordx2 per (value) fragment — rows/columns outside the matrix get an out-of-range offset and are
// dropped by the buffer range check — so the store count per wave is a compile-time constant and the K loop can wait
// for the next tile's loads with a counted vmcnt while these stores are still in flight.
template <typename T, int MF, int NF, bool GEGLU>
__device__ __forceinline__ void idb_pl_epilogue(const GemmParams& p, f32x4 (&acc)[MF][NF], const f32x4 (&cb)[NF], const f32x4 (&cu)[NF],
                                                const float2 (&mr)[MF], int m0, int n0, int wm, int wn, int fr, int fg) {
    using V4 = typename Op<T>::v4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, IDB_RSRC_FLAGS);
    const bool ln = p.ln_stats != nullptr;              // folded LayerNorm: out = rstd (scale acc - mean u) + v, cb = ln_v, cu = ln_u
    const bool hb = p.bias != nullptr || ln;            // cb holds dummy reads otherwise
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + (wm * MF + i) * 16 + fr;
        const bool mok = m < p.M;
        const int mc = mok ? m : 0;
        const float* sb = p.sbias ? p.sbias + (long long)(mc / p.HW) * p.sbias_ld : nullptr;
        const float rs = ln ? mr[i].y * p.scale : p.scale, kb = ln ? -mr[i].x * mr[i].y : 0.f;
#pragma unroll
        for (int j = 0; j < NF; j += (GEGLU ? 2 : 1)) {
            const int n = n0 + (wn * NF + j) * 16 + fg * 4;               // packed row (value part for GEGLU)
            const int oc = GEGLU ? (n0 + (wn * NF + j) * 16) / 2 + fg * 4 : n;
            const bool nok = GEGLU ? (n + 16 < p.N) : (n < p.N);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e] * rs + (ln ? kb * cu[j][e] : 0.f);
            if constexpr (GEGLU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float gt = acc[i][j + 1][e] * rs + (ln ? kb * cu[j + 1][e] : 0.f);
                    o[e] += hb ? cb[j][e] : 0.f;                        // columns past N are never stored
                    gt += hb ? cb[j + 1][e] : 0.f;
                    o[e] *= gelu_erf_f(gt);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += hb ? cb[j][e] : 0.f;
                if (nok && sb) {
                    const f32x4 b4 = *(const f32x4*)(sb + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += b4[e];
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gelu_erf_f(o[e]);
                }
                if (p.res && mok && nok) {
                    const V4 r4 = *(const V4*)((const T*)p.res + (long long)m * p.out_ld + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += to_f32<T>(r4[e]);
                }
            }
            const V4 pk = {from_f32<T>(o[0]), from_f32<T>(o[1]), from_f32<T>(o[2]), from_f32<T>(o[3])};
            const unsigned voff = (mok && nok) ? (unsigned)(((long long)m * p.out_ld + oc) * 2) : IDB_OOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, pk), rs_o, voff, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Persistent variant for plain [M][K] x [N][K]^T GEMMs with a SHORT K loop (the K = C projection layers of the
// transformer blocks at large batch: 5-20 K-steps).  A grid of 2 workgroups per CU walks the output tiles; the
// 2-deep LDS ring runs straight across tile boundaries, so the first K tile of output tile t+1 is already in
// flight while tile t's accumulators go through the epilogue, and no workgroup launch / address prologue /
// pipeline fill is paid per tile.  Same fragment layout and epilogue arithmetic as idb_gemm_kernel.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int MF, int NF>
__global__ __launch_bounds__(256, 2) void idb_gemm_kernel_pl(const GemmParams p) {   // 2 waves/SIMD: two workgroups per CU
#if defined(__HIP_DEVICE_COMPILE__)
    using V8 = typename Op<T>::v8;
    constexpr int BM = 32 * MF, BN = 32 * NF, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * p.tiles_n;
    const int G = gridDim.x, KT = p.ktiles;
    // Tiles b, b + G, ...: the 20-80 column tiles of one row tile run on all eight XCDs at once and every L2 fetches that A tile for
    // itself (FETCH_SIZE x 2 = 2.86 GB per launch against 337 MB of operands, GEGLU K = 320 at B_eff 128).  Measured alternative: each XCD
    // owning whole row tiles (its 64 workgroups walking [tm][tn] in order) cut that to 517 MB and was SLOWER — 575 vs 593 TFLOP/s at
    // K = 320, 719 vs 795 at K = 640, 832 vs 922 at K = 1280, batch 64 15.30 vs 15.37 images/s: the re-reads are served by the
    // Infinity Cache, and 64 workgroups pulling the same three A tiles through one L2 at the same moment queue on its channels.
    // Also measured: a 64x128 form with three workgroups per CU (more loads in flight against the one-round-trip K-step): 480 vs 574,
    // 624 vs 756, 676 vs 896 TFLOP/s — half the rows per weight tile costs more than the third workgroup hides.
    if ((int)blockIdx.x >= ntiles) return;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * KT;

    const int lrow = tid >> 3;
    const unsigned cg16 = ((tid & 7) ^ (lrow & 7)) * 16;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src[0].ptr, 0, p.src[0].bytes, IDB_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, IDB_RSRC_FLAGS);
    const unsigned a_row_bytes = (unsigned)p.src[0].C * 2u;        // A is [M][K], K = C of the single 1x1 source

    // load-side cursor: output tile lt, K-step lk
    int lt = blockIdx.x, lk = 0;
    unsigned a_voff[MF], w_voff[NF];
    auto set_tile = [&](int tile) {
        const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = tm * BM + i * 32 + lrow;
            a_voff[i] = m < p.M ? (unsigned)m * a_row_bytes + cg16 : IDB_OOB;
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = tn * BN + j * 32 + lrow;
            w_voff[j] = n < p.N ? (unsigned)(n >> 4) * p.w_blk_bytes + (unsigned)(n & 15) * p.w_row_bytes + cg16 : IDB_OOB;
        }
    };
    set_tile(lt);
    auto stage = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + BM * 128;
        const unsigned soff = (unsigned)lk * 128u;
#pragma unroll
        for (int i = 0; i < MF; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(sA + (i * 256 + wave * 64) * 16), 16, a_voff[i], soff, 0, 0);
        const unsigned wsoff = (unsigned)lk * p.w_kstep;
#pragma unroll
        for (int j = 0; j < NF; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + (j * 256 + wave * 64) * 16), 16, w_voff[j], wsoff, 0, 0);
        if (++lk == KT) {
            lk = 0;
            lt += G;
            if (lt < ntiles) set_tile(lt);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int ct = blockIdx.x, ck = 0;        // compute-side cursor
    // bias vectors of this lane's NF fragments, loaded at the FIRST K-step of every output tile (one round trip under the tile's
    // MFMAs): inside the epilogue they sat behind a runtime branch and were awaited one fragment at a time
    f32x4 cb[NF], cu[NF];
    const bool ln = p.ln_stats != nullptr;
    const float* bias_or_dummy = ln ? p.ln_v : (p.bias ? p.bias : (const float*)p.w);
    const bool has_cb = ln || p.bias;
    // folded LayerNorm: the ln_nt partial {sum, sum of squares} of a tile row are split over its four fg lane groups (two loads
    // per lane and row, ln_nt <= 8; more: an accumulating loop), loaded RAW at the tile's first K-step and reduced at the second
    // (behind that step's vmcnt(0) + barrier: no exposed round trip); {mean, rstd} then wait in mr for the epilogue
    float2 sp[MF][2], mr[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) mr[i] = make_float2(0.f, 1.f);
    int stores_in_flight = 0;           // 1 / 2: the previous step ended with an epilogue that issued exactly MF*NF / MF*NF/2 stores
    stage(0);
    for (int st = 0; st < total; ++st) {
        const int cur = st & 1;
        // the next tile's loads are OLDER than the epilogue's stores: a counted wait retires the loads only
        if (stores_in_flight == 1)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MF * NF) : "memory");
        else if (stores_in_flight == 2)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MF * NF / 2) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        stores_in_flight = 0;
        if (ln && ck == 1) {                            // the statistics loads of step 0 have landed (vmcnt(0) above)
            const float inv_c = 1.0f / (float)p.ln_c;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                float a = sp[i][0].x + sp[i][1].x, q = sp[i][0].y + sp[i][1].y;
                a += __shfl_xor(a, 16, 64);
                q += __shfl_xor(q, 16, 64);
                a += __shfl_xor(a, 32, 64);
                q += __shfl_xor(q, 32, 64);
                const float mean = a * inv_c;
                double var = (double)q * (double)inv_c - (double)mean * (double)mean;     // see idb_ln_row_table
                if (var < 0.0) var = 0.0;
                mr[i] = make_float2(mean, __builtin_amdgcn_rsqf((float)var + p.ln_eps));
            }
        }
        if (st + 1 < total) stage(cur ^ 1);
        if (ck == 0) {
            const int tm = ct / p.tiles_n, tn = ct - tm * p.tiles_n;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int nc = min(tn * BN + (wn * NF + j) * 16 + fg * 4, p.N - 4);
                cb[j] = *(const f32x4*)(bias_or_dummy + (has_cb ? nc : 0));   // raw: first use (and the wait) is in the epilogue
                cu[j] = *(const f32x4*)(ln ? p.ln_u + nc : bias_or_dummy);
            }
            if (ln) {
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const int m = min(tm * BM + (wm * MF + i) * 16 + fr, p.M - 1);
                    const float* ps = p.ln_stats + (long long)m * p.ln_nt * 2;
                    if (p.ln_nt <= 8) {
                        const float2 v0 = *(const float2*)(ps + 2 * min(fg, p.ln_nt - 1)), v1 = *(const float2*)(ps + 2 * min(fg + 4, p.ln_nt - 1));
                        sp[i][0] = fg < p.ln_nt ? v0 : make_float2(0.f, 0.f);
                        sp[i][1] = fg + 4 < p.ln_nt ? v1 : make_float2(0.f, 0.f);
                    } else {
                        float a = 0.f, q = 0.f;
                        for (int t = fg; t < p.ln_nt; t += 4) {
                            const float2 v = *(const float2*)(ps + 2 * t);
                            a += v.x;
                            q += v.y;
                        }
                        sp[i][0] = make_float2(a, q);
                        sp[i][1] = make_float2(0.f, 0.f);
                    }
                }
            }
        }
        const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
        const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ((ks * 4 + fg) ^ (fr & 7)) * 16;
            V8 af[MF], wf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const V8*)(sA + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *(const V8*)(sB + j * 16 * 128 + pos);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = Op<T>::mfma16(wf[j], af[i], acc[i][j]);
        }
        if (++ck == KT) {
            const int tm = ct / p.tiles_n, tn = ct - tm * p.tiles_n;
            if (p.geglu) {
                if constexpr ((NF & 1) == 0) {
                    idb_pl_epilogue<T, MF, NF, true>(p, acc, cb, cu, mr, tm * BM, tn * BN, wm, wn, fr, fg);
                    stores_in_flight = 2;
                }
            } else {
                idb_pl_epilogue<T, MF, NF, false>(p, acc, cb, cu, mr, tm * BM, tn * BN, wm, wn, fr, fg);
                stores_in_flight = 1;
            }
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ck = 0;
            ct += G;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct TileCfg { int mf, nf, wm; };   // tile = (16*mf*wm) x (32*nf), 128*wm threads
const TileCfg kTiles[] = {{0, 0, 2}, {4, 5, 2}, {4, 4, 2}, {2, 5, 2}, {1, 2, 4}, {4, 1, 2},
                          {1, 5, 4}, {1, 4, 4}, {2, 5, 4}, {2, 4, 4}};   // index = desc.tile % 10
// desc.tile = id + 10 * variant.  ids 4 and 6-9 have 8 waves per workgroup (2- or 3-stage ring): 4 = 64x64, 6 = 64x160,
// 7 = 64x128, 8 = 128x160, 9 = 128x128 — 6-9 are the tiles 3/-/1/2 with half the LDS-DMA instructions per wave and K-step;
// 4 exists to put more workgroups on the chip for the smallest batch-1 projections (M = 512, N = 1280: 160 instead of 80)
// (a 256x160 tile with 8 waves was measured 3-8 % slower than 128x160 with two workgroups per CU and is not kept).
constexpr int kNumTiles = 9;

struct Plan {
    int tile, splitk, tiles_m, tiles_n, ktiles, kt_per_split, M;
    long long K;
};

// idb_conv_patch_kernel's shapes: whole tiles of bm (64 / 128 / 256) output pixels that are whole image rows (or whole images), every
// K segment on the output grid; the halo patch of one tile fits the kernel's patch buffer (208 / 272 / 400 pixels)
static bool conv_patch_ok(const idb_gemm_desc* d, long long M, int bm) {
    if (d->stride != 1 || d->pad_mode == 1 || d->geglu || d->ln_stats || d->nsrc < 1 || d->src[0].taps != 9 || d->act) return false;
    const int W = d->out_w, H = d->out_h;
    if (!(W == 8 || W == 16 || W == 32 || W == 64) || W > bm || M % bm) return false;
    const long long HW = (long long)H * W;
    if (!(HW % bm == 0 || bm % HW == 0)) return false;
    const int nimg = HW >= bm ? 1 : (int)(bm / HW);
    if (nimg * (bm / W / nimg + 2) * (W + 2) > (bm == 64 ? 208 : bm == 128 ? 272 : 400)) return false;
    if (d->gn_in_partials) {
        // fused GroupNorm (transforming patch loaders): small tiles inside ONE sample, every K segment a normalised 3x3 source
        if (bm > 128 || HW < bm || d->gn_in_nsrc != d->nsrc || d->gn_in_groups <= 0 || d->gn_in_groups > 32 || d->gn_in_chunks < 1 || d->gn_in_chunks > 64) return false;
        long long cn = 0;
        for (int s = 0; s < d->nsrc; ++s) {
            if (d->src[s].taps != 9) return false;
            cn += d->src[s].channels;
        }
        if (cn % d->gn_in_groups) return false;
    }
    // 1x1 K segments (a fused conv_shortcut) run through the patch buffers with ONE K-step per chunk, i.e. on a two-deep pipeline:
    // auto plans take the patch kernel for pure 3x3 convs only unless IDB_CONV_PATCH_SHORTCUT=1 (forced tile ids run every mix)
    static const int env_sc = [] { const char* e = getenv("IDB_CONV_PATCH_SHORTCUT"); return e ? atoi(e) : 0; }();
    for (int s = 0; s < d->nsrc; ++s) {
        if (d->src[s].upsample || d->src[s].in_h != H || d->src[s].in_w != W || d->src[s].channels % 64) return false;
        if (d->src[s].taps != 9 && d->tile == 0 && !env_sc) return false;
    }
    return true;
}

int plan_gemm(const idb_gemm_desc* d, Plan* pl) {
    IDB_REQUIRE(d != nullptr, "idb_gemm: null descriptor");
    IDB_REQUIRE(idb_is_operand_dtype(d->dtype), "idb_gemm: dtype must be bf16 or f16 (got %d)", d->dtype);
    IDB_REQUIRE(d->batch > 0 && d->out_h > 0 && d->out_w > 0 && d->n > 0, "idb_gemm: non-positive dims");
    IDB_REQUIRE(d->stride == 1 || d->stride == 2, "idb_gemm: stride must be 1 or 2");
    IDB_REQUIRE(d->pad_mode == 0 || (d->pad_mode == 1 && d->stride == 2), "idb_gemm: pad_mode must be 0, or 1 with stride 2");
    if (d->gn_partials) {
        const long long hw = (long long)d->out_h * d->out_w;
        IDB_REQUIRE(d->gn_groups > 0 && d->n % d->gn_groups == 0 && d->n / d->gn_groups >= 2 && !d->geglu && d->out_dtype == d->dtype &&
                        hw % 64 == 0 && hw <= 4096 && d->out_ld == d->n && idb_aligned16(d->gn_partials),
                    "idb_gemm: gn_partials needs n %% gn_groups == 0, dense operand-dtype output, no GEGLU, out_h*out_w %% 64 == 0 and <= 4096");
    }
    IDB_REQUIRE(d->nsrc >= 1 && d->nsrc <= IDB_MAX_SRC, "idb_gemm: nsrc out of range");
    IDB_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == IDB_F32, "idb_gemm: out_dtype must be dtype or f32");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    IDB_REQUIRE(M < (1LL << 31), "idb_gemm: M too large");
    long long K = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        const idb_gemm_src& S = d->src[s];
        IDB_REQUIRE(S.ptr != nullptr && idb_aligned16(S.ptr), "idb_gemm: src[%d] pointer null or not 16-byte aligned", s);
        IDB_REQUIRE(S.channels > 0 && S.channels % 64 == 0, "idb_gemm: src[%d].channels=%d must be a multiple of 64", s, S.channels);
        IDB_REQUIRE(S.taps == 9 || S.taps == 1, "idb_gemm: src[%d].taps must be 9 or 1", s);
        IDB_REQUIRE(S.upsample == 0 || S.upsample == 1, "idb_gemm: src[%d].upsample must be 0/1", s);
        IDB_REQUIRE(S.in_h > 0 && S.in_w > 0, "idb_gemm: src[%d] spatial dims", s);
        const int lh = S.in_h << S.upsample, lw = S.in_w << S.upsample;
        if (S.taps == 9) {
            IDB_REQUIRE(d->out_h == (lh + d->stride - 1) / d->stride && d->out_w == (lw + d->stride - 1) / d->stride,
                        "idb_gemm: src[%d] %dx%d (up=%d, stride=%d) does not produce %dx%d", s, S.in_h, S.in_w,
                        S.upsample, d->stride, d->out_h, d->out_w);
        } else {
            IDB_REQUIRE(d->stride == 1 && lh == d->out_h && lw == d->out_w,
                        "idb_gemm: 1x1 src[%d] must match the output grid", s);
        }
        K += (long long)S.taps * S.channels;
        IDB_REQUIRE((long long)d->batch * S.in_h * S.in_w * S.channels * 2 < (1LL << 31),
                    "idb_gemm: src[%d] tensor is >= 2 GiB; split the batch", s);
    }
    IDB_REQUIRE(d->w && idb_aligned16(d->w) && d->out && idb_aligned16(d->out), "idb_gemm: w/out null or unaligned");
    IDB_REQUIRE(d->out_ld >= (d->geglu ? d->n / 2 : d->n), "idb_gemm: out_ld too small");
    if (d->geglu) {
        IDB_REQUIRE(d->n % 32 == 0 && d->out_dtype == d->dtype && !d->residual && !d->sample_bias,
                    "idb_gemm: GEGLU needs n %% 32 == 0, operand-dtype output, no residual/sample_bias");
        IDB_REQUIRE(d->out_ld % 4 == 0, "idb_gemm: GEGLU out_ld must be a multiple of 4");
    }
    if (d->n % 4 == 0) IDB_REQUIRE(d->out_ld % 4 == 0, "idb_gemm: out_ld must be a multiple of 4 when n is");
    IDB_REQUIRE(d->act == 0 || (d->act == 1 && !d->geglu && !d->residual), "idb_gemm: act must be 0, or 1 (GELU) without GEGLU/residual");
    if (d->sample_bias) IDB_REQUIRE(d->sample_bias_ld == 0 || d->sample_bias_ld >= d->n, "idb_gemm: sample_bias_ld must be 0 (broadcast) or >= n");

    IDB_REQUIRE(((long long)d->n + 15) / 16 * 16 * K * 2 < (1LL << 31), "idb_gemm: weight matrix is >= 2 GiB");
    IDB_REQUIRE(d->w_layout == 0 || d->w_layout == 1, "idb_gemm: w_layout must be 0 (rows) or 1 (idb_tile_weight)");
    pl->M = (int)M;
    pl->K = K;
    pl->ktiles = (int)(K / 64);
    int tile = d->tile % 10, ring3 = d->tile / 10;     // ring3: 0 -> 2-stage, 1 -> 3-stage, 2 -> 4-stage LDS ring
    // ring3: 0 -> 2-stage LDS-DMA ring, 1 -> 3-stage, 2 -> 4-stage, 3 -> register-staged double buffer, 4 -> persistent (plain matrices)
    // 5 / 6 / 7 -> loader-wave variant (idb_gemm_kernel_lw): 4 loader waves + 3-stage ring / 8 loader waves + 3 stages / 4 loader waves + 4 stages
    // 8 -> loader-wave variant with TWICE the rows (shapes 8 / 9 only: 256x160 / 256x128, 8 MFMA waves + 4 loader waves, 3-stage ring, one
    //      workgroup per CU with all 160 KB of LDS): 97 / 85 FLOP per byte moved L2 -> LDS instead of 73 for large grids
    const bool lw_tile = tile == 4 || (tile >= 6 && tile <= 9);
    IDB_REQUIRE(d->tile >= 0 && tile <= kNumTiles && ring3 <= 10 && !(ring3 && tile == 0) && !((ring3 == 1 || ring3 == 2) && tile == 5) &&
                    !(ring3 == 4 && tile > 2) && !(ring3 > 1 && ring3 < 5 && (tile >= 6 || tile == 4)) && !(ring3 >= 5 && !lw_tile) &&
                    !((ring3 == 8 || ring3 == 9) && tile < 8),
                "idb_gemm: tile id out of range");
    const bool plain = d->nsrc == 1 && d->src[0].taps == 1 && d->src[0].in_h == 1 && d->src[0].in_w == 1;
    const bool pl_ok = plain && d->split_k <= 1 && d->out_dtype == d->dtype && (d->geglu ? d->n / 2 : d->n) % 4 == 0 &&
                       d->out_ld % 4 == 0 && M * d->out_ld * 2 < (1LL << 31);
    IDB_REQUIRE(ring3 != 4 || pl_ok, "idb_gemm: the persistent variant needs one plain [M][K] source, operand-dtype output < 2 GiB, no split-K");
    int auto_sk = 0;                                    // split-K chosen together with the tile (0: by the rules below)
    if (tile == 0) {
        const bool n160 = (d->n % 160 == 0) && !d->geglu;
        const int bn = d->n <= 32 ? 32 : (n160 ? 160 : 128);
        const long long blocks_big = ((M + 127) / 128) * ((d->n + bn - 1) / bn);
        const long long blocks64 = ((M + 63) / 64) * ((d->n + 127) / 128);
        // measured on MI355X with HBM-cold operands inside a HIP graph (tools/bench_conv.py, tools/bench_small.py; warm
        // back-to-back timings rank the variants differently and mislead):
        //  * >= 2 workgroups per CU: 128-row 8-wave tiles, 2-deep ring (two co-resident workgroups hide each other's latency);
        //  * at most ONE workgroup per CU (the whole batch-1 UNet): the K-step of a lone workgroup is the issue time of its
        //    LDS-DMA instructions plus one memory round trip (tools/bench_latency.py: unchanged with the MFMAs removed), so
        //    8 waves (half the DMA instructions per wave) and a 3-deep ring (two K-steps in flight): -25...-45 % per launch;
        //  * long K (convs, the big FF projections): 128x160 ring-3 tile, split-K to one workgroup per CU (256 / blocks; 384
        //    blocks = 1.5 rounds is worse than 256 or 512), at least 8 K-steps per split; if that cannot fill half the chip
        //    (M = 128) the 64x128 ring-3 tile with the same rule;
        //  * short K: 64-row ring-3 tiles, split only for tiny grids (the reduce launch costs more than a short K loop).
        const bool short_k = pl->ktiles <= 10;
        const bool lin = d->nsrc == 1 && d->src[0].taps == 1;          // a plain linear / 1x1 conv
        bool keep7 = false;                                            // a measured 64x128 plan: not narrowed to 64x64 below
        static const int plan_ab_early = [] { const char* e = getenv("IDB_GEMM_PLAN_AB"); return e ? atoi(e) : 0; }();
        if (d->n <= 32) tile = 5;
        else if (d->geglu && blocks_big >= 256) tile = pl->ktiles >= 16 ? 2 : 9;   // N = 8C, no split-K: 128-row (persistent form for K >= 1024)
        else if (blocks_big >= (plan_ab_early & 2 ? 512 : 384)) tile = n160 ? 8 : 9;   // measured at batch 3 (384 workgroups): +6 % over the 64-row tiles; at 256 (batch 2): -1.4 %
        // (the 4-wave 128x160 tile reads 9 instead of 14 ds_read_b128 per 20 MFMAs and is 2-5 % faster on every long-K conv with
        //  M >= 32,768 in tools/bench_conv.py at B_eff 128 — end to end it changed nothing: batch 64 15.39 vs 15.32-15.35, batch 8 / 16
        //  -0.3 %; not adopted)
        else if (!d->geglu && pl->ktiles >= 32 && blocks_big <= (plan_ab_early & 4 ? 255 : 256)) {   // = 256 (batch 2, 64x64 level): 37.7 vs 49.8 us
            int sk = (int)(256 / blocks_big);
            const int cap = pl->ktiles / 8;
            if (sk > cap) sk = cap;
            if (sk > 32) sk = 32;
            if (sk < 1) sk = 1;
            if (M <= 128 && M > 64) {
                // ONE 128-row tile of rows (the 8x8 level at batch 1): two 64-row tiles with half the split reach 256 workgroups where the
                // 128-row plan stops at 192, and their slabs are half as many — conv 1280->1280 @8x8 21.7 -> 19.0 us, 2560->1280 26.2 -> 26.0
                // (tools/bench_conv.py 2); a 64-row tile also lies inside one 8x8 sample, so the GroupNorm in front of it can fuse
                tile = n160 ? 6 : 7;
                const long long b64 = 2 * ((d->n + 32 * kTiles[tile].nf - 1) / (32 * kTiles[tile].nf));
                sk = (int)(256 / b64);
                if (sk > cap) sk = cap;
                if (sk > 32) sk = 32;
                auto_sk = sk < 1 ? 1 : sk;
            } else if (blocks_big * sk >= 128) {
                tile = n160 ? 8 : 9;
                auto_sk = sk;
                if (n160 && sk == 2 && pl->ktiles < 64) { tile = 6; auto_sk = 1; }   // 64x160: the same 256 workgroups without a reduce launch
                else if (lin && sk <= 4 && pl->ktiles <= 48 && blocks64 >= 128 && blocks64 <= 256) {
                    // a short split of a plain linear: 64x128 tiles fill more than half the chip without one — FF-out at 32x32 (M = 2048, N = 640,
                    // K = 2560): 22.3 -> 16.7 us incl. the reduce launch it no longer needs (tools/bench_small.py)
                    tile = 7;
                    auto_sk = 1;
                    keep7 = true;
                }
            } else {
                tile = 7;
                sk = (int)(256 / blocks64);
                if (sk > cap) sk = cap;
                if (sk > 32) sk = 32;
                auto_sk = sk < 1 ? 1 : sk;
            }
            ring3 = 1;
        }
        else if (short_k && M >= 4096) { tile = (n160 && blocks_big < 256) ? 6 : 9; ring3 = tile == 6; }
        else if (lin && short_k && M < 4096 && ((M + 127) / 128) * ((d->n + 127) / 128) >= 128 && ((M + 127) / 128) * ((d->n + 127) / 128) <= 256) {
            // QKV at 32x32 (M = 2048, N = 1920, K = 640): 240 workgroups of 128x128 with loader waves instead of 480 of 64x128 two per CU:
            // 13.9 -> 12.5 us (tools/bench_small.py)
            tile = 9;
            ring3 = 1;
        }
        else if (lin && n160 && pl->ktiles >= 16 && M >= 4096 && ((M + 63) / 64) * (d->n / 160) <= 256) {
            // FF-out at 64x64 (M = 8192, N = 320, K = 1280): 256 workgroups of 64x160 with loader waves instead of 384 of 64x128 two per CU:
            // 17.9 -> 14.1 us (tools/bench_small.py)
            tile = 6;
            ring3 = 1;
        }
        else tile = (n160 && pl->ktiles >= 32) ? (M >= 4096 ? 6 : 3) : 7;
        if (tile == 7 && blocks64 <= 256) ring3 = 1;
        static const int env_ab = [] { const char* e = getenv("IDB_GEMM_PLAN_AB"); return e ? atoi(e) : 0; }();   // A/B switches for measurements
        if (tile == 7 && blocks64 <= 160 && !d->geglu && !(env_ab & 1) && !keep7) tile = 4;   // 64x64: twice the workgroups (m=512 n=1280 k=1280: 12.6 -> 9.9 us)
    }
    if (d->geglu) IDB_REQUIRE(kTiles[tile].nf % 2 == 0, "idb_gemm: GEGLU needs an even-NF tile");
    if (d->tile == 0 && tile != 5) {
        // weight-bound layers stream cold weights from HBM once per launch: a deeper ring keeps more loads in flight
        static const int env_ring = [] { const char* e = getenv("IDB_GEMM_RING_SMALL_M"); return e ? atoi(e) : 2; }();
        static const int env_m = [] { const char* e = getenv("IDB_GEMM_RING_M"); return e ? atoi(e) : 2048; }();
        if (M <= env_m && env_ring >= 3 && env_ring <= 4) ring3 = env_ring - 2;
    }
    if (d->tile == 0) {
        static const int env_rs = [] { const char* e = getenv("IDB_GEMM_REGSTAGE"); return e ? atoi(e) : 0; }();
        if (env_rs) ring3 = 3;
        static const int env_pl = [] { const char* e = getenv("IDB_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
        const long long tiles = ((M + 127) / 128) * ((d->n + (32 * kTiles[tile].nf) - 1) / (32 * kTiles[tile].nf));
        // measured: +12-15 % on the GEGLU projections (N = 8C, K = C), neutral or slightly negative on the other K = C layers
        if (env_pl && pl_ok && d->geglu && tile == 2 && pl->ktiles <= 24 && tiles >= 256 && d->split_k <= 1) ring3 = 4;
        // with the bias vectors loaded at the tile's first K-step (round 2) the persistent form also wins the SHORT-K GEGLU
        // projections once the grid is many rounds deep: K = 320: 593 vs 529 TFLOP/s, K = 640: 795 vs 637 (B_eff = 128,
        // tools/bench_proj.py).  It cannot fold the LayerNorm (ln_stats -> IDB_EUNSUPPORTED, the caller keeps idb_layernorm); end
        // to end with that launch back: batch 64 14.76 -> 15.12, batch 8 13.63 -> 13.87, batch 1 6.44 -> 6.52 images/s
        static const int env_plg = [] { const char* e = getenv("IDB_GEMM_PL_GEGLU_TILES"); return e ? atoi(e) : 512; }();
        if (env_pl && pl_ok && d->geglu && tile == 9 && env_plg > 0 && tiles >= env_plg && d->split_k <= 1) { tile = 2; ring3 = 4; }
    }
    if (d->tile == 0 && ring3 == 1 && (tile == 4 || (tile >= 6 && tile <= 9))) {
        // one workgroup per CU (ring-3 plans): loader waves take the DMA issue and the waits off the MFMA waves (idb_gemm_kernel_lw)
        // measured (tools/bench_lw.py, B_eff 2, cold weights in a HIP graph): conv 320->320 @64x64 31.3 -> 23.7 us, the other convs of the
        // 64x64 / 32x32 levels -5...-10 %, weight-streaming 16x16 / 8x8 layers and short-K linears 0...-3 %; 4 loader waves with a
        // 4-stage ring (7) >= 4 with 3 stages (5) >= 8 with 3 stages (6); end to end batch 1 +1.1 %.  IDB_GEMM_LW=0: ring-3 kernels
        static const int env_lw = [] { const char* e = getenv("IDB_GEMM_LW"); return e ? atoi(e) : 7; }();
        if (env_lw >= 5 && env_lw <= 7) ring3 = env_lw;
    }
    if (d->tile == 0 && ring3 == 0 && (tile == 8 || tile == 9) && !d->geglu && pl->ktiles >= 20) {
        // large grids, K >= 1280: 256-row loader-wave tiles once the 256-row grid is still >= IDB_GEMM_BIG_TILES workgroups (two rounds of
        // the chip; 0 = off).  Measured at B_eff 128 (tools/bench_conv.py / bench_proj.py): every 3x3 conv +8...+19 % (1.01-1.17 ->
        // 1.08-1.32 PFLOP/s), K >= 1280 projections +3...+15 %, K = 640 equal, K = 320 -19 % (excluded); batch 64 end to end +4.1 %
        static const int env_big = [] { const char* e = getenv("IDB_GEMM_BIG_TILES"); return e ? atoi(e) : 512; }();
        const long long blocks256 = ((M + 255) / 256) * ((d->n + 32 * kTiles[tile].nf - 1) / (32 * kTiles[tile].nf));
        if (env_big > 0 && blocks256 >= env_big) ring3 = 8;
        // 3x3 stride-1 convs on that plan: the patch-resident form (idb_conv_patch_kernel).  IDB_CONV_PATCH=0: tap-major 256-row tiles
        static const int env_patch = [] { const char* e = getenv("IDB_CONV_PATCH"); return e ? atoi(e) : 1; }();
        if (ring3 == 8 && env_patch && d->split_k <= 1 && conv_patch_ok(d, M, 256)) ring3 = 9;
    }
    if (d->tile == 0 && ring3 >= 5 && ring3 <= 7 && !d->geglu) {
        // the same for the one-workgroup-per-CU plans (64- / 128-row tiles, split-K by whole chunks): IDB_CONV_PATCH_SMALL=1
        static const int env_ps = [] { const char* e = getenv("IDB_CONV_PATCH_SMALL"); return e ? atoi(e) : 0; }();
        if ((env_ps || d->gn_in_partials) && conv_patch_ok(d, M, 16 * kTiles[tile].mf * kTiles[tile].wm)) ring3 = 10;
    }
    if ((ring3 == 9 && !(d->split_k <= 1 && conv_patch_ok(d, M, 256))) || (ring3 == 10 && !conv_patch_ok(d, M, 16 * kTiles[tile].mf * kTiles[tile].wm))) {
        idb_set_error("idb_gemm: tile %d (patch-resident conv) needs a 3x3 stride-1 pad-1 first source, 1x1 / 3x3 sources on the output grid without "
                      "upsampling, out_w in {8,16,32,64}, whole tiles of 256 pixels, no folded LayerNorm / fused GroupNorm / GEGLU / split-K", d->tile);
        return IDB_EUNSUPPORTED;
    }
    pl->tile = tile + 10 * ring3;
    const int bm = 16 * kTiles[tile].mf * kTiles[tile].wm * (ring3 == 8 || ring3 == 9 ? 2 : 1), bn = 32 * kTiles[tile].nf;
    pl->tiles_m = (int)((M + bm - 1) / bm);
    pl->tiles_n = (d->n + bn - 1) / bn;
    const long long blocks = (long long)pl->tiles_m * pl->tiles_n;
    IDB_REQUIRE(blocks < (1LL << 31), "idb_gemm: grid too large");
    int sk = d->split_k;
    if (sk <= 0) {
        sk = 1;
        const bool small_tile = kTiles[tile].mf * kTiles[tile].wm <= 4 && ring3 != 8 && ring3 != 9;     // 64-row tiles
        if (auto_sk) {
            sk = auto_sk;
        } else if (!d->geglu && small_tile && blocks < 96 && pl->ktiles >= 10) {
            // measured (tools/bench_small.py): with the 3-deep ring a short K loop is cheaper than a split + reduce launch
            // unless the grid is tiny (M = 128)
            sk = (int)((160 + blocks - 1) / blocks);
            const int max_by_k = pl->ktiles / (blocks >= 64 ? 16 : 5);
            if (sk > max_by_k) sk = max_by_k;
            if (sk > 32) sk = 32;
            if (sk < 1) sk = 1;
        } else if (!d->geglu && !small_tile && blocks < 192 && pl->ktiles >= 10) {
            sk = (int)((384 + blocks - 1) / blocks);
            const int max_by_k = pl->ktiles / 8;
            if (sk > max_by_k) sk = max_by_k;
            if (sk > 32) sk = 32;
            if (sk < 1) sk = 1;
        }
    }
    if (pl->tile / 10 == 4 || pl->tile / 10 == 9) sk = 1;
    IDB_REQUIRE(!(d->geglu && sk > 1), "idb_gemm: GEGLU does not support split-K");
    if (d->act) sk = 1;
    if (sk > pl->ktiles) sk = pl->ktiles;
    if (sk < 1) sk = 1;
    if (d->split_k <= 0 && !(d->flags & 16) && (pl->tile / 10 <= 1 || pl->tile / 10 >= 5) && kTiles[tile].wm == 4) {
        // one K-slice per XCD (kernel remap modes 1/2) needs S % 8 == 0 or S == 4: round the heuristic's choice
        static const int env_xcd = [] { const char* e = getenv("IDB_GEMM_XCD_SLICES"); return e ? atoi(e) : 1; }();
        if (env_xcd) {
            if (sk > 8) {
                const int up = ((sk + 7) / 8) * 8;
                sk = (up <= 32 && up * 6 <= pl->ktiles) ? up : (sk / 8) * 8;
            } else if (sk >= 5) {
                sk = (8 * 6 <= pl->ktiles) ? 8 : 4;
            }
        }
    }
    pl->splitk = sk;
    pl->kt_per_split = (pl->ktiles + sk - 1) / sk;
    return IDB_OK;
}

template <typename T, int MF, int NF, int NS, int WM = 2>
int launch_tile(const GemmParams& p, const Plan& pl, hipStream_t st) {
    constexpr int RS = 16 * WM, NJ = (32 * NF + RS - 1) / RS;
    constexpr int LDS = (16 * MF * WM + NJ * RS) * 128 * NS;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm_kernel<T, MF, NF, NS, WM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.splitk);
    hipLaunchKernelGGL((idb_gemm_kernel<T, MF, NF, NS, WM>), grid, dim3(128 * WM), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm");
    return IDB_OK;
}

template <typename T, int MF, int NF, int NS, int WM, int LW>
int launch_tile_lw(const GemmParams& p, const Plan& pl, hipStream_t st) {
    constexpr int LR = 8 * LW, NJ = (32 * NF + LR - 1) / LR;
    constexpr int LDS = (16 * MF * WM + NJ * LR) * 128 * NS;
    static_assert(LDS <= 160 * 1024, "LDS ring does not fit");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm_kernel_lw<T, MF, NF, NS, WM, LW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.splitk);
    hipLaunchKernelGGL((idb_gemm_kernel_lw<T, MF, NF, NS, WM, LW>), grid, dim3(128 * WM + 64 * LW), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm(lw)");
    return IDB_OK;
}

template <typename T, int MF, int NF, int NS, bool GN = false>
int launch_conv_patch(const GemmParams& p, const Plan& pl, hipStream_t st) {
    constexpr int LDS = 2 * (MF == 1 ? (GN ? 224 : 208) : MF == 2 ? (GN ? 288 : 272) : 400) * 128 + NS * 32 * NF * 128 + (GN ? 256 : 0);
    static_assert(LDS <= 160 * 1024, "patch buffers + weight ring do not fit");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_conv_patch_kernel<T, MF, NF, NS, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((idb_conv_patch_kernel<T, MF, NF, NS, GN>), dim3(pl.tiles_m * pl.tiles_n, 1, pl.splitk), dim3(GN ? 896 : 768), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm(patch)");
    return IDB_OK;
}

// fused GroupNorm: LDS = ring + {k, h} table + group statistics; 0 if the shape cannot run (caller: IDB_EUNSUPPORTED)
static size_t gn_fused_lds(int bm, int stage_rows, int ns, const GemmParams& p) {
    const int nsamp = bm > p.HW ? bm / p.HW : 1;
    return (size_t)stage_rows * 128 * ns + (size_t)nsamp * p.gn_in_c * 8 + (size_t)nsamp * p.gn_in_groups * 8;
}

template <typename T, int MF, int NF, int NS, int WM>
int launch_tile_gn(const GemmParams& p, const Plan& pl, hipStream_t st) {
    constexpr int NV = 8, LR = 32, NJ = (32 * NF + LR - 1) / LR, BM = 16 * MF * WM;
    const size_t lds = gn_fused_lds(BM, BM + NJ * LR, NS, p);
    if (lds > 160 * 1024) {
        idb_set_error("idb_gemm: fused GroupNorm needs %zu bytes of LDS for this tile", lds);
        return IDB_EUNSUPPORTED;
    }
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm_kernel_gn<T, MF, NF, NS, WM, NV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_lds = 160 * 1024;
    }
    dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.splitk);
    hipLaunchKernelGGL((idb_gemm_kernel_gn<T, MF, NF, NS, WM, NV>), grid, dim3(128 * WM + 256 + 64 * NV), lds, st, p);
    IDB_CHECK_LAUNCH("idb_gemm(gn)");
    return IDB_OK;
}

template <typename T>
int launch_gn_by_tile(const GemmParams& p, const Plan& pl, hipStream_t st) {
    // 4 MFMA waves (2 x 2, one per SIMD: the wave tile is twice that of the 8-wave kernels of the same workgroup tile) + 4 loader waves +
    // 8 normalizer waves = 16 waves: with 4 normalizer waves the transform (9x redundant for a 3x3 conv: every tap re-reads its pixels)
    // was the K-step's critical path — 50 vs 34 us on conv 320->320 @64x64 at B_eff 2.  64-row tiles: 4-stage ring; 128-row: 3 (LDS)
    switch (pl.tile % 10) {
        case 4: return launch_tile_gn<T, 2, 2, 4, 2>(p, pl, st);
        case 6: return launch_tile_gn<T, 2, 5, 4, 2>(p, pl, st);
        case 7: return launch_tile_gn<T, 2, 4, 4, 2>(p, pl, st);
        default:       // 128-row tiles: a 4-wave MFMA role needs > 128 VGPRs there (16 waves per workgroup): not instantiated, gemm_fuses_gn says no
            idb_set_error("idb_gemm: fused GroupNorm is built for 64-row tiles only");
            return IDB_EUNSUPPORTED;
    }
}

template <typename T, int NS, int LW>
int launch_lw_by_tile(const GemmParams& p, const Plan& pl, hipStream_t st) {
    switch (pl.tile % 10) {
        case 4: return launch_tile_lw<T, 1, 2, NS, 4, LW>(p, pl, st);
        case 6: return launch_tile_lw<T, 1, 5, NS, 4, LW>(p, pl, st);
        case 7: return launch_tile_lw<T, 1, 4, NS, 4, LW>(p, pl, st);
        case 8: return launch_tile_lw<T, 2, 5, NS, 4, LW>(p, pl, st);
        default: return launch_tile_lw<T, 2, 4, NS, 4, LW>(p, pl, st);
    }
}

template <typename T, int MF, int NF>
int launch_tile_rs(const GemmParams& p, const Plan& pl, hipStream_t st) {
    constexpr int LDS = (32 * MF + 32 * NF) * 128 * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm_kernel_rs<T, MF, NF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.splitk);
    hipLaunchKernelGGL((idb_gemm_kernel_rs<T, MF, NF>), grid, dim3(256), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm(rs)");
    return IDB_OK;
}

template <typename T, int MF, int NF>
int launch_tile_pl(const GemmParams& p, const Plan& pl, hipStream_t st) {
    constexpr int LDS = (32 * MF + 32 * NF) * 128 * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm_kernel_pl<T, MF, NF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    const int tiles = pl.tiles_m * pl.tiles_n;
    hipLaunchKernelGGL((idb_gemm_kernel_pl<T, MF, NF>), dim3(tiles < 512 ? tiles : 512), dim3(256), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm(pl)");
    return IDB_OK;
}

template <typename T>
int launch_all(const idb_gemm_desc* d, const GemmParams& p, const Plan& pl, hipStream_t st) {
    int rc;
    if (p.gn_in_part) {
        if (pl.tile / 10 == 10) {                              // the patch-resident conv with transforming patch loaders
            switch (pl.tile % 10) {
                case 4: rc = launch_conv_patch<T, 1, 2, 4, true>(p, pl, st); break;
                case 6: rc = launch_conv_patch<T, 1, 5, 4, true>(p, pl, st); break;
                case 7: rc = launch_conv_patch<T, 1, 4, 4, true>(p, pl, st); break;
                case 8: rc = launch_conv_patch<T, 2, 5, 4, true>(p, pl, st); break;
                default: rc = launch_conv_patch<T, 2, 4, 4, true>(p, pl, st); break;
            }
        } else
            rc = launch_gn_by_tile<T>(p, pl, st);
        if (rc != IDB_OK || (d->flags & 1)) return rc;
        return idb_finish_splitk<T>(p, pl.M, d->n, d->batch, pl.splitk, d->gn_partials, d->gn_groups, d->dtype, st);
    }
    if (pl.tile / 10 >= 5) {
        if (pl.tile / 10 == 10) {
            switch (pl.tile % 10) {
                case 4: rc = launch_conv_patch<T, 1, 2, 4>(p, pl, st); break;
                case 6: rc = launch_conv_patch<T, 1, 5, 4>(p, pl, st); break;
                case 7: rc = launch_conv_patch<T, 1, 4, 4>(p, pl, st); break;
                case 8: rc = launch_conv_patch<T, 2, 5, 4>(p, pl, st); break;
                default: rc = launch_conv_patch<T, 2, 4, 4>(p, pl, st); break;
            }
        } else if (pl.tile / 10 == 9)
            rc = pl.tile % 10 == 8 ? launch_conv_patch<T, 4, 5, 3>(p, pl, st) : launch_conv_patch<T, 4, 4, 3>(p, pl, st);
        else if (pl.tile / 10 == 8)
            rc = pl.tile % 10 == 8 ? launch_tile_lw<T, 4, 5, 3, 4, 4>(p, pl, st) : launch_tile_lw<T, 4, 4, 3, 4, 4>(p, pl, st);
        else
            rc = pl.tile / 10 == 5 ? launch_lw_by_tile<T, 3, 4>(p, pl, st) : pl.tile / 10 == 6 ? launch_lw_by_tile<T, 3, 8>(p, pl, st)
                                                                                               : launch_lw_by_tile<T, 4, 4>(p, pl, st);
        if (rc != IDB_OK || (d->flags & 1)) return rc;
        return idb_finish_splitk<T>(p, pl.M, d->n, d->batch, pl.splitk, d->gn_partials, d->gn_groups, d->dtype, st);
    }
    switch (pl.tile) {
        case 1: rc = launch_tile<T, 4, 5, 2>(p, pl, st); break;
        case 2: rc = launch_tile<T, 4, 4, 2>(p, pl, st); break;
        case 3: rc = launch_tile<T, 2, 5, 2>(p, pl, st); break;
        case 4: rc = launch_tile<T, 1, 2, 2, 4>(p, pl, st); break;
        case 11: rc = launch_tile<T, 4, 5, 3>(p, pl, st); break;
        case 12: rc = launch_tile<T, 4, 4, 3>(p, pl, st); break;
        case 13: rc = launch_tile<T, 2, 5, 3>(p, pl, st); break;
        case 14: rc = launch_tile<T, 1, 2, 3, 4>(p, pl, st); break;
        case 6: rc = launch_tile<T, 1, 5, 2, 4>(p, pl, st); break;
        case 7: rc = launch_tile<T, 1, 4, 2, 4>(p, pl, st); break;
        case 8: rc = launch_tile<T, 2, 5, 2, 4>(p, pl, st); break;
        case 16: rc = launch_tile<T, 1, 5, 3, 4>(p, pl, st); break;
        case 18: rc = launch_tile<T, 2, 5, 3, 4>(p, pl, st); break;
        case 17: rc = launch_tile<T, 1, 4, 3, 4>(p, pl, st); break;
        case 19: rc = launch_tile<T, 2, 4, 3, 4>(p, pl, st); break;
        case 9: rc = launch_tile<T, 2, 4, 2, 4>(p, pl, st); break;
        case 41: rc = launch_tile_pl<T, 4, 5>(p, pl, st); break;
        case 42: rc = launch_tile_pl<T, 4, 4>(p, pl, st); break;
        case 31: rc = launch_tile_rs<T, 4, 5>(p, pl, st); break;
        case 32: rc = launch_tile_rs<T, 4, 4>(p, pl, st); break;
        case 33: rc = launch_tile_rs<T, 2, 5>(p, pl, st); break;
        case 35: rc = launch_tile_rs<T, 4, 1>(p, pl, st); break;
        case 21: rc = launch_tile<T, 4, 5, 4>(p, pl, st); break;
        case 22: rc = launch_tile<T, 4, 4, 4>(p, pl, st); break;
        case 23: rc = launch_tile<T, 2, 5, 4>(p, pl, st); break;
        default: rc = launch_tile<T, 4, 1, 2>(p, pl, st); break;
    }
    if (rc != IDB_OK || (d->flags & 1)) return rc;
    return idb_finish_splitk<T>(p, pl.M, d->n, d->batch, pl.splitk, d->gn_partials, d->gn_groups, d->dtype, st);
}

}  // namespace

extern "C" size_t idb_gemm_workspace_bytes(const idb_gemm_desc* d) {
    Plan pl;
    if (plan_gemm(d, &pl) != IDB_OK) return 0;
    return pl.splitk > 1 ? (size_t)pl.splitk * pl.M * d->n * sizeof(float) : 0;
}

// the LDS-staged coalesced epilogue runs for this (descriptor, plan): the only epilogue that emits row statistics / folds a LayerNorm
static bool gemm_uses_lds_epilogue(const idb_gemm_desc* d, const Plan& pl) {
    const int no = d->geglu ? d->n / 2 : d->n;
    return pl.tile / 10 != 4 && pl.tile / 10 != 3 && d->out_dtype == d->dtype && pl.splitk == 1 && no % 8 == 0 && d->out_ld % 8 == 0 && !(d->flags & 4) &&
           (!d->residual || idb_aligned16(d->residual));
}

// a folded LayerNorm needs the LDS-staged epilogue, or the persistent variant with >= 2 K-steps (statistics loaded at the first,
// reduced at the second)
static bool gemm_folds_ln(const idb_gemm_desc* d, const Plan& pl) {
    if (d->nsrc != 1 || d->src[0].taps != 1 || d->n % 4 || d->bias || d->sample_bias) return false;
    // persistent variant: built and tested, but off unless asked for (flags bit 8 / IDB_GEMM_PL_LN=1) — measured on one box, the
    // producer's statistics pass + the fold cost what the idb_layernorm launch costs: batch 64 14.77 -> 14.65 images/s (twice),
    // batch 1 6.645 vs 6.641 with 15 fewer launches
    static const int env_pl_ln = [] { const char* e = getenv("IDB_GEMM_PL_LN"); return e ? atoi(e) : 0; }();
    if (pl.tile / 10 == 4) return (env_pl_ln || (d->flags & 256)) && pl.ktiles >= 2;
    return gemm_uses_lds_epilogue(d, pl);
}

// the GEMM's own LDS-staged epilogue can emit the first GroupNorm pass of the output (no split-K, whole groups per column tile)
static bool gemm_epilogue_emits_gn(const idb_gemm_desc* d, const Plan& pl, int groups) {
    if ((pl.tile / 10 > 2 && pl.tile / 10 < 5) || d->geglu || !gemm_uses_lds_epilogue(d, pl)) return false;
    const TileCfg& t = kTiles[pl.tile % 10];
    const int epi_threads = d->gn_in_partials ? 256 : 128 * t.wm;      // the fused-GroupNorm kernel has 4 MFMA waves
    return idb_epilogue_emits_gn(16 * t.mf * t.wm * (pl.tile / 10 == 8 || pl.tile / 10 == 9 ? 2 : 1), 32 * t.nf, epi_threads, pl.M, d->n, groups) && (long long)d->out_h * d->out_w % 64 == 0 &&
           d->out_ld == d->n;
}

extern "C" int32_t idb_gemm_emits_gn_partials(const idb_gemm_desc* d, int32_t groups) {
    Plan pl;
    if (plan_gemm(d, &pl) != IDB_OK) return 0;
    if (pl.splitk > 1) return d->out_dtype == d->dtype && pl.M % 64 == 0 && gn_reduce_slice(d->n, groups) != 0 ? 1 : 0;
    return gemm_epilogue_emits_gn(d, pl, groups) ? 2 : 0;
}

// the fused GroupNorm runs on the one-workgroup-per-CU loader-wave plans only (variants 5-7), stride 1, every normalised source on
// the output grid, tiles inside one sample or covering whole samples, no folded LayerNorm on the same launch
static bool gemm_fuses_gn(const idb_gemm_desc* d, const Plan& pl) {
    if (!d->gn_in_partials) return false;
    if (pl.tile / 10 == 10) return conv_patch_ok(d, pl.M, 16 * kTiles[pl.tile % 10].mf * kTiles[pl.tile % 10].wm);
    // the tap-major normalizer-wave kernel loses on every 3x3 conv (it re-normalises each pixel per tap): auto plans use it for 1x1 sources
    // only (Transformer2DModel norm + proj_in); forced tile ids and IDB_GN_TAPMAJOR=1 still reach it for 3x3 sources
    static const int env_tm = [] { const char* e = getenv("IDB_GN_TAPMAJOR"); return e ? atoi(e) : 0; }();
    if (d->src[0].taps == 9 && d->tile == 0 && !env_tm) return false;
    if (pl.tile / 10 < 5 || pl.tile / 10 > 7 || d->stride != 1 || d->ln_stats || d->geglu || d->gn_in_nsrc < 1 || d->gn_in_nsrc > d->nsrc) return false;
    const int bm = 16 * kTiles[pl.tile % 10].mf * kTiles[pl.tile % 10].wm;
    if (bm != 64) return false;                        // 64x160 / 64x128 / 64x64 plans (launch_gn_by_tile)
    const long long hw = (long long)d->out_h * d->out_w;
    if (!(hw % bm == 0 || (bm % hw == 0 && bm / hw <= 2))) return false;
    long long cn = 0;
    for (int s = 0; s < d->gn_in_nsrc; ++s) {
        if (d->src[s].upsample || d->src[s].in_h != d->out_h || d->src[s].in_w != d->out_w) return false;
        cn += d->src[s].channels;
    }
    if (d->gn_in_groups <= 0 || cn % d->gn_in_groups || cn / d->gn_in_groups < 1 || d->gn_in_chunks < 1 || d->gn_in_chunks > 64) return false;
    const int nsamp = bm > hw ? (int)(bm / hw) : 1;
    const int stage_rows = bm + (32 * kTiles[pl.tile % 10].nf + 31) / 32 * 32;
    const size_t lds = (size_t)stage_rows * 128 * (bm == 64 ? 4 : 3) + (size_t)nsamp * cn * 8 + (size_t)nsamp * d->gn_in_groups * 8;
    return lds <= 160 * 1024;
}

extern "C" int32_t idb_gemm_fuses_groupnorm(const idb_gemm_desc* d) {
    Plan pl;
    return plan_gemm(d, &pl) == IDB_OK && gemm_fuses_gn(d, pl) ? 1 : 0;
}

extern "C" int32_t idb_gemm_folds_layernorm(const idb_gemm_desc* d) {
    Plan pl;
    return plan_gemm(d, &pl) == IDB_OK && gemm_folds_ln(d, pl) ? 1 : 0;
}

extern "C" int32_t idb_gemm_row_stats_tiles(const idb_gemm_desc* d) {
    Plan pl;
    if (plan_gemm(d, &pl) != IDB_OK || !gemm_uses_lds_epilogue(d, pl)) return 0;
    return pl.tiles_n;
}

extern "C" int idb_gemm_plan(const idb_gemm_desc* d, int32_t* tile, int32_t* split_k, int32_t* blocks) {
    Plan pl;
    int rc = plan_gemm(d, &pl);
    if (rc != IDB_OK) return rc;
    if (tile) *tile = pl.tile;
    if (split_k) *split_k = pl.splitk;
    if (blocks) *blocks = pl.tiles_m * pl.tiles_n * pl.splitk;
    return IDB_OK;
}

extern "C" int idb_gemm(const idb_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    Plan pl;
    int rc = plan_gemm(d, &pl);
    if (rc != IDB_OK) return rc;
    const size_t need = pl.splitk > 1 ? (size_t)pl.splitk * pl.M * d->n * sizeof(float) : 0;
    IDB_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && idb_aligned16(workspace)),
                "idb_gemm: workspace too small (%zu < %zu) or unaligned", workspace_bytes, need);
    GemmParams p = {};
    for (int s = 0; s < IDB_MAX_SRC; ++s) {
        const idb_gemm_src& S = d->src[s < d->nsrc ? s : d->nsrc - 1];
        p.src[s] = GemmSrcK{(const char*)S.ptr, (unsigned)((long long)d->batch * S.in_h * S.in_w * S.channels * 2), S.channels,
                            S.taps, S.in_h, S.in_w, S.upsample};
    }
    p.M = pl.M;
    p.N = d->n;
    p.HW = d->out_h * d->out_w;
    p.OW = d->out_w;
    p.stride = d->stride;
    p.pad = d->pad_mode == 1 ? 0 : 1;
    if (d->w_layout == 1) {          // K-tiled 16-row blocks (idb_tile_weight): [ceil(n/16)][K/64][16 rows][64 k]
        p.w_row_bytes = 128u;
        p.w_kstep = 2048u;
        p.w_blk_bytes = (unsigned)pl.ktiles * 2048u;
        p.w_bytes = (unsigned)(((long long)d->n + 15) / 16 * pl.ktiles * 2048);
    } else {
        p.w_row_bytes = (unsigned)(pl.K * 2);
        p.w_kstep = 128u;
        p.w_blk_bytes = (unsigned)(pl.K * 32);
        p.w_bytes = (unsigned)((long long)d->n * pl.K * 2);
    }
    p.ktiles = pl.ktiles;
    p.kt_per_split = pl.kt_per_split;
    {
        // K-slice-per-XCD remap: the 8-wave ring kernels only (the register-staged and persistent variants keep mode 0)
        static const int env_xcd = [] { const char* e = getenv("IDB_GEMM_XCD_SLICES"); return e ? atoi(e) : 1; }();
        const long long X = (long long)pl.tiles_m * pl.tiles_n;
        p.xcd_mode = 0;
        if (env_xcd && (pl.tile / 10 <= 1 || pl.tile / 10 >= 5) && pl.tile % 10 != 5 && pl.tile % 10 != 0) {
            if (pl.splitk >= 8 && pl.splitk % 8 == 0) p.xcd_mode = 1;
            else if (pl.splitk == 4 && X % 2 == 0) p.xcd_mode = 2;
        }
    }
    p.splitk = pl.splitk;
    p.w = (const char*)d->w;
    p.bias = d->bias;
    p.sbias = d->sample_bias;
    p.sbias_ld = d->sample_bias_ld;
    p.res = (const char*)d->residual;
    p.out = d->out;
    p.out_ld = d->out_ld;
    p.out_f32 = d->out_dtype == IDB_F32;
    p.geglu = d->geglu;
    p.scale = d->out_scale == 0.f ? 1.f : d->out_scale;
    p.partial = (float*)workspace;
    p.tiles_n = pl.tiles_n;
#ifndef IDB_PROFILING
    IDB_REQUIRE(!(d->flags & (2 | 32 | 64 | 128)), "idb_gemm: flags bits 1/5/6/7 are profiling switches of libidb_kernels_prof.so (make prof)");
#endif
    p.dbg_skip_store = (d->flags & 2) ? 1 : ((d->flags & 32) ? 2 : 0);
    p.out_bytes = (unsigned)((long long)pl.M * d->out_ld * 2);
    p.act = d->act;
    p.dbg_loop = (d->flags & 64) ? 1 : ((d->flags & 128) ? 2 : 0);
    {
        const int no = d->geglu ? d->n / 2 : d->n;
        // In-kernel split-K reduce (flags bit 4 only).  Measured on MI355X it LOSES to the separate reduce launch in the
        // sampling loop (batch 1: 4.49 vs 5.11 images/s): every split workgroup pays an agent-scope release (buffer_wbl2
        // of its XCD's L2) and one workgroup per tile re-reads all slabs — the "splitk-seam" price of the CDNA guide.
        // The path is kept, tested bit-identical to the two-launch form, for shapes where a launch boundary is dearer.
        static const int env_fused = [] { const char* e = getenv("IDB_GEMM_FUSED_REDUCE"); return e ? atoi(e) : 0; }();
        const bool want_fused = (d->flags & 16) || (env_fused > 0 && pl.splitk <= env_fused);
        const bool fused_reduce = pl.splitk > 1 && d->counters && want_fused && !(d->flags & 8) && !(d->flags & 1) &&
                                  (long long)pl.tiles_m * pl.tiles_n <= d->counters_len;
        p.counters = fused_reduce ? d->counters : nullptr;
        p.lds_epi = (pl.tile / 10 != 4 && !p.out_f32 && (pl.splitk == 1 || fused_reduce) && no % 8 == 0 && d->out_ld % 8 == 0 && !(d->flags & 4) &&
                     (!d->residual || idb_aligned16(d->residual))) ? 1 : 0;
    }
    if (d->row_stats_out || d->ln_stats) {
        if ((d->row_stats_out && (!p.lds_epi || pl.tile / 10 == 3)) || (d->ln_stats && !gemm_folds_ln(d, pl))) {
            idb_set_error("idb_gemm: row_stats_out needs a plan with the LDS-staged epilogue (no split-K, no persistent / register-staged variant); "
                          "ln_stats that or the persistent variant");
            return IDB_EUNSUPPORTED;
        }
        IDB_REQUIRE(!d->ln_stats || (d->ln_tiles > 0 && d->ln_u && d->ln_v && idb_aligned16(d->ln_u) && idb_aligned16(d->ln_v) && ((uintptr_t)d->ln_stats & 7) == 0 &&
                                     d->nsrc == 1 && d->src[0].taps == 1 && d->n % 4 == 0 && !d->bias && !d->sample_bias),
                    "idb_gemm: ln_stats needs ln_tiles > 0, aligned ln_u / ln_v, one 1x1 source, n %% 4 == 0, and no bias / sample_bias "
                    "(add the layer's bias into ln_v)");
        IDB_REQUIRE(!d->row_stats_out || (((uintptr_t)d->row_stats_out & 7) == 0 && !d->geglu), "idb_gemm: row_stats_out must be 8-byte aligned, no GEGLU");
    }
    if (d->gn_partials && pl.splitk == 1 && !(d->flags & 1) && gemm_epilogue_emits_gn(d, pl, d->gn_groups)) {
        p.gn_part = d->gn_partials;          // launch_all's idb_finish_splitk then has nothing left to launch
        p.gn_groups = d->gn_groups;
    }
    if (d->gn_in_partials) {
        if (!gemm_fuses_gn(d, pl)) {
            idb_set_error("idb_gemm: this plan cannot fuse the GroupNorm (needs a one-workgroup-per-CU loader-wave plan, stride 1, sources on the "
                          "output grid, tile inside one sample): run idb_groupnorm + idb_gemm");
            return IDB_EUNSUPPORTED;
        }
        IDB_REQUIRE(d->gn_in_gamma && d->gn_in_beta && d->gn_in_eps > 0.f && idb_aligned16(d->gn_in_partials), "idb_gemm: gn_in_gamma / gn_in_beta / gn_in_eps invalid");
        p.gn_in_part = d->gn_in_partials;
        p.gn_in_gamma = d->gn_in_gamma;
        p.gn_in_beta = d->gn_in_beta;
        p.gn_in_chunks = d->gn_in_chunks;
        p.gn_in_groups = d->gn_in_groups;
        p.gn_in_nsrc = d->gn_in_nsrc;
        p.gn_in_silu = d->gn_in_silu;
        p.gn_in_eps = d->gn_in_eps;
        p.gn_in_c = 0;
        for (int s = 0; s < d->gn_in_nsrc; ++s) p.gn_in_c += d->src[s].channels;
    }
    p.w_groups = d->w_groups > 1 ? d->w_groups : 1;
    p.w_group_rows = d->w_group_rows;
    p.w_group_stride = d->w_group_stride;
    if (p.w_groups > 1) {
        const int bm_t = 16 * kTiles[pl.tile % 10].mf * kTiles[pl.tile % 10].wm * (pl.tile / 10 == 8 || pl.tile / 10 == 9 ? 2 : 1);
        IDB_REQUIRE(d->w_group_rows > 0 && d->w_group_stride >= (long long)p.w_bytes && d->w_group_stride % 16 == 0 &&
                        (long long)d->w_groups * d->w_group_stride < (1LL << 40), "idb_gemm: w_groups needs w_group_rows > 0 and a 16-byte-multiple w_group_stride >= one matrix");
        if (pl.tile / 10 == 4 || d->w_group_rows % bm_t != 0) {
            idb_set_error("idb_gemm: w_group_rows = %d is not a multiple of the plan's tile height %d (or the persistent variant was chosen): run one launch per group",
                          d->w_group_rows, bm_t);
            return IDB_EUNSUPPORTED;
        }
    }
    p.rowstat_out = d->row_stats_out;
    p.ln_stats = d->ln_stats;
    p.ln_u = d->ln_u;
    p.ln_v = d->ln_v;
    p.ln_nt = d->ln_tiles;
    p.ln_c = (int)pl.K;
    p.ln_eps = d->ln_eps;
    p.slab_swc = 0;
    if (d->gn_partials && pl.splitk > 1 && !p.counters && !(d->flags & 1) && !p.out_f32 && pl.M % 64 == 0) {
        // the reduce launch will be idb_splitk_reduce_gn_kernel: same conditions and slice width as launch_all computes
        if (idb_reduce_vec_ok(p, d->n)) p.slab_swc = gn_reduce_slice(d->n, d->gn_groups);
    }
    hipStream_t st = (hipStream_t)stream;
    return d->dtype == IDB_BF16 ? launch_all<__bf16>(d, p, pl, st) : launch_all<_Float16>(d, p, pl, st);
}


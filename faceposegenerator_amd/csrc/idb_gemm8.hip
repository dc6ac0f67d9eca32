// fp8 (OCP e4m3) implicit GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 — the first piece of BASELINE configs[4]'s "fp8 MFMA weight
// path" (768x768 v-prediction model selected at /root/reference/inference_ID-Booth.py:63-64,103): conv3x3 / 1x1 / Linear with BOTH
// operands in 8 bits, fp32 accumulation, twice the MFMA rate of the bf16/f16 kernels and half their operand bytes.
//
// Same structure as idb_gemm_kernel (idb_gemm.hip): NHWC activations, LDS-DMA staging with hardware zero fill, 128-byte LDS rows
// XOR-swizzled on the source side, weight fragment = MFMA A operand, 8 waves, counted vmcnt, one barrier per K-step, XCD-aware
// block remap, epilogues of idb_gemm_epi.h.  What changes with 1-byte elements:
//   * one K-step is 128 channels of one tap = one 128-byte row (the same LDS geometry) and ONE MFMA per fragment pair
//     (16x16x128: 32 cycles for 4x the K of the 16-cycle bf16 instruction);
//   * channel counts are multiples of 64, so the last K-step of a tap may hold 64 channels (C = 320: 2.5 steps): its upper
//     64 bytes are zero-filled (out-of-range voffset for those lanes) and the weights are packed with the matching zero columns
//     ([n][tap][C rounded up to 128]);
//   * a lane's fragment is 32 contiguous bytes of its row (two ds_read_b128): lane group g takes bytes 32g..32g+31 of BOTH
//     operands — the hardware pairs byte j of group g of A with byte j of group g of B, and a sum over k does not care which
//     32 channels those are (tools/fp8_layout_probe.hip: three different k orders give bit-identical results);
//   * the block scales of the MX instruction are fixed at 1.0 (E8M0 127); the real scales — one per tensor for the activations,
//     one per output channel for the weights — multiply the fp32 accumulator in the epilogue.
#include "idb_gemm_epi.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct Gemm8Params {
    GemmParams g;              // epilogue fields (M, N, HW, bias, sbias, res, out, ...); g.src[0] describes x (bytes = tensor size)
    const float* w_scale;      // [N]
    float x_scale;
    int cpad;                  // channels rounded up to 128: K-steps per tap = cpad / 128
};

template <typename T, int MF, int NF, int NS>
__global__ __launch_bounds__(512, 2) void idb_gemm8_kernel(const Gemm8Params p8) {
#if defined(__HIP_DEVICE_COMPILE__)
    const GemmParams& p = p8.g;
    constexpr int WM = 4, BM = 16 * MF * WM, BN = 32 * NF, THREADS = 512, RS = 64;
    constexpr int NJ = (BN + RS - 1) / RS, STAGE = (BM + NJ * RS) * 128;
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
    const int nk = p.ktiles;

    // staging: thread loads chunk position (tid&7) of rows (tid>>3)+64i; it fetches chunk (tid&7) ^ (row&7) of the 128-byte K-step
    const int lrow = tid >> 3;
    const int chunk = (tid & 7) ^ (lrow & 7);
    const unsigned cg16 = (unsigned)chunk * 16u;
    const bool upper = chunk >= 4;                                   // bytes 64..127: absent in a 64-channel tail step
    int a_b[MF], a_oy[MF], a_ox[MF];
    bool a_ok[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + i * RS + lrow;
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
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, IDB_RSRC_FLAGS);
    const GemmSrcK S = p.src[0];
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
    unsigned w_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + j * RS + lrow;
        w_voff[j] = (n < p.N && j * RS + lrow < BN) ? (unsigned)n * p.w_row_bytes + cg16 : IDB_OOB;
    }
    unsigned w_soff = 0;

    // K-step state: tap (a 1x1 source sits on the centre tap), channel offset c0 (multiples of 128)
    const int ntaps = S.taps == 9 ? 9 : 1;
    int tap = S.taps == 9 ? 0 : 4, c0 = 0;
    unsigned a_voff[MF];
    bool need_retap = true;
    auto stage = [&](int buf) __attribute__((always_inline)) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + BM * 128;
        if (need_retap) {
            const int t3 = tap / 3;
            const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
            const int LH = S.H << S.up, LW = S.W << S.up;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;
                const int pix = (a_b[i] * S.H + (iy >> S.up)) * S.W + (ix >> S.up);
                a_voff[i] = ok ? (unsigned)pix * (unsigned)S.C + cg16 : IDB_OOB;
            }
            need_retap = false;
        }
        const bool tail = c0 + 128 > S.C;                            // 64 real channels: the upper half of the row reads zeros
#pragma unroll
        for (int i = 0; i < MF; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(sA + (i * THREADS + wave * 64) * 16), 16, (tail && upper) ? IDB_OOB : a_voff[i],
                                                     (unsigned)c0, 0, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + (j * THREADS + wave * 64) * 16), 16, w_voff[j], w_soff, 0, 0);
        w_soff += 128u;
        c0 += 128;
        if (c0 >= S.C) {
            c0 = 0;
            need_retap = true;
            ++tap;
        }
    };
    (void)ntaps;

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int LOADS = MF + NJ;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) stage(st);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        if (NS > 2 && it + NS - 2 < nk)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * LOADS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (it + NS - 1 < nk) stage(cur == 0 ? NS - 1 : cur - 1);
        const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
        const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
        const int p0 = ((2 * fg) ^ (fr & 7)) * 16, p1 = ((2 * fg + 1) ^ (fr & 7)) * 16;
        i32x8 af[MF], wf[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const i32x4 lo = *(const i32x4*)(sA + i * 16 * 128 + p0), hi = *(const i32x4*)(sA + i * 16 * 128 + p1);
            af[i] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const i32x4 lo = *(const i32x4*)(sB + j * 16 * 128 + p0), hi = *(const i32x4*)(sB + j * 16 * 128 + p1);
            wf[j] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        cur = cur + 1 == NS ? 0 : cur + 1;
    }

    // dequantise: acc * x_scale * w_scale[n]; lane holds out[m][n .. n+3], n = n0 + (wn*NF + j)*16 + fg*4
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = n0 + (wn * NF + j) * 16 + fg * 4;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        if (n + 3 < p.N) sc = *(const f32x4*)(p8.w_scale + n);
        else
            for (int e = 0; e < 4; ++e) sc[e] = n + e < p.N ? p8.w_scale[n + e] : 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] *= sc[e] * p8.x_scale;
    }
    idb_gemm_epilogue<T, MF, NF, WM>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, 0);
#endif
}


// Loader-wave form (round 3), the fp8 twin of idb_gemm_kernel_lw: LW extra waves do the address arithmetic, the LDS-DMA issue and the
// counted waits; the 8 MFMA waves run barrier -> ds_read -> MFMA only, with the same fragment mapping, accumulation order and epilogue
// (bit-identical outputs).  One s_barrier per K-step for both roles, loaders leave after the loop.  Used as a 256-row tile
// (MF = 4: 256x160 / 256x128, 3-stage ring = all 160 KB of LDS, one workgroup per CU) for large grids.
template <typename T, int MF, int NF, int NS, int LW>
__global__ __launch_bounds__(512 + 64 * LW) void idb_gemm8_kernel_lw(const Gemm8Params p8) {
#if defined(__HIP_DEVICE_COMPILE__)
    const GemmParams& p = p8.g;
    constexpr int WM = 4, BM = 16 * MF * WM, BN = 32 * NF, CT = 512, LR = 8 * LW;
    constexpr int NA = BM / LR, NJ = (BN + LR - 1) / LR, STAGE = (BM + NJ * LR) * 128, LOADS = NA + NJ;
    static_assert(BM % LR == 0, "loader sweep must divide the row tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.ktiles;

    if (wave >= 8) {                                       // ---------------- loader waves ----------------
        const int lw = wave - 8, lt = tid - CT, lrow = lt >> 3;
        const int chunk = (lt & 7) ^ (lrow & 7);
        const unsigned cg16 = (unsigned)chunk * 16u;
        const bool upper = chunk >= 4;
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
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, IDB_RSRC_FLAGS);
        const GemmSrcK S = p.src[0];
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)S.ptr, 0, S.bytes, IDB_RSRC_FLAGS);
        unsigned w_voff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + j * LR + lrow;
            w_voff[j] = (n < p.N && j * LR + lrow < BN) ? (unsigned)n * p.w_row_bytes + cg16 : IDB_OOB;
        }
        unsigned w_soff = 0;
        int tap = S.taps == 9 ? 0 : 4, c0 = 0;
        unsigned a_voff[NA];
        bool need_retap = true;
        auto stage = [&](int buf) __attribute__((always_inline)) {
            char* sA = smem + buf * STAGE;
            char* sB = sA + BM * 128;
            if (need_retap) {
                const int t3 = tap / 3;
                const int dy = t3 - p.pad, dx = tap - t3 * 3 - p.pad;
                const int LH = S.H << S.up, LWd = S.W << S.up;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LWd;
                    const int pix = (a_b[i] * S.H + (iy >> S.up)) * S.W + (ix >> S.up);
                    a_voff[i] = ok ? (unsigned)pix * (unsigned)S.C + cg16 : IDB_OOB;
                }
                need_retap = false;
            }
            const bool tail = c0 + 128 > S.C;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(sA + (i * 64 * LW + lw * 64) * 16), 16, (tail && upper) ? IDB_OOB : a_voff[i],
                                                         (unsigned)c0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(sB + (j * 64 * LW + lw * 64) * 16), 16, w_voff[j], w_soff, 0, 0);
            w_soff += 128u;
            c0 += 128;
            if (c0 >= S.C) {
                c0 = 0;
                need_retap = true;
                ++tap;
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
        return;                                            // s_barrier counts the surviving waves only
    }

    // ---------------- MFMA waves ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* sA = smem + cur * STAGE + (wm * 16 * MF + fr) * 128;
        const char* sB = smem + cur * STAGE + BM * 128 + (wn * 16 * NF + fr) * 128;
        const int p0 = ((2 * fg) ^ (fr & 7)) * 16, p1 = ((2 * fg + 1) ^ (fr & 7)) * 16;
        i32x8 wf[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const i32x4 lo = *(const i32x4*)(sB + j * 16 * 128 + p0), hi = *(const i32x4*)(sB + j * 16 * 128 + p1);
            wf[j] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const i32x4 lo = *(const i32x4*)(sA + i * 16 * 128 + p0), hi = *(const i32x4*)(sA + i * 16 * 128 + p1);
            const i32x8 af = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int j = 0; j < NF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = n0 + (wn * NF + j) * 16 + fg * 4;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        if (n + 3 < p.N) sc = *(const f32x4*)(p8.w_scale + n);
        else
            for (int e = 0; e < 4; ++e) sc[e] = n + e < p.N ? p8.w_scale[n + e] : 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] *= sc[e] * p8.x_scale;
    }
    idb_gemm_epilogue<T, MF, NF, WM>(p, smem, acc, m0, n0, tid, wm, wn, fr, fg, 0);
#endif
}

// x (operand dtype) -> fp8 e4m3 of x * inv_scale, saturating at +-448; 8 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* x, unsigned char* out, long long count, float inv_scale) {
    const long long i8 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i8 >= count) return;
    const typename Op<T>::v8 raw = *(const typename Op<T>::v8*)(x + i8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(to_f32<T>(raw[e]) * inv_scale, -448.f), 448.f);
    unsigned lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0u, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    unsigned hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0u, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    *(u32x2*)(out + i8) = (u32x2){lo, hi};
}

// weights: fp32 [cout][cin][taps] (torch Conv2d layout, taps = kh*kw; Linear: taps = 1) -> fp8 [cout][taps][cpad] with a per-row scale
// scale[n] = absmax(row) / 448 (1 for an all-zero row), zero columns for cin..cpad-1.  One workgroup per output channel.
__global__ __launch_bounds__(256) void pack_weight_fp8_kernel(const float* src, unsigned char* dst, float* scales, int cin, int taps, int cpad) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* row = src + (long long)n * cin * taps;
    float mx = 0.f;
    for (int i = tid; i < cin * taps; i += 256) mx = fmaxf(mx, fabsf(row[i]));
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = mx > 0.f ? mx / 448.f : 1.f;
    const float inv = 1.f / scale;
    if (tid == 0) scales[n] = scale;
    unsigned char* out = dst + (long long)n * taps * cpad;
    for (int i = tid * 2; i < taps * cpad; i += 512) {
        const int t = i / cpad, c = i - t * cpad;                    // cpad is even: the pair stays inside one tap
        const float a = c < cin ? row[(long long)c * taps + t] * inv : 0.f;
        const float b = c + 1 < cin ? row[(long long)(c + 1) * taps + t] * inv : 0.f;
        const unsigned pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0u, false);
        *(unsigned short*)(out + i) = (unsigned short)(pk & 0xffffu);
    }
}

template <typename T, int MF, int NF, int NS>
int launch_gemm8(const Gemm8Params& p, int tiles, hipStream_t st) {
    constexpr int NJ = (32 * NF + 63) / 64;
    constexpr int LDS = (16 * MF * 4 + NJ * 64) * 128 * NS;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm8_kernel<T, MF, NF, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm_fp8: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((idb_gemm8_kernel<T, MF, NF, NS>), dim3(tiles), dim3(512), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm_fp8");
    return IDB_OK;
}

template <typename T, int MF, int NF, int NS, int LW>
int launch_gemm8_lw(const Gemm8Params& p, int tiles, hipStream_t st) {
    constexpr int LR = 8 * LW, NJ = (32 * NF + LR - 1) / LR;
    constexpr int LDS = (16 * MF * 4 + NJ * LR) * 128 * NS;
    static_assert(LDS <= 160 * 1024, "LDS ring does not fit");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&idb_gemm8_kernel_lw<T, MF, NF, NS, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            idb_set_error("idb_gemm_fp8: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
            return IDB_EHIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((idb_gemm8_kernel_lw<T, MF, NF, NS, LW>), dim3(tiles), dim3(512 + 64 * LW), LDS, st, p);
    IDB_CHECK_LAUNCH("idb_gemm_fp8(lw)");
    return IDB_OK;
}

}  // namespace

extern "C" int idb_quantize_fp8(const void* x, void* out, int64_t count, float inv_scale, int32_t dtype, void* stream) {
    IDB_REQUIRE(x && out && count > 0 && count % 8 == 0 && idb_aligned16(x) && (((uintptr_t)out) & 7) == 0 && idb_is_operand_dtype(dtype),
                "idb_quantize_fp8: count must be a positive multiple of 8, pointers aligned, dtype bf16/f16");
    const unsigned blocks = (unsigned)((count / 8 + 255) / 256);
    if (dtype == IDB_BF16)
        hipLaunchKernelGGL(quantize_fp8_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (unsigned char*)out, (long long)count, inv_scale);
    else
        hipLaunchKernelGGL(quantize_fp8_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (unsigned char*)out, (long long)count, inv_scale);
    IDB_CHECK_LAUNCH("idb_quantize_fp8");
    return IDB_OK;
}

extern "C" int idb_pack_weight_fp8(const float* src, void* dst, float* scales, int32_t cout, int32_t cin, int32_t ktaps, void* stream) {
    IDB_REQUIRE(src && dst && scales && cout > 0 && cin > 0 && cin % 64 == 0 && (ktaps == 9 || ktaps == 1), "idb_pack_weight_fp8: cin must be a multiple of 64, ktaps 9 or 1");
    const int cpad = (cin + 127) / 128 * 128;
    hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3(cout), dim3(256), 0, (hipStream_t)stream, src, (unsigned char*)dst, scales, cin, ktaps, cpad);
    IDB_CHECK_LAUNCH("idb_pack_weight_fp8");
    return IDB_OK;
}

extern "C" int idb_gemm_fp8(const idb_gemm_fp8_desc* d, void* stream) {
    IDB_REQUIRE(d != nullptr, "idb_gemm_fp8: null descriptor");
    IDB_REQUIRE(idb_is_operand_dtype(d->out_dtype), "idb_gemm_fp8: out_dtype must be bf16 or f16");
    IDB_REQUIRE(d->batch > 0 && d->out_h > 0 && d->out_w > 0 && d->n > 0 && (d->stride == 1 || d->stride == 2), "idb_gemm_fp8: bad dims");
    IDB_REQUIRE(d->x && idb_aligned16(d->x) && d->channels > 0 && d->channels % 64 == 0 && (d->taps == 9 || d->taps == 1) && d->in_h > 0 && d->in_w > 0 &&
                    (d->upsample == 0 || d->upsample == 1), "idb_gemm_fp8: source invalid (channels %% 64, taps 9 or 1)");
    const int lh = d->in_h << d->upsample, lw = d->in_w << d->upsample;
    if (d->taps == 9)
        IDB_REQUIRE(d->out_h == (lh + d->stride - 1) / d->stride && d->out_w == (lw + d->stride - 1) / d->stride, "idb_gemm_fp8: source does not produce the output grid");
    else
        IDB_REQUIRE(d->stride == 1 && lh == d->out_h && lw == d->out_w, "idb_gemm_fp8: 1x1 source must match the output grid");
    IDB_REQUIRE(d->w && idb_aligned16(d->w) && d->w_scale && idb_aligned16(d->w_scale) && d->out && idb_aligned16(d->out) && d->x_scale > 0.f, "idb_gemm_fp8: w / w_scale / out / x_scale invalid");
    IDB_REQUIRE(d->out_ld >= d->n && d->n % 8 == 0 && d->out_ld % 8 == 0 && (!d->residual || idb_aligned16(d->residual)), "idb_gemm_fp8: n and out_ld multiples of 8, residual aligned");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    const int cpad = (d->channels + 127) / 128 * 128;
    const long long K = (long long)d->taps * cpad;
    IDB_REQUIRE(M < (1LL << 31) && (long long)d->batch * d->in_h * d->in_w * d->channels < (1LL << 31) && (long long)d->n * K < (1LL << 31) &&
                    M * d->out_ld * 2 < (1LL << 31), "idb_gemm_fp8: a tensor is >= 2 GiB; split the batch");
    Gemm8Params p = {};
    GemmParams& g = p.g;
    for (int s = 0; s < IDB_MAX_SRC; ++s)
        g.src[s] = GemmSrcK{(const char*)d->x, (unsigned)((long long)d->batch * d->in_h * d->in_w * d->channels), d->channels, d->taps, d->in_h, d->in_w, d->upsample};
    g.M = (int)M;
    g.N = d->n;
    g.HW = d->out_h * d->out_w;
    g.OW = d->out_w;
    g.stride = d->stride;
    g.pad = 1;
    g.w_row_bytes = (unsigned)K;
    g.w_bytes = (unsigned)((long long)d->n * K);
    g.ktiles = (int)(K / 128);
    g.kt_per_split = g.ktiles;
    g.splitk = 1;
    g.w = (const char*)d->w;
    g.bias = d->bias;
    g.sbias = d->sample_bias;
    g.sbias_ld = d->sample_bias_ld;
    g.res = (const char*)d->residual;
    g.out = d->out;
    g.out_ld = d->out_ld;
    g.scale = 1.f;
    g.lds_epi = 1;
    p.w_scale = d->w_scale;
    p.x_scale = d->x_scale;
    p.cpad = cpad;
    const int nf = d->n % 160 == 0 ? 5 : 4;
    g.tiles_n = (d->n + 32 * nf - 1) / (32 * nf);
    // large grids with K >= 1280: 256-row loader-wave tiles (one workgroup per CU, 8 MFMA + 4 loader waves); IDB_GEMM8_BIG_TILES=0: off
    const char* env_big_s = getenv("IDB_GEMM8_BIG_TILES");          // read per call: tests compare both tile forms in one process
    const int env_big = env_big_s ? atoi(env_big_s) : 512;
    const bool big = env_big > 0 && g.ktiles >= 10 && ((M + 255) / 256) * g.tiles_n >= env_big;
    const int bm = big ? 256 : 128;
    const int tiles = (int)((M + bm - 1) / bm) * g.tiles_n;
    hipStream_t st = (hipStream_t)stream;
    bool gn_after = false;                 // first GroupNorm pass of the output: from the shared LDS-staged epilogue when it can
    if (d->gn_partials) {
        IDB_REQUIRE(d->gn_groups > 0 && d->n % d->gn_groups == 0 && g.HW % 64 == 0 && g.HW <= 4096 && d->out_ld == d->n && idb_aligned16(d->gn_partials),
                    "idb_gemm_fp8: gn_partials needs n %% gn_groups == 0, out_h*out_w %% 64 == 0 and <= 4096, dense output");
        if (idb_epilogue_emits_gn(bm, 32 * nf, 512, M, d->n, d->gn_groups)) {
            g.gn_part = d->gn_partials;
            g.gn_groups = d->gn_groups;
        } else {
            gn_after = true;
        }
    }
    int rc;
    if (big) {
        if (d->out_dtype == IDB_BF16) rc = nf == 5 ? launch_gemm8_lw<__bf16, 4, 5, 3, 4>(p, tiles, st) : launch_gemm8_lw<__bf16, 4, 4, 3, 4>(p, tiles, st);
        else rc = nf == 5 ? launch_gemm8_lw<_Float16, 4, 5, 3, 4>(p, tiles, st) : launch_gemm8_lw<_Float16, 4, 4, 3, 4>(p, tiles, st);
    } else if (d->out_dtype == IDB_BF16) rc = nf == 5 ? launch_gemm8<__bf16, 2, 5, 2>(p, tiles, st) : launch_gemm8<__bf16, 2, 4, 2>(p, tiles, st);
    else rc = nf == 5 ? launch_gemm8<_Float16, 2, 5, 2>(p, tiles, st) : launch_gemm8<_Float16, 2, 4, 2>(p, tiles, st);
    if (rc != IDB_OK || !gn_after) return rc;
    return idb_launch_gn_stats64(d->out, d->n, d->batch, g.HW, d->gn_groups, d->gn_partials, d->out_dtype, st);
}

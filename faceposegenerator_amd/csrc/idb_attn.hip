// K4/K5 — fused (flash-style) attention for head_dim 64 on gfx950 MFMA, fp32 online softmax.
//
// Replaces F.scaled_dot_product_attention inside diffusers AttnProcessor2_0 (self-attention over
// 4096/1024/256/64 latent tokens, cross-attention to 77 text tokens), reached from
// /root/reference/inference_ID-Booth.py:138 via UNet2DConditionModel.forward.
//
// Structure (MI355X): one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32
// query rows for the whole K/V sweep.  K/V tiles of 64 keys are double-buffered in LDS (register
// staging: the next tile's global loads are issued before the current tile's MFMAs, written to the
// other LDS buffer afterwards; one barrier per tile).
//   S^T = K Q^T   v_mfma_f32_32x32x16 with K as the A operand and Q^T as the B operand, so the query
//                 index sits on the lane and a lane holds 32 of its query's 64 scores: row max / row sum
//                 are in-register reductions plus ONE exchange with lane^32.
//   O^T = V^T P^T the S accumulator, converted to bf16/f16, IS the B operand of the second product
//                 (no LDS round trip for P); V^T fragments come from the row-major V tile through
//                 ds_read_b64_tr_b16 (hardware transpose read), in the k-order the accumulator imposes.
// LDS rows are padded (K: 144 B, V: 192 B) so ds_read_b128 / ds_read_b64_tr_b16 are conflict-free.
#include "idb_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KS = 144;   // K tile row stride (bytes): 128 + one 16-B access width
constexpr int VS = 192;   // V tile row stride (bytes): (VS/4) mod 64 == 48 -> tr reads conflict-free
constexpr int KV_TILE = 64;
constexpr int K_BYTES = KV_TILE * KS, V_BYTES = KV_TILE * VS, BUF_BYTES = K_BYTES + V_BYTES;

// NW waves per workgroup; KSPLIT = 1: every wave owns 32 query rows and all 64 keys of a tile (32*NW query rows per workgroup;
// NW = 4 for large grids, 2 for small ones).  KSPLIT = 2 (NW = 8): waves w and w + 4 share 32 query rows and take the lower /
// upper 32 keys of every tile — the same K/V tile in LDS, half the MFMA and softmax work per wave and tile, two waves per SIMD —
// and merge their (max, sum, O) through LDS at the end.  For grids of about one workgroup per CU (the 64x64 level at batch 1:
// 320 workgroups) a lone 4-wave workgroup leaves one wave per SIMD, whose softmax VALU work and MFMAs cannot overlap.
template <typename T, int NW, int KSPLIT = 1>
__global__ __launch_bounds__(64 * NW) void attn_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ k,
                                                   const T* __restrict__ v, int kv_ld, T* __restrict__ out, int out_ld,
                                                   int n_q, int n_kv, int n_kv_alloc, float scale_log2e, int causal) {
    using V8 = typename Op<T>::v8;
    using V4 = typename Op<T>::v4;
    constexpr int MERGE_BYTES = KSPLIT == 2 ? 34 * (NW / KSPLIT) * 64 * 4 : 0;       // (m, l, O[32]) per lane of the upper-key waves
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES > MERGE_BYTES ? 2 * BUF_BYTES : MERGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    constexpr int NWQ = NW / KSPLIT, NKB = 2 / KSPLIT;        // waves along the query rows; 32-key blocks per wave and tile
    const int wq = wave % NWQ, kb0 = (wave / NWQ) * NKB;
    const int q0 = blockIdx.x * (32 * NWQ) + wq * 32;

    // ---- Q^T fragments (B operand): lane (r,h) holds Q[q0+r][16s + 8h .. +7], s = 0..3
    V8 qf[4];
    {
        const int qr = min(q0 + r, n_q - 1);
        const T* qp = q + ((long long)b * n_q + qr) * q_ld + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const V8*)(qp + 16 * s);
    }

    const T* kbase = k + (long long)b * n_kv_alloc * kv_ld + head * 64;
    const T* vbase = v + (long long)b * n_kv_alloc * kv_ld + head * 64;
    const int ntiles = (n_kv + KV_TILE - 1) / KV_TILE;

    // staging: thread loads 16-B chunk (tid&7) of rows (tid>>3) + 8*NW*i of the K and V tiles into registers and writes them
    // to the other LDS buffer after the current tile's MFMAs.  PF register sets: PF = 1 fetches tile t+1 during tile t; PF = 2
    // would fetch tile t+2 (kept as a switch; see below).
    constexpr int PF = 1;   // PF = 2 measured slower: 167 VGPRs (one workgroup per CU) or spills under a 128-VGPR cap, and a tile's
                            // time is the S -> softmax -> PV dependency chain of a wave, not the K/V round trip
    constexpr int CH = NW >= 8 ? 1 : 8 / NW, RSTEP = NW >= 8 ? 64 : 8 * NW;     // NW > 8: waves 8.. do not stage
    const bool stager = NW <= 8 || tid < 512;                                    // wave-uniform
    const int srow = tid >> 3, schunk = tid & 7;
    V8 kreg[PF][CH], vreg[PF][CH];
    auto gload = [&](auto SET, int t) {
        constexpr int st = decltype(SET)::value;
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int row = min(t * KV_TILE + srow + RSTEP * i, n_kv - 1);   // clamp: masked later, must stay finite
            kreg[st][i] = *(const V8*)(kbase + (long long)row * kv_ld + schunk * 8);
            vreg[st][i] = *(const V8*)(vbase + (long long)row * kv_ld + schunk * 8);
        }
    };
    auto lstore = [&](auto SET, int buf) {
        constexpr int st = decltype(SET)::value;
        if (!stager) return;
        char* kb_ = smem + buf * BUF_BYTES;
        char* vb_ = kb_ + K_BYTES;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            *(V8*)(kb_ + (srow + RSTEP * i) * KS + schunk * 16) = kreg[st][i];
            *(V8*)(vb_ + (srow + RSTEP * i) * VS + schunk * 16) = vreg[st][i];
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, PF - 1>;

    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    gload(Set0{}, 0);
    if constexpr (PF == 2) {
        if (ntiles > 1) gload(Set1{}, 1);
    }
    lstore(Set0{}, 0);
    __syncthreads();

    // V^T fragment addressing: 16-lane group G = lane>>4 = 2h + (r>>4); lane i of the group supplies the
    // address of V[row + (i>>2)][col + 4*(i&3)] and receives column i of those 4 rows.
    const int tr_row = 4 * h + ((lane & 15) >> 2);
    const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    auto tile_body = [&](auto TAIL, auto PAR, int t) {      // PAR: register set of tile t (PF = 2: t & 1; PF = 1: 0)
        const int cur = t & 1;
        using Next = std::integral_constant<int, PF == 2 ? 1 - decltype(PAR)::value : 0>;
        if constexpr (PF == 2) {
            if (t + 2 < ntiles) gload(PAR, t + 2);           // tile t's set was written to LDS before this tile started
        } else {
            if (t + 1 < ntiles) gload(PAR, t + 1);
        }
        const char* kt = smem + cur * BUF_BYTES;
        const char* vt = kt + K_BYTES;

        // ---- S^T[key][q] for the 64 keys of this tile: 2 key blocks x 4 k-steps over d
        f32x16 sacc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const V8 kf = *(const V8*)(kt + ((kb0 + kb) * 32 + r) * KS + (2 * s + h) * 16);
                sacc[kb] = Op<T>::mfma32(kf, qf[s], s == 0 ? zero : sacc[kb]);     // literal-0 C operand on the first step
            }
        }
        // online softmax on the RAW scores: the scale (and log2 e) rides in the exp2 argument as one FMA per element
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (decltype(TAIL)::value) {        // mask keys beyond n_kv (last, partial tile only)
                    const int key = t * KV_TILE + (kb0 + kb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= n_kv || (causal && key > q0 + r)) sacc[kb][i] = -INFINITY;
                }
                mx = fmaxf(mx, sacc[kb][i]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
        const bool grew = m_new > m_run;
        m_run = m_new;
        const float mb = m_new * scale_log2e;
        float rs = 0.f;
        V8 pf[NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][8 * s2 + j], scale_log2e, -mb));
                    rs += pv;
                    pf[kb][s2][j] = from_f32<T>(pv);
                }
        l_run = l_run * alpha + rs;
        if (__builtin_amdgcn_ballot_w64(grew) != 0) {         // wave-uniform: no row max moved -> alpha == 1 everywhere
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
        }

        // ---- O^T[d][q] += V^T P^T
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int base_row = (kb0 + kb) * 32 + 16 * s2 + tr_row;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const char* va = vt + base_row * VS + (d * 32 + tr_col) * 2;
                    const V4 lo = Op<T>::ds_read_tr(va);
                    const V4 hi = Op<T>::ds_read_tr(va + 8 * VS);
                    V8 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[j] = lo[j];
                        vf[j + 4] = hi[j];
                    }
                    oacc[d] = Op<T>::mfma32(vf, pf[kb][s2], oacc[d]);
                }
            }

        if (t + 1 < ntiles) lstore(Next{}, cur ^ 1);
        __syncthreads();
    };
    auto run = [&](auto TAIL, int t) {
        if constexpr (PF == 2) {
            if (t & 1) tile_body(TAIL, Set1{}, t);
            else tile_body(TAIL, Set0{}, t);
        } else {
            tile_body(TAIL, Set0{}, t);
        }
    };
    const int nfull = causal ? 0 : n_kv / KV_TILE;            // tiles without masking code at all
    for (int t = 0; t < nfull; ++t) run(std::false_type{}, t);
    for (int t = nfull; t < ntiles; ++t) run(std::true_type{}, t);     // partial last tile, or every tile when causal

    if constexpr (KSPLIT == 2) {
        // merge the upper-key waves into the lower-key waves of the same query rows (the K loop ended on a barrier: LDS is free);
        // component-major layout [34][NWQ * 64]: conflict-free, 34 * 256 * 4 B = 34.8 KB of the 43 KB tile buffers
        float* xs = (float*)smem;
        constexpr int LN = NWQ * 64;
        const int slot = wq * 64 + lane;
        if (kb0 != 0) {
            xs[slot] = m_run;
            xs[LN + slot] = l_run;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) xs[(2 + d * 16 + i) * LN + slot] = oacc[d][i];
        }
        __syncthreads();
        if (kb0 != 0) return;
        const float m_o = xs[slot], l_o = xs[LN + slot];
        const float m_new = fmaxf(m_run, m_o);
        // a wave that saw only masked keys has m = -inf and l = 0, O = 0: its weight must be 0, not exp2(nan)
        const float a_s = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
        const float a_o = m_o == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_o - m_new) * scale_log2e);
        l_run = l_run * a_s + l_o * a_o;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] = oacc[d][i] * a_s + xs[(2 + d * 16 + i) * LN + slot] * a_o;
    }
    // ---- epilogue: lane (q = r, half h) holds O[q][32*d + 8*(i>>2) + 4h + (i&3)]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q0 + r < n_q) {
        T* op = out + ((long long)b * n_q + q0 + r) * out_ld + head * 64 + 4 * h;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                V4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = from_f32<T>(oacc[d][4 * g4 + e] * inv);
                *(V4*)(op + d * 32 + 8 * g4) = pk;
            }
    }
}

}  // namespace

extern "C" int idb_attention(const void* q, int32_t q_ld, const void* k, const void* v, int32_t kv_ld, void* out,
                             int32_t out_ld, int32_t batch, int32_t heads, int32_t n_q, int32_t n_kv, int32_t n_kv_alloc,
                             float scale, int32_t causal, int32_t dtype, void* stream) {
    IDB_REQUIRE(idb_is_operand_dtype(dtype), "idb_attention: dtype must be bf16/f16");
    IDB_REQUIRE(q && k && v && out, "idb_attention: null pointer");
    IDB_REQUIRE(idb_aligned16(q) && idb_aligned16(k) && idb_aligned16(v) && idb_aligned16(out), "idb_attention: unaligned pointer");
    IDB_REQUIRE(batch > 0 && heads > 0 && n_q > 0 && n_kv > 0 && n_kv_alloc >= n_kv, "idb_attention: bad dims");
    IDB_REQUIRE(q_ld % 8 == 0 && kv_ld % 8 == 0 && out_ld % 4 == 0, "idb_attention: row strides must be multiples of 8 (q,kv) / 4 (out)");
    IDB_REQUIRE(q_ld >= heads * 64 && kv_ld >= heads * 64 && out_ld >= heads * 64, "idb_attention: row stride < heads*64");
    IDB_REQUIRE(batch <= 65535 && heads <= 65535, "idb_attention: grid too large");
    IDB_REQUIRE(!causal || n_q == n_kv, "idb_attention: causal needs n_q == n_kv");
    const float sl2 = scale * 1.44269504088896340736f;
    hipStream_t st = (hipStream_t)stream;
    // 64-row blocks only when 128-row blocks would leave half the CUs idle (measured: slower otherwise)
    const long long blocks128 = (long long)((n_q + 127) / 128) * heads * batch;
    const bool small = blocks128 < 128;
    // about one 128-row workgroup per CU and a long key sweep: 8 waves, key halves split between wave pairs (attn_kernel)
    static const int env_ks = [] { const char* e = getenv("IDB_ATTN_KSPLIT"); return e ? atoi(e) : 1; }();
    // (on larger grids the 8-wave form loses: batch 64 15.44 -> 15.22 images/s, batch 8 14.16 -> 13.96 with it everywhere)
    const bool ksplit = env_ks && !small && !causal && blocks128 < 512 && n_kv >= 512;
    // 257-511 workgroups of 128 rows = two uneven rounds (the CUs that get two set the time): 192-row workgroups (12 waves, three
    // per SIMD) when that grid fits one round — the 64x64 level at batch 1: 320 -> 220 workgroups
    const long long blocks192 = (long long)((n_q + 191) / 192) * heads * batch;
    const bool wide = ksplit && env_ks != 2 && blocks128 > 256 && blocks192 <= 256;
    const dim3 grid(wide ? (n_q + 191) / 192 : (n_q + (small ? 63 : 127)) / (small ? 64 : 128), heads, batch);
#define IDB_ATTN_LAUNCH(T, NW)                                                                                          \
    hipLaunchKernelGGL((attn_kernel<T, NW>), grid, dim3(64 * NW), 0, st, (const T*)q, q_ld, (const T*)k, (const T*)v, kv_ld, \
                       (T*)out, out_ld, n_q, n_kv, n_kv_alloc, sl2, causal)
#define IDB_ATTN_LAUNCH_KS(T)                                                                                           \
    do {                                                                                                                \
        if (wide)                                                                                                       \
            hipLaunchKernelGGL((attn_kernel<T, 12, 2>), grid, dim3(768), 0, st, (const T*)q, q_ld, (const T*)k, (const T*)v, kv_ld, \
                               (T*)out, out_ld, n_q, n_kv, n_kv_alloc, sl2, causal);                                  \
        else                                                                                                            \
            hipLaunchKernelGGL((attn_kernel<T, 8, 2>), grid, dim3(512), 0, st, (const T*)q, q_ld, (const T*)k, (const T*)v, kv_ld,  \
                               (T*)out, out_ld, n_q, n_kv, n_kv_alloc, sl2, causal);                                  \
    } while (0)
    if (dtype == IDB_BF16) {
        if (small) IDB_ATTN_LAUNCH(__bf16, 2);
        else if (ksplit) IDB_ATTN_LAUNCH_KS(__bf16);
        else IDB_ATTN_LAUNCH(__bf16, 4);
    } else {
        if (small) IDB_ATTN_LAUNCH(_Float16, 2);
        else if (ksplit) IDB_ATTN_LAUNCH_KS(_Float16);
        else IDB_ATTN_LAUNCH(_Float16, 4);
    }
#undef IDB_ATTN_LAUNCH
#undef IDB_ATTN_LAUNCH_KS
    IDB_CHECK_LAUNCH("idb_attention");
    return IDB_OK;
}

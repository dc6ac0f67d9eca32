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
#include <type_traits>

namespace {

constexpr int KS = 144;   // K tile row stride (bytes): 128 + one 16-B access width
constexpr int VS = 192;   // V tile row stride (bytes): (VS/4) mod 64 == 48 -> tr reads conflict-free
constexpr int KV_TILE = 64;
constexpr int K_BYTES = KV_TILE * KS, V_BYTES = KV_TILE * VS, BUF_BYTES = K_BYTES + V_BYTES;

template <typename T, int NW>   // NW waves = 32*NW query rows per workgroup (4: large grids, 2: small grids)
__global__ __launch_bounds__(64 * NW) void attn_kernel(const T* __restrict__ q, int q_ld, const T* __restrict__ k,
                                                   const T* __restrict__ v, int kv_ld, T* __restrict__ out, int out_ld,
                                                   int n_q, int n_kv, int n_kv_alloc, float scale_log2e, int causal) {
    using V8 = typename Op<T>::v8;
    using V4 = typename Op<T>::v4;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;

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

    // staging: thread loads 16-B chunk (tid&7) of rows (tid>>3) + 8*NW*i of the K and V tiles
    constexpr int CH = 8 / NW, RSTEP = 8 * NW;
    const int srow = tid >> 3, schunk = tid & 7;
    V8 kreg[CH], vreg[CH];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int row = min(t * KV_TILE + srow + RSTEP * i, n_kv - 1);   // clamp: masked later, must stay finite
            kreg[i] = *(const V8*)(kbase + (long long)row * kv_ld + schunk * 8);
            vreg[i] = *(const V8*)(vbase + (long long)row * kv_ld + schunk * 8);
        }
    };
    auto lstore = [&](int buf) {
        char* kb_ = smem + buf * BUF_BYTES;
        char* vb_ = kb_ + K_BYTES;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            *(V8*)(kb_ + (srow + RSTEP * i) * KS + schunk * 16) = kreg[i];
            *(V8*)(vb_ + (srow + RSTEP * i) * VS + schunk * 16) = vreg[i];
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();

    // V^T fragment addressing: 16-lane group G = lane>>4 = 2h + (r>>4); lane i of the group supplies the
    // address of V[row + (i>>2)][col + 4*(i&3)] and receives column i of those 4 rows.
    const int tr_row = 4 * h + ((lane & 15) >> 2);
    const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    auto tile_body = [&](auto TAIL, int t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const char* kt = smem + cur * BUF_BYTES;
        const char* vt = kt + K_BYTES;

        // ---- S^T[key][q] for the 64 keys of this tile: 2 key blocks x 4 k-steps over d
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const V8 kf = *(const V8*)(kt + (kb * 32 + r) * KS + (2 * s + h) * 16);
                sacc[kb] = Op<T>::mfma32(kf, qf[s], s == 0 ? zero : sacc[kb]);     // literal-0 C operand on the first step
            }
        }
        // online softmax on the RAW scores: the scale (and log2 e) rides in the exp2 argument as one FMA per element
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (decltype(TAIL)::value) {        // mask keys beyond n_kv (last, partial tile only)
                    const int key = t * KV_TILE + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
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
        V8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
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
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int base_row = kb * 32 + 16 * s2 + tr_row;
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

        if (t + 1 < ntiles) lstore(cur ^ 1);
        __syncthreads();
    };
    const int nfull = causal ? 0 : n_kv / KV_TILE;            // tiles without masking code at all
    for (int t = 0; t < nfull; ++t) tile_body(std::false_type{}, t);
    for (int t = nfull; t < ntiles; ++t) tile_body(std::true_type{}, t);     // partial last tile, or every tile when causal

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
    const dim3 grid((n_q + (small ? 63 : 127)) / (small ? 64 : 128), heads, batch);
#define IDB_ATTN_LAUNCH(T, NW)                                                                                          \
    hipLaunchKernelGGL((attn_kernel<T, NW>), grid, dim3(64 * NW), 0, st, (const T*)q, q_ld, (const T*)k, (const T*)v, kv_ld, \
                       (T*)out, out_ld, n_q, n_kv, n_kv_alloc, sl2, causal)
    if (dtype == IDB_BF16) {
        if (small) IDB_ATTN_LAUNCH(__bf16, 2);
        else IDB_ATTN_LAUNCH(__bf16, 4);
    } else {
        if (small) IDB_ATTN_LAUNCH(_Float16, 2);
        else IDB_ATTN_LAUNCH(_Float16, 4);
    }
#undef IDB_ATTN_LAUNCH
    IDB_CHECK_LAUNCH("idb_attention");
    return IDB_OK;
}

// K4/K5: flash-style attention for head dim 64 on bf16 MFMA (gfx950), online softmax in fp32.
//
// Workgroup = 4 waves; each wave owns 32 query rows of one (image, head); the workgroup walks the keys in
// tiles of 64, K and V tiles double-buffered in LDS (register-staged loads issued before the MFMAs of the
// current tile, written after them, one barrier per tile).
//
// Per 32-key block the wave computes S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS via
// ds_read_b128, B = Q^T held in registers), so the query index sits on the LANE and the 32 keys of the
// block sit in the 16 accumulator registers of the two half-waves: row max / row sum are in-lane
// reductions plus one exchange with lane^32.  The exponentiated block, converted pairwise to bf16, is
// directly the B operand of O^T += V^T . P^T (k-order of an accumulator used as operand: key
// 16s + 8(j>>2) + 4h + (j&3) for element j of lane-half h), and V^T fragments in exactly that k-order come
// from ds_read_b64_tr_b16 on the row-major V tile.  O^T keeps the query on the lane as well, so the
// running rescale is one scalar per lane.
#include "common.h"
#include <type_traits>

namespace {

constexpr int AT_THREADS = 256;
constexpr int KV_TILE = 64;
constexpr int TILE_BYTES = KV_TILE * 128;  // 64 rows x 64 bf16

struct AttnArgs {
    const __bf16* Q; const __bf16* K; const __bf16* V; __bf16* O;
    int ldq, ldk, ldv, ldo, B, H, Sq, Skv;
    float scale_log2e;
    unsigned k_bytes, v_bytes;
    float* lse;  // optional [B][H][Sq]: log2 of the softmax denominator in the scaled log2 domain (training: the backward's P)
    // backward passes only
    const __bf16* dO; __bf16* dQ; __bf16* dK; __bf16* dV; const float* delta;
    int lddo, lddq, lddk, lddv;
    float scale;
};

__device__ __forceinline__ s16x4 ds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// launch bound of 2 waves per SIMD: caps the kernel at 256 registers so hipcc keeps MFMA results in VGPRs (with 512
// allowed it parks S and O in AGPRs and pays ~160 v_accvgpr moves per tile around the softmax).
// CAUSAL: key j contributes to query i only for j <= i (the text tower's mask; Sq == Skv there)
template <bool CAUSAL>
__global__ __launch_bounds__(AT_THREADS, 2) void attn_d64_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // K0 V0 K1 V1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // Q^T fragments (B operand): lane holds Q[q0 + l31][16 ks + 8 hh .. +7]
    bf16x8 qf[4];
    {
        const int qrow = q0 + l31;
        const bool ok = qrow < a.Sq;
        const __bf16* qp = a.Q + ((long long)b * a.Sq + (ok ? qrow : 0)) * a.ldq + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 v = *(const u32x4*)(qp + ks * 16);
            if (!ok) v = u32x4{0, 0, 0, 0};
            qf[ks] = __builtin_bit_cast(bf16x8, v);
        }
    }

    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.K + (long long)b * a.Skv * a.ldk + h * 64), 0, a.k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.V + (long long)b * a.Skv * a.ldv + h * 64), 0, a.v_bytes, 0x00020000);

    // staging: 64 rows x 8 chunks of 16 B = 512 chunks per tile, 2 per thread for K and for V
    const int lrow = tid >> 3, lchunk = tid & 7;
    u32x4 kr[2], vr[2];
    auto issue_loads = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = tile * KV_TILE + lrow + i * 32;
            const bool ok = key < a.Skv;
            const unsigned ko = ok ? (unsigned)(key * a.ldk + lchunk * 8) * 2u : 0x80000000u;
            const unsigned vo = ok ? (unsigned)(key * a.ldv + lchunk * 8) * 2u : 0x80000000u;
            kr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)ko, 0, 0));
            vr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)vo, 0, 0));
        }
    };
    auto store_stage = [&](int s) {
        char* kb = smem + s * 2 * TILE_BYTES;
        char* vb = kb + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lrow + i * 32;
            *(u32x4*)(kb + r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4)) = kr[i];
            *(u32x4*)(vb + r * 128 + ((lchunk ^ (((r >> 1) & 1) << 2)) << 4)) = vr[i];
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int kswz = (l31 >> 1) & 7;
    // tr16 read addressing: 16-lane group g covers d columns 16g..16g+15 of the 32-wide d block; lane 4q+p of
    // the group supplies row q, columns 4p..4p+3.
    const int g16 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    int vlane[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int chunk = d * 4 + g16 * 2 + (tp >> 1);
        vlane[d] = (4 * hh + tq) * 128 + ((chunk ^ (((tq >> 1) & 1) << 2)) << 4) + (tp & 1) * 8;
    }

    // CAUSAL: the 128 queries of the block see keys < (blockIdx.x + 1) * 128 at most: key tiles entirely above the diagonal are not walked
    const int kv_end = CAUSAL ? min(a.Skv, ((int)blockIdx.x + 1) * 128) : a.Skv;
    const int ntiles = (kv_end + KV_TILE - 1) / KV_TILE;
    issue_loads(0);
    store_stage(0);
    __syncthreads();
    auto tile_body = [&](int t, auto tail_c, auto nb_c) {
        constexpr bool tail = decltype(tail_c)::value;
        constexpr int NB = decltype(nb_c)::value;  // 32-key blocks of this tile that hold keys (1 only in a ragged tail)
        const bool more = t + 1 < ntiles;
        if (more) issue_loads(t + 1);
        const char* kb = smem + (t & 1) * 2 * TILE_BYTES;
        const char* vb = kb + TILE_BYTES;

        // S^T for the two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int kbk = 0; kbk < NB; ++kbk) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kbk][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kb + (kbk * 32 + l31) * 128 + (((2 * ks + hh) ^ kswz) << 4));
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kbk], 0, 0, 0);
            }
        }
        // scale into the log2 domain, mask keys beyond Skv (only possible in the last tile)
        const int kbase = t * KV_TILE + 4 * hh;
        if constexpr (tail || CAUSAL) {
            const int klim = CAUSAL ? min(a.Skv, q0 + l31 + 1) : a.Skv;  // this lane's query sees keys < klim
#pragma unroll
            for (int kbk = 0; kbk < NB; ++kbk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kbase + kbk * 32 + (e & 3) + 8 * (e >> 2);
                    if (key >= klim) s[kbk][e] = -INFINITY;
                }
        }
        float mx = -INFINITY;  // raw-score maximum of this lane's 32 keys, then of the whole row
#pragma unroll
        for (int kbk = 0; kbk < NB; ++kbk)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kbk][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2e;
        // deferred rescale: keep the old running max while no row of the wave grew by more than 2^THR; P then
        // stays <= 2^THR (exact in fp32, a bf16 float afterwards) and the O / l rescale pass is skipped.
        constexpr float THR = 6.0f;
        if (!__all(mx - m_run <= THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        }
        float rs = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < NB; ++kbk)
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p[j] = __builtin_amdgcn_exp2f(fmaf(s[kbk][8 * sI + j], a.scale_log2e, -m_run));
                    rs += p[j];
                }
                pf[kbk][sI] = __builtin_bit_cast(bf16x8, pack8(p));
            }
        l_run += rs;

        // O^T[d][q] += V^T[d][key] * P^T[key][q].  A lane's transposed-read address = stage base + a lane constant per d
        // block (row 4 hh + tq, swizzled chunk: the swizzle bit is bit 1 of tq, which the block offsets never touch) + the
        // block's compile-time offset (kbk, sI, second four keys), so the sixteen reads carry immediate offsets only.
        const char* vd[2] = {vb + vlane[0], vb + vlane[1]};
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int kbk = 0; kbk < NB; ++kbk)
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const s16x4 lo = ds_read_tr16(vd[d] + kbk * 4096 + sI * 2048);
                    const s16x4 hi = ds_read_tr16(vd[d] + kbk * 4096 + sI * 2048 + 1024);
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[kbk][sI], o[d],
                                                                  0, 0, 0);
                }
        if (more) store_stage((t + 1) & 1);
        __syncthreads();
    };
    // only the last tile can hold keys beyond Skv: the masking code exists in that instance alone
    const bool ragged = (kv_end % KV_TILE) != 0;
    using two = std::integral_constant<int, 2>;
    using one = std::integral_constant<int, 1>;
    for (int t = 0; t < ntiles - 1; ++t) tile_body(t, std::false_type{}, two{});
    // a tail of <= 32 keys (77 prompt tokens = 64 + 13) skips its empty second block: scores, softmax and P.V alike
    if (ragged && kv_end - (ntiles - 1) * KV_TILE <= 32) tile_body(ntiles - 1, std::true_type{}, one{});
    else if (ragged) tile_body(ntiles - 1, std::true_type{}, two{});
    else tile_body(ntiles - 1, std::false_type{}, two{});

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + l31;
    if (a.lse && hh == 0 && qrow < a.Sq) a.lse[((long long)b * a.H + h) * a.Sq + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
    if (qrow < a.Sq) {
        __bf16* op = a.O + ((long long)b * a.Sq + qrow) * a.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 v;
                v[0] = pack_bf16x2(o[d][4 * q] * inv, o[d][4 * q + 1] * inv);
                v[1] = pack_bf16x2(o[d][4 * q + 2] * inv, o[d][4 * q + 3] * inv);
                *(u32x2*)(op + d * 32 + 8 * q + 4 * hh) = v;
            }
    }
}

// ------------------------------------------------------------------ long-key instance: 64 queries per wave
// The same arithmetic as attn_d64_kernel for the self-attention shapes (Sq % 64 == 0, Skv % 64 == 0, no mask), arranged so
// that a wave's LDS traffic per MFMA halves and nothing on the tile's critical path waits on LDS round trips:
//   * a wave owns TWO 32-query blocks (A: q0 + l31, B: q0 + 32 + l31) at 2 waves per SIMD (256 registers): every K
//     fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16 pair) feeds two MFMAs, and the two blocks'
//     softmax chains are independent, so one block's VALU work sits beside the other block's MFMAs;
//   * K / V tiles arrive by LDS-DMA (buffer_load ... lds, four 1-KiB pieces per wave and tile: no staging registers, no
//     ds_write, no address arithmetic in the loop) into a 3-stage ring, issued two tiles ahead; one counted
//     s_waitcnt vmcnt(4) + raw s_barrier per tile;
//   * the lane^32 exchange of the row maximum / row sum is v_permlane32_swap (VALU) instead of ds_bpermute;
//   * O leaves as 16-byte stores (permlane32 swap of the packed quads: a lane then holds 8 consecutive channels).
// At head dim 64 the tile is bound by vector issue, not by the matrix pipe: per wave and 64-key tile 64 exp + 64 fma + 64 add
// + 32 max3 + 32 cvt ~ 1500 issue cycles against 1024 cycles of MFMA (SQ counters: VALU active 0.40 of wave cycles at two
// waves per SIMD, matrix pipe busy 0.51), so the two waves of a SIMD overlap each other's phases and that is most of what
// there is to get.  Measured against this form and dropped (profiles/r02_attn_variants.txt): an 8-wave block whose two groups
// are held half a tile apart by barriers (one wave of a SIMD in its MFMAs while the other is in its softmax: -12 %), and a
// software pipeline over 32-key blocks inside the wave (S(j+1) and P.V(j-1) beside the softmax of block j, all LDS reads by
// inline asm with counted waits: -3 %: twice the maxima / branches per tile).
__device__ __forceinline__ void attn_dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff,
                                             (int)soff, 0, 0);
}
__device__ __forceinline__ float other_half_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ unsigned lds_off(const char* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// two transposed 8-byte LDS reads (rows r and r + 8 of a V tile) = one bf16x8 A-operand fragment; the caller waits
template <int OFF>
__device__ __forceinline__ u32x4 tr_read_pair(unsigned addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 1024));
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// PRE: Q already carries scale * log2(e) (folded into the query projection's weights when they are packed), and the scores
// leave the matrix core already relative to the row's reference maximum: one more MFMA per score block multiplies a column of
// ones on the K side with a row of -m on the Q side (k slot 0 of an extra 16-deep step), with m kept bf16-representable so
// that the product is exact.  The exponential's argument then needs no fma: 64 of the tile's ~315 vector instructions go.
template <bool PRE>
__global__ __launch_bounds__(AT_THREADS, 2) void attn_d64_w64_kernel(const AttnArgs a) {
    constexpr int NSTG = 3, STG = 2 * TILE_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STG];  // per stage: K tile, V tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    // XCD-aware order: blocks go to the 8 XCDs round-robin by their linear index, so the query blocks of one (image, head)
    // would land on all eight and each L2 would fetch that head's K / V; instead XCD x walks the heads bh = x, x + 8, ...,
    // all query blocks of a head in a row (needs a head count divisible by 8; otherwise the plain order)
    int b = blockIdx.z, h = blockIdx.y, qb = blockIdx.x;
#ifndef ATTN_NO_XCD
    {
        const int QB = gridDim.x, HB = gridDim.y * gridDim.z;
        if ((HB & 7) == 0) {
            const int L = blockIdx.x + QB * (blockIdx.y + gridDim.y * blockIdx.z);
            const int xcd = L & 7, idx = L >> 3, bh = (idx / QB) * 8 + xcd;
            qb = idx % QB;
            h = bh % gridDim.y;
            b = bh / gridDim.y;
        }
    }
#endif
    const int q0 = qb * 256 + wave * 64;
    const bool wok = q0 < a.Sq;  // wave-uniform (Sq % 64 == 0); a wave beyond Sq still stages tiles and meets the barriers

    // Q^T fragments (B operand) of the two query blocks
    bf16x8 qf[2][4];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const int qrow = wok ? q0 + 32 * X + l31 : 0;
        const __bf16* qp = a.Q + ((long long)b * a.Sq + qrow) * a.ldq + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[X][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(qp + ks * 16));
    }
    __builtin_amdgcn_sched_barrier(0);

    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.K + (long long)b * a.Skv * a.ldk + h * 64), 0, a.k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.V + (long long)b * a.Skv * a.ldv + h * 64), 0, a.v_bytes, 0x00020000);
    // DMA pieces: a tile is 8 K pieces + 8 V pieces of 8 rows x 128 B; wave w brings pieces 2w, 2w + 1 of each.  Lane l of a
    // piece lands at chunk position l & 7 of row l >> 3 and fetches the chunk that the swizzled read expects there.
    unsigned kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 8 + (lane >> 3), c = lane & 7;
        kvo[i] = (unsigned)(r * a.ldk + ((c ^ ((r >> 1) & 7)) << 3)) * 2u;
        vvo[i] = (unsigned)(r * a.ldv + ((c ^ (((r >> 1) & 1) << 2)) << 3)) * 2u;
    }
    const unsigned ktile_b = (unsigned)(KV_TILE * a.ldk) * 2u, vtile_b = (unsigned)(KV_TILE * a.ldv) * 2u;
    auto issue_tile = [&](int t, int s) {
        char* kb = smem + s * STG + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) attn_dma16(rk, kb + i * 1024, kvo[i], (unsigned)t * ktile_b);
#pragma unroll
        for (int i = 0; i < 2; ++i) attn_dma16(rv, kb + TILE_BYTES + i * 1024, vvo[i], (unsigned)t * vtile_b);
    };

    f32x16 o[2][2];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[X][d][e] = 0.f;
    // (PRE: the reference starts at 0 and the first tile always takes the re-reference path)
    float m_run[2] = {PRE ? 0.f : -1e30f, PRE ? 0.f : -1e30f}, l_run[2] = {0.f, 0.f};
    // PRE: A fragment with 1.0 in k slot 0 of every key row, B fragments with -m of the lane's query in k slot 0
    const u32x4 ones = {hh == 0 ? 0x3f80u : 0u, 0u, 0u, 0u};
    u32x4 qm[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};

    const int kswz = (l31 >> 1) & 7;
    const int g16 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    int vlane[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int chunk = d * 4 + g16 * 2 + (tp >> 1);
        vlane[d] = (4 * hh + tq) * 128 + ((chunk ^ (((tq >> 1) & 1) << 2)) << 4) + (tp & 1) * 8;
    }
    int kread[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kread[ks] = l31 * 128 + (((2 * ks + hh) ^ kswz) << 4);

    const int ntiles = a.Skv / KV_TILE;
    issue_tile(0, 0);
    if (ntiles > 1) {
        issue_tile(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int s = 0, sn = 2;
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 2 < ntiles;
        if (more) issue_tile(t + 2, sn);
        const char* kb = smem + s * STG;
        const char* vb = kb + TILE_BYTES;

        // S^T of both query blocks: a K fragment feeds two MFMAs
        f32x16 sc[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(kb + kbk * 4096 + kread[ks]);
#pragma unroll
            for (int X = 0; X < 2; ++X) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[X][kbk][e] = 0.f;
                if constexpr (PRE)
                    sc[X][kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones),
                                                                        __builtin_bit_cast(bf16x8, qm[X]), sc[X][kbk], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    sc[X][kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[X][ks], sc[X][kbk], 0, 0, 0);
            }
        }
        // V^T fragments of the whole tile, read by inline asm: hipcc puts an s_waitcnt vmcnt(0) in front of the
        // ds_read_tr16_b64 builtin whenever LDS-DMAs are in flight (it cannot tell their target from the read's), which
        // would drain the two-tile prefetch every tile.  Issued here, behind the S MFMAs, and waited for (explicit
        // lgkmcnt) only in front of the P.V MFMAs, a softmax later.
        u32x4 vfr[2][2][2];
        const unsigned va0 = lds_off(vb) + (unsigned)vlane[0], va1 = lds_off(vb) + (unsigned)vlane[1];
#define CCNET_VFRAG(kbk, sI)                                                          \
        vfr[kbk][sI][0] = tr_read_pair<(kbk) * 4096 + (sI) * 2048>(va0);              \
        vfr[kbk][sI][1] = tr_read_pair<(kbk) * 4096 + (sI) * 2048>(va1);
        CCNET_VFRAG(0, 0) CCNET_VFRAG(0, 1) CCNET_VFRAG(1, 0) CCNET_VFRAG(1, 1)
#undef CCNET_VFRAG
        // online softmax per query block (deferred rescale, threshold 2^6 as in attn_d64_kernel)
        float mx[2];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            float m = -INFINITY;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int e = 0; e < 16; ++e) m = fmaxf(m, sc[X][kbk][e]);
            mx[X] = PRE ? other_half_max(m) : other_half_max(m) * a.scale_log2e;
        }
        constexpr float THR = 6.0f;
        if constexpr (PRE) {
            // scores are relative to m_run already; the first tile sets the reference (in either direction), later tiles
            // only raise it.  The new reference is rounded to bf16 (the -m fragment must hold it exactly).
            if (t == 0 || !__all(mx[0] <= THR && mx[1] <= THR)) {
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    const float want = m_run[X] + (t == 0 ? mx[X] : fmaxf(mx[X], 0.f));
                    const unsigned short mb = f32_to_bf16_bits(want);
                    const float m_new = bf16_bits_to_f32(mb);
                    const float delta = m_new - m_run[X];
                    const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                    m_run[X] = m_new;
                    qm[X][0] = hh == 0 ? (unsigned)(mb ^ 0x8000u) : 0u;
                    l_run[X] *= alpha;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[X][d][e] *= alpha;
#pragma unroll
                    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                        for (int e = 0; e < 16; ++e) sc[X][kbk][e] -= delta;
                }
            }
        } else if (!__all(mx[0] - m_run[0] <= THR && mx[1] - m_run[1] <= THR)) {
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const float m_new = fmaxf(m_run[X], mx[X]);
                const float alpha = __builtin_amdgcn_exp2f(m_run[X] - m_new);
                m_run[X] = m_new;
                l_run[X] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[X][d][e] *= alpha;
            }
        }
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            bf16x8 pf[2][2];
            float rs[4];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    float p[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        p[j] = PRE ? __builtin_amdgcn_exp2f(sc[X][kbk][8 * sI + j])
                                   : __builtin_amdgcn_exp2f(fmaf(sc[X][kbk][8 * sI + j], a.scale_log2e, -m_run[X]));
                    rs[2 * kbk + sI] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                    pf[kbk][sI] = __builtin_bit_cast(bf16x8, pack8(p));
                }
            l_run[X] += (rs[0] + rs[1]) + (rs[2] + rs[3]);
            // O^T += V^T . P^T.  The wait names the fragments, so every MFMA below depends on it.
            if (X == 0)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(vfr[0][0][0]), "+v"(vfr[0][0][1]), "+v"(vfr[0][1][0]), "+v"(vfr[0][1][1]),
                               "+v"(vfr[1][0][0]), "+v"(vfr[1][0][1]), "+v"(vfr[1][1][0]), "+v"(vfr[1][1][1]));
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int sI = 0; sI < 2; ++sI)
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        o[X][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfr[kbk][sI][d]),
                                                                         pf[kbk][sI], o[X][d], 0, 0, 0);
        }
        // tile t + 1 has landed (mine: counted wait; everyone's: barrier); tile t + 2 stays in flight
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s = s + 1 == NSTG ? 0 : s + 1;
        sn = sn + 1 == NSTG ? 0 : sn + 1;
    }

    if (!wok) return;
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[X]), __float_as_uint(l_run[X]), false, false);
        const float l_tot = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + 32 * X + l31;
        if (a.lse && hh == 0) a.lse[((long long)b * a.H + h) * a.Sq + qrow] = m_run[X] + __builtin_amdgcn_logf(l_tot);
        __bf16* op = a.O + ((long long)b * a.Sq + qrow) * a.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                // quads 2 qp (channels 16 qp + 4 hh ..) and 2 qp + 1 (16 qp + 8 + 4 hh ..): after the swap the lower
                // lane half holds channels 16 qp .. + 7 of its query, the upper half channels 16 qp + 8 .. + 15
                u32x4 v;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const unsigned e0 = pack_bf16x2(o[X][d][8 * qp + 2 * w] * inv, o[X][d][8 * qp + 2 * w + 1] * inv);
                    const unsigned e1 = pack_bf16x2(o[X][d][8 * qp + 4 + 2 * w] * inv, o[X][d][8 * qp + 4 + 2 * w + 1] * inv);
                    const auto sw = __builtin_amdgcn_permlane32_swap(e0, e1, false, false);
                    v[w] = sw[0];
                    v[2 + w] = sw[1];
                }
                *(u32x4*)(op + d * 32 + 16 * qp + 8 * hh) = v;
            }
    }
}

// ------------------------------------------------------------------ attention backward (training), head dim 64
// Two passes, each a re-arrangement of the forward kernel, both recomputing P from Q, K and the forward's LSE (no N x N
// tensor in memory, no atomics: every gradient element is written by exactly one lane, so the step is bit-reproducible):
//   * attn_bwd_dq_kernel  -- the query on the lane, exactly the forward's layout: S^T = K Q^T, dP^T = V dO^T,
//     dS^T = scale P^T o (dP^T - delta_q) is directly the B operand of dQ^T += K^T dS^T (K^T by transposed LDS reads).
//   * attn_bwd_dkv_kernel -- the key on the lane (a wave owns 32 keys, the block walks the queries): S = Q K^T, dP = dO V^T
//     with K^T / V^T fragments held in registers; P and dS (bf16) are directly the B operands of dV^T += dO^T P and
//     dK^T += Q^T dS (Q^T, dO^T by transposed LDS reads); LSE and delta of the tile's 64 queries ride in LDS.
// delta_q = sum_d dO[q][d] O[q][d] (attn_delta_kernel).  Tiles that are read both by rows and transposed are staged twice,
// once per swizzle (the forward's K image and V image).

// delta[b][h][q] = sum over the head's 64 channels of dO * O: a wave per (row, head)
__global__ __launch_bounds__(256) void attn_delta_kernel(const __bf16* O, int ldo, const __bf16* dO, int lddo, int rows, int H, int Sq,
                                                         float* delta) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long long)rows * H) return;
    const int row = (int)(item / H), h = (int)(item % H);
    const float v = bf16_bits_to_f32(*(const unsigned short*)(O + (long long)row * ldo + h * 64 + lane)) *
                    bf16_bits_to_f32(*(const unsigned short*)(dO + (long long)row * lddo + h * 64 + lane));
    const float s = wave_sum(v);
    if (lane == 0) delta[((long long)(row / Sq) * H + h) * Sq + (row % Sq)] = s;
}

__global__ __launch_bounds__(AT_THREADS, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE_BYTES];  // per stage: K (row image), K (transposed-read image), V
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qrow = q0 + l31;
    const bool qok = qrow < a.Sq;
    bf16x8 qf[4], dof[4];
    {
        const __bf16* qp = a.Q + ((long long)b * a.Sq + (qok ? qrow : 0)) * a.ldq + h * 64 + hh * 8;
        const __bf16* dp = a.dO + ((long long)b * a.Sq + (qok ? qrow : 0)) * a.lddo + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 v = *(const u32x4*)(qp + ks * 16), w = *(const u32x4*)(dp + ks * 16);
            if (!qok) { v = u32x4{0, 0, 0, 0}; w = u32x4{0, 0, 0, 0}; }
            qf[ks] = __builtin_bit_cast(bf16x8, v);
            dof[ks] = __builtin_bit_cast(bf16x8, w);
        }
    }
    const long long sidx = ((long long)b * a.H + h) * a.Sq + (qok ? qrow : 0);
    const float lse = qok ? a.lse[sidx] : 1e30f, dlt = qok ? a.delta[sidx] : 0.f;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.K + (long long)b * a.Skv * a.ldk + h * 64), 0, a.k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.V + (long long)b * a.Skv * a.ldv + h * 64), 0, a.v_bytes, 0x00020000);
    const int lrow = tid >> 3, lchunk = tid & 7;
    u32x4 kr[2], vr[2];
    auto issue_loads = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = tile * KV_TILE + lrow + i * 32;
            const bool ok = key < a.Skv;
            const unsigned ko = ok ? (unsigned)(key * a.ldk + lchunk * 8) * 2u : 0x80000000u;
            const unsigned vo = ok ? (unsigned)(key * a.ldv + lchunk * 8) * 2u : 0x80000000u;
            kr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)ko, 0, 0));
            vr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)vo, 0, 0));
        }
    };
    auto store_stage = [&](int s) {
        char* kb = smem + s * 3 * TILE_BYTES;
        char* kt = kb + TILE_BYTES;
        char* vb = kt + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lrow + i * 32;
            *(u32x4*)(kb + r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4)) = kr[i];
            *(u32x4*)(kt + r * 128 + ((lchunk ^ (((r >> 1) & 1) << 2)) << 4)) = kr[i];
            *(u32x4*)(vb + r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4)) = vr[i];
        }
    };
    f32x16 dq[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;
    const int kswz = (l31 >> 1) & 7;
    const int g16 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const int ntiles = (a.Skv + KV_TILE - 1) / KV_TILE;
    issue_loads(0);
    store_stage(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) issue_loads(t + 1);
        const char* kb = smem + (t & 1) * 3 * TILE_BYTES;
        const char* kt = kb + TILE_BYTES;
        const char* vb = kt + TILE_BYTES;
        bf16x8 dsf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = (kbk * 32 + l31) * 128 + (((2 * ks + hh) ^ kswz) << 4);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kb + off), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(vb + off), dof[ks], dp, 0, 0, 0);
            }
            const int kbase = t * KV_TILE + kbk * 32 + 4 * hh;
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                float ds[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 8 * sI + j;
                    const int key = kbase + (e & 3) + 8 * (e >> 2);
                    const float p = key < a.Skv ? __builtin_amdgcn_exp2f(fmaf(s[e], a.scale_log2e, -lse)) : 0.f;
                    ds[j] = a.scale * p * (dp[e] - dlt);
                }
                dsf[kbk][sI] = __builtin_bit_cast(bf16x8, pack8(ds));
            }
        }
        // dQ^T[d][q] += K^T[d][key] * dS^T[key][q]
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const int key0 = kbk * 32 + 16 * sI + 4 * hh;
                    const int chunk = (d * 32 + g16 * 16 + tp * 4) >> 3;
                    const int within = ((d * 32 + g16 * 16 + tp * 4) & 7) * 2;
                    const int r0 = key0 + tq, r1 = key0 + 8 + tq;
                    const s16x4 lo = ds_read_tr16(kt + r0 * 128 + ((chunk ^ (((r0 >> 1) & 1) << 2)) << 4) + within);
                    const s16x4 hi = ds_read_tr16(kt + r1 * 128 + ((chunk ^ (((r1 >> 1) & 1) << 2)) << 4) + within);
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), dsf[kbk][sI], dq[d], 0, 0, 0);
                }
        if (more) store_stage((t + 1) & 1);
        __syncthreads();
    }
    if (qok) {
        __bf16* op = a.dQ + ((long long)b * a.Sq + qrow) * a.lddq + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 v;
                v[0] = pack_bf16x2(dq[d][4 * q], dq[d][4 * q + 1]);
                v[1] = pack_bf16x2(dq[d][4 * q + 2], dq[d][4 * q + 3]);
                *(u32x2*)(op + d * 32 + 8 * q + 4 * hh) = v;
            }
    }
}

__global__ __launch_bounds__(AT_THREADS, 2) void attn_bwd_dkv_kernel(const AttnArgs a) {
    // per stage: Q (row image), Q (transposed-read image), dO (row), dO (transposed), then LSE and delta of the tile's 64 queries
    constexpr int STAGE_B = 4 * TILE_BYTES + 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int krow = k0 + l31;
    const bool kok = krow < a.Skv;
    bf16x8 kf[4], vf[4];  // K^T / V^T fragments (B operands): lane holds K[krow][16 ks + 8 hh .. +7]
    {
        const __bf16* kp = a.K + ((long long)b * a.Skv + (kok ? krow : 0)) * a.ldk + h * 64 + hh * 8;
        const __bf16* vp = a.V + ((long long)b * a.Skv + (kok ? krow : 0)) * a.ldv + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 v = *(const u32x4*)(kp + ks * 16), w = *(const u32x4*)(vp + ks * 16);
            if (!kok) { v = u32x4{0, 0, 0, 0}; w = u32x4{0, 0, 0, 0}; }
            kf[ks] = __builtin_bit_cast(bf16x8, v);
            vf[ks] = __builtin_bit_cast(bf16x8, w);
        }
    }
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Q + (long long)b * a.Sq * a.ldq + h * 64), 0, a.k_bytes, 0x00020000);   // (k_bytes / v_bytes: Q / dO extents here)
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.dO + (long long)b * a.Sq * a.lddo + h * 64), 0, a.v_bytes, 0x00020000);
    const float* lse_p = a.lse + ((long long)b * a.H + h) * a.Sq;
    const float* dlt_p = a.delta + ((long long)b * a.H + h) * a.Sq;
    const int lrow = tid >> 3, lchunk = tid & 7;
    u32x4 qr[2], dr[2];
    float st_l = 0.f, st_d = 0.f;
    auto issue_loads = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tile * KV_TILE + lrow + i * 32;
            const bool ok = q < a.Sq;
            const unsigned qo = ok ? (unsigned)(q * a.ldq + lchunk * 8) * 2u : 0x80000000u;
            const unsigned dofs = ok ? (unsigned)(q * a.lddo + lchunk * 8) * 2u : 0x80000000u;
            qr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, (int)qo, 0, 0));
            dr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)dofs, 0, 0));
        }
        if (tid < 64) {  // a query beyond Sq gets P = exp2(-1e30) = 0
            const int q = tile * KV_TILE + tid;
            st_l = q < a.Sq ? lse_p[q] : 1e30f;
            st_d = q < a.Sq ? dlt_p[q] : 0.f;
        }
    };
    auto store_stage = [&](int s) {
        char* qb = smem + s * STAGE_B;
        char* qt = qb + TILE_BYTES;
        char* db = qt + TILE_BYTES;
        char* dt = db + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lrow + i * 32;
            const int orow = r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4), otr = r * 128 + ((lchunk ^ (((r >> 1) & 1) << 2)) << 4);
            *(u32x4*)(qb + orow) = qr[i];
            *(u32x4*)(qt + otr) = qr[i];
            *(u32x4*)(db + orow) = dr[i];
            *(u32x4*)(dt + otr) = dr[i];
        }
        if (tid < 64) {
            float* f = (float*)(dt + TILE_BYTES);
            f[tid] = st_l;
            f[64 + tid] = st_d;
        }
    };
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }
    const int swz = (l31 >> 1) & 7;
    const int g16 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const int ntiles = (a.Sq + KV_TILE - 1) / KV_TILE;
    issue_loads(0);
    store_stage(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) issue_loads(t + 1);
        const char* qb = smem + (t & 1) * STAGE_B;
        const char* qt = qb + TILE_BYTES;
        const char* db = qt + TILE_BYTES;
        const char* dt = db + TILE_BYTES;
        const float* stat = (const float*)(dt + TILE_BYTES);
#pragma unroll
        for (int qbk = 0; qbk < 2; ++qbk) {
            // S[q][key] and dP[q][key] of 32 queries x this wave's 32 keys: accumulator register e is query
            // qbk * 32 + (e & 3) + 8 (e >> 2) + 4 hh of the tile, the lane is the key
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = (qbk * 32 + l31) * 128 + (((2 * ks + hh) ^ swz) << 4);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(qb + off), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(db + off), vf[ks], dp, 0, 0, 0);
            }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                float p[8], ds[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int qo = qbk * 32 + 16 * sI + 8 * g + 4 * hh;  // 4 consecutive queries: registers 8 sI + 4 g .. + 3
                    const f32x4 l4 = *(const f32x4*)(stat + qo), d4 = *(const f32x4*)(stat + 64 + qo);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 8 * sI + 4 * g + j;
                        p[4 * g + j] = __builtin_amdgcn_exp2f(fmaf(s[e], a.scale_log2e, -l4[j]));
                        ds[4 * g + j] = a.scale * p[4 * g + j] * (dp[e] - d4[j]);
                    }
                }
                pf[sI] = __builtin_bit_cast(bf16x8, pack8(p));
                dsf[sI] = __builtin_bit_cast(bf16x8, pack8(ds));
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const int qq0 = qbk * 32 + 16 * sI + 4 * hh;
                    const int chunk = (d * 32 + g16 * 16 + tp * 4) >> 3;
                    const int within = ((d * 32 + g16 * 16 + tp * 4) & 7) * 2;
                    const int r0 = qq0 + tq, r1 = qq0 + 8 + tq;
                    const int o0 = r0 * 128 + ((chunk ^ (((r0 >> 1) & 1) << 2)) << 4) + within;
                    const int o1 = r1 * 128 + ((chunk ^ (((r1 >> 1) & 1) << 2)) << 4) + within;
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x4 lo = ds_read_tr16(dt + o0), hi = ds_read_tr16(dt + o1);
                    s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[sI], dv[d], 0, 0, 0);
                    lo = ds_read_tr16(qt + o0); hi = ds_read_tr16(qt + o1);
                    vv = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), dsf[sI], dk[d], 0, 0, 0);
                }
        }
        if (more) store_stage((t + 1) & 1);
        __syncthreads();
    }
    if (kok) {
        __bf16* kp = a.dK + ((long long)b * a.Skv + krow) * a.lddk + h * 64;
        __bf16* vp = a.dV + ((long long)b * a.Skv + krow) * a.lddv + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 v;
                v[0] = pack_bf16x2(dk[d][4 * q], dk[d][4 * q + 1]);
                v[1] = pack_bf16x2(dk[d][4 * q + 2], dk[d][4 * q + 3]);
                *(u32x2*)(kp + d * 32 + 8 * q + 4 * hh) = v;
                v[0] = pack_bf16x2(dv[d][4 * q], dv[d][4 * q + 1]);
                v[1] = pack_bf16x2(dv[d][4 * q + 2], dv[d][4 * q + 3]);
                *(u32x2*)(vp + d * 32 + 8 * q + 4 * hh) = v;
            }
    }
}

// ------------------------------------------------------------------ generic-head-dim fallback pieces
// row softmax over `cols` columns; columns [cols, cols_pad) are written as zeros so the probabilities can feed a
// GEMM whose K is padded to a multiple of 64
__global__ __launch_bounds__(256) void softmax_rows_kernel(const __bf16* x, int ldx, int rows, int cols, int cols_pad,
                                                           float scale, __bf16* out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const unsigned short* xr = (const unsigned short*)(x + (long long)row * ldx);
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, bf16_bits_to_f32(xr[c]) * scale);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += __expf(bf16_bits_to_f32(xr[c]) * scale - mx);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    unsigned short* orow = (unsigned short*)(out + (long long)row * ldo);
    for (int c = lane; c < cols; c += 64)
        orow[c] = f32_to_bf16_bits(__expf(bf16_bits_to_f32(xr[c]) * scale - mx) * inv);
    for (int c = cols + lane; c < cols_pad; c += 64) orow[c] = 0;
}

// out[z][c][r] = x[z][r][c] for r < rows, 0 for rows <= r < rows_pad  (V^T of the fallback's P.V product)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const __bf16* x, int ldx, int rows, int cols, long long sx,
                                                            __bf16* out, int ldo, int rows_pad, long long so) {
    __shared__ unsigned short tile[64][65];
    const unsigned short* xp = (const unsigned short*)(x + (long long)blockIdx.z * sx);
    unsigned short* op = (unsigned short*)(out + (long long)blockIdx.z * so);
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? xp[(long long)r * ldx + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) op[(long long)c * ldo + r] = tile[tx][i];
    }
}

}  // namespace

int g_attn_mode = 0;  // tuning (ccnet_set_option key 7): 0 = by shape, 1 = always attn_d64_kernel, 2 = attn_d64_w64_kernel wherever it applies

namespace {
int launch_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B, int H,
                     int Sq, int Skv, float scale, bool causal, void* stream, float* lse = nullptr, bool prescaled = false) {
    if (!Q || !K || !V || !O) return CCNET_EINVAL;
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return CCNET_EINVAL;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 4)) return CCNET_EINVAL;
    if (ldq < H * 64 || ldk < H * 64 || ldv < H * 64 || ldo < H * 64) return CCNET_EINVAL;
    const long long kb = ((long long)Skv - 1) * ldk * 2 + 128, vb = ((long long)Skv - 1) * ldv * 2 + 128;
    if (kb >= 0x7fffffffLL || vb >= 0x7fffffffLL) return CCNET_EUNSUP;
    if (B > 65535 || H > 65535) return CCNET_EUNSUP;
    AttnArgs a{};
    a.Q = (const __bf16*)Q; a.K = (const __bf16*)K; a.V = (const __bf16*)V; a.O = (__bf16*)O;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv;
    a.scale_log2e = prescaled ? 1.0f : scale * 1.4426950408889634f;
    a.k_bytes = (unsigned)kb; a.v_bytes = (unsigned)vb;
    a.lse = lse;
    // 64 queries per wave: the long-key self-attention shapes (whole 64-key tiles, no mask)
    const bool w64_ok = !causal && (Sq % 64) == 0 && (Skv % 64) == 0 && (ldo % 8) == 0;
    const bool w64 = w64_ok && (g_attn_mode == 2 || (g_attn_mode == 0 && Skv >= 256));
    if (causal)
        hipLaunchKernelGGL(attn_d64_kernel<true>, dim3((Sq + 127) / 128, H, B), dim3(AT_THREADS), 0, (hipStream_t)stream, a);
    else if (w64 && prescaled)
        hipLaunchKernelGGL(attn_d64_w64_kernel<true>, dim3((Sq + 255) / 256, H, B), dim3(AT_THREADS), 0, (hipStream_t)stream, a);
    else if (w64)
        hipLaunchKernelGGL(attn_d64_w64_kernel<false>, dim3((Sq + 255) / 256, H, B), dim3(AT_THREADS), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn_d64_kernel<false>, dim3((Sq + 127) / 128, H, B), dim3(AT_THREADS), 0, (hipStream_t)stream, a);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}
}  // namespace

extern "C" int ccnet_attention_d64_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                        void* O, int ldo, int B, int H, int Sq, int Skv, float scale, void* stream) {
    return launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Sq, Skv, scale, false, stream);
}

extern "C" int ccnet_attention_d64_prescaled_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                                  void* O, int ldo, int B, int H, int Sq, int Skv, void* stream) {
    return launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Sq, Skv, 1.0f, false, stream, nullptr, true);
}

extern "C" int ccnet_attention_d64_lse_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                                            int B, int H, int Sq, int Skv, float scale, float* lse, void* stream) {
    if (!lse) return CCNET_EINVAL;
    return launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Sq, Skv, scale, false, stream, lse);
}

extern "C" int ccnet_attention_d64_bwd_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                                            int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                                            void* dK, int lddk, void* dV, int lddv, int B, int H, int Sq, int Skv, float scale,
                                            void* stream) {
    if (!Q || !K || !V || !O || !dO || !lse || !delta) return CCNET_EINVAL;
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || B > 65535 || H > 65535) return CCNET_EINVAL;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (lddo % 8) || (dQ && (lddq % 4)) || (dK && (lddk % 4)) || (dV && (lddv % 4)))
        return CCNET_EINVAL;
    if ((dK == nullptr) != (dV == nullptr)) return CCNET_EINVAL;
    const long long kb = ((long long)Skv - 1) * ldk * 2 + 128, vb = ((long long)Skv - 1) * ldv * 2 + 128;
    const long long qb = ((long long)Sq - 1) * ldq * 2 + 128, db = ((long long)Sq - 1) * lddo * 2 + 128;
    if (kb >= 0x7fffffffLL || vb >= 0x7fffffffLL || qb >= 0x7fffffffLL || db >= 0x7fffffffLL) return CCNET_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const long long items = (long long)B * Sq * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, (const __bf16*)O, ldo,
                       (const __bf16*)dO, lddo, B * Sq, H, Sq, delta);
    CCNET_LAUNCH_CHECK();
    AttnArgs a{};
    a.Q = (const __bf16*)Q; a.K = (const __bf16*)K; a.V = (const __bf16*)V; a.dO = (const __bf16*)dO;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo; a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    a.lse = (float*)lse; a.delta = delta;
    a.dQ = (__bf16*)dQ; a.lddq = lddq; a.dK = (__bf16*)dK; a.lddk = lddk; a.dV = (__bf16*)dV; a.lddv = lddv;
    if (dQ) {
        a.k_bytes = (unsigned)kb; a.v_bytes = (unsigned)vb;
        hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((Sq + 127) / 128, H, B), dim3(AT_THREADS), 0, st, a);
        CCNET_LAUNCH_CHECK();
    }
    if (dK) {
        a.k_bytes = (unsigned)qb; a.v_bytes = (unsigned)db;  // the staged tensors of this pass are Q and dO
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((Skv + 127) / 128, H, B), dim3(AT_THREADS), 0, st, a);
        CCNET_LAUNCH_CHECK();
    }
    return CCNET_OK;
}

extern "C" int ccnet_attention_d64_causal_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                               void* O, int ldo, int B, int H, int S, float scale, void* stream) {
    return launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, S, S, scale, true, stream);
}

extern "C" int ccnet_softmax_rows_bf16(const void* x, int ldx, int rows, int cols, int cols_pad, float scale,
                                       void* out, int ldo, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || cols_pad < cols || ldo < cols_pad) return CCNET_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x, ldx, rows, cols, cols_pad, scale, (__bf16*)out, ldo);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_transpose_pad_bf16(const void* x, int ldx, int rows, int cols, int64_t sx, void* out, int ldo,
                                        int rows_pad, int64_t so, int batch, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || rows_pad < rows || ldo < rows_pad || batch <= 0 || batch > 65535)
        return CCNET_EINVAL;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((rows_pad + 63) / 64, (cols + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)x, ldx, rows, cols, (long long)sx, (__bf16*)out, ldo,
                       rows_pad, (long long)so);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

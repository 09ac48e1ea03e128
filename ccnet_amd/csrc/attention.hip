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

    const int ntiles = (a.Skv + KV_TILE - 1) / KV_TILE;
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

        // O^T[d][q] += V^T[d][key] * P^T[key][q]
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int kbk = 0; kbk < NB; ++kbk)
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const int key0 = kbk * 32 + 16 * sI + 4 * hh;  // first 4 keys; second 4 are +8
                    const int chunk = (d * 32 + g16 * 16 + tp * 4) >> 3;       // 16-B chunk of the d column
                    const int within = ((d * 32 + g16 * 16 + tp * 4) & 7) * 2;  // byte offset inside chunk
                    const int r0 = key0 + tq, r1 = key0 + 8 + tq;
                    const s16x4 lo = ds_read_tr16(vb + r0 * 128 + ((chunk ^ (((r0 >> 1) & 1) << 2)) << 4) + within);
                    const s16x4 hi = ds_read_tr16(vb + r1 * 128 + ((chunk ^ (((r1 >> 1) & 1) << 2)) << 4) + within);
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[kbk][sI], o[d],
                                                                  0, 0, 0);
                }
        if (more) store_stage((t + 1) & 1);
        __syncthreads();
    };
    // only the last tile can hold keys beyond Skv: the masking code exists in that instance alone
    const bool ragged = (a.Skv % KV_TILE) != 0;
    using two = std::integral_constant<int, 2>;
    using one = std::integral_constant<int, 1>;
    for (int t = 0; t < ntiles - 1; ++t) tile_body(t, std::false_type{}, two{});
    // a tail of <= 32 keys (77 prompt tokens = 64 + 13) skips its empty second block: scores, softmax and P.V alike
    if (ragged && a.Skv - (ntiles - 1) * KV_TILE <= 32) tile_body(ntiles - 1, std::true_type{}, one{});
    else if (ragged) tile_body(ntiles - 1, std::true_type{}, two{});
    else tile_body(ntiles - 1, std::false_type{}, two{});

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + l31;
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

namespace {
int launch_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B, int H,
                     int Sq, int Skv, float scale, bool causal, void* stream) {
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
    a.scale_log2e = scale * 1.4426950408889634f;
    a.k_bytes = (unsigned)kb; a.v_bytes = (unsigned)vb;
    if (causal)
        hipLaunchKernelGGL(attn_d64_kernel<true>, dim3((Sq + 127) / 128, H, B), dim3(AT_THREADS), 0, (hipStream_t)stream, a);
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

// K3 GroupNorm(+SiLU) and K6 LayerNorm over NHWC bf16 rows, fp32 statistics.  HBM-bound kernels:
// every access is 16 B per lane (8 bf16 channels) and consecutive lanes read consecutive channel chunks.
//
// GroupNorm, NHWC: a group is C/groups adjacent channels of every pixel of one image.
//   pass 1 (gn_stats): grid (chunks, N); a block streams its pixel range for all channels, each thread
//     owning a fixed 8-channel column so it accumulates per-channel (sum, sumsq) in registers; columns are
//     folded into per-group partials through LDS and written to the workspace [N][chunks][groups][2]
//     as (mean, M2) of that chunk;
//   pass 2 (gn_apply): every block merges the chunk partials of its image with Chan's parallel-variance
//     formula, builds per-channel scale/shift in LDS and streams x*A+B (+SiLU) -> bf16.
//   Two reads + one write per element (6 B vs the 4 B algorithmic minimum); the second read normally hits
//   the 256 MiB Infinity Cache.  A channel concat of two tensors (x1 | x2) is read in place.
#include "common.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAXC = 4096;

struct GnArgs {
    const __bf16* x1; const __bf16* x2; int ldx1, ldx2, C1, C2, C, N, HW, groups, cpg, chunks, ppc;
    float eps; const float* gamma; const float* beta; int silu; __bf16* out; int ldo; float* ws;
};

__device__ __forceinline__ u32x4 gn_load(const GnArgs& a, long long pixrow, int c) {
    // c = first channel of an 8-channel column
    if (c < a.C1) return *(const u32x4*)(a.x1 + pixrow * a.ldx1 + c);
    return *(const u32x4*)(a.x2 + pixrow * a.ldx2 + (c - a.C1));
}

__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const GnArgs a) {
    // per-(pixel-row-slot, channel) partials; rows_par * C <= GN_MAXC by construction.  Reduced in a fixed order
    // (no float atomics) so the statistics, and with them the whole sampler, are bit-reproducible run to run.
    __shared__ float s_sum[GN_MAXC], s_sq[GN_MAXC];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int cols = a.C / 8;
    const int p0 = chunk * a.ppc, p1 = min(a.HW, p0 + a.ppc);
    const int cstride = cols < GN_THREADS ? cols : GN_THREADS;  // columns covered per sweep
    const int rows_par = GN_THREADS / cstride;                   // pixel rows processed in parallel
    const int my_col = tid % cstride, my_row = tid / cstride;
    if (my_row < rows_par) {
        for (int col = my_col; col < cols; col += cstride) {
            float s[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
            const long long rb = (long long)n * a.HW;
            int p = p0 + my_row;
            // four independent 16-B loads in flight per lane: this pass is pure HBM streaming
            for (; p + 3 * rows_par < p1; p += 4 * rows_par) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = gn_load(a, rb + p + u * rows_par, col * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    unpack8(v[u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
                }
            }
            for (; p < p1; p += rows_par) {
                float f[8];
                unpack8(gn_load(a, rb + p, col * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s_sum[my_row * a.C + col * 8 + e] = s[e];
                s_sq[my_row * a.C + col * 8 + e] = q[e];
            }
        }
    }
    __syncthreads();
    if (tid < a.groups) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < rows_par; ++r)
            for (int c = tid * a.cpg; c < (tid + 1) * a.cpg; ++c) { s += s_sum[r * a.C + c]; q += s_sq[r * a.C + c]; }
        const float cnt = (float)((p1 - p0) * a.cpg);
        const float mean = s / cnt;
        float* w = a.ws + (((long long)n * a.chunks + chunk) * a.groups + tid) * 2;
        w[0] = mean;
        w[1] = fmaxf(q - s * mean, 0.f);  // M2 of this chunk
    }
}

__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const GnArgs a) {
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    {
        // Chan merge of the per-chunk (mean, M2) partials: `per` lanes per group take every per-th chunk, then combine
        // pairwise (a serial walk over 64 chunks was 64 dependent L2 round trips, ~20 us with one image in flight)
        int per = 1;
        while (per * 2 * a.groups <= GN_THREADS && per < 64) per *= 2;
        const int g = tid / per, j = tid % per;
        if (g < a.groups) {
            float mean = 0.f, m2 = 0.f, cnt = 0.f;
            for (int k = j; k < a.chunks; k += per) {
                const float* w = a.ws + (((long long)n * a.chunks + k) * a.groups + g) * 2;
                const int q0 = k * a.ppc, q1 = min(a.HW, q0 + a.ppc);
                const float cb = (float)((q1 - q0) * a.cpg);
                const float delta = w[0] - mean, tot = cnt + cb;
                mean += delta * (cb / tot);
                m2 += w[1] + delta * delta * (cnt * cb / tot);
                cnt = tot;
            }
            for (int o = 1; o < per; o *= 2) {
                const float mean_b = __shfl_xor(mean, o), m2_b = __shfl_xor(m2, o), cnt_b = __shfl_xor(cnt, o);
                const float tot = cnt + cnt_b;
                if (tot > 0.f) {
                    // symmetric form: both lanes of a pair compute the same value
                    const float delta = mean_b - mean;
                    mean = (mean * cnt + mean_b * cnt_b) / tot;
                    m2 = m2 + m2_b + delta * delta * (cnt * cnt_b / tot);
                }
                cnt = tot;
            }
            if (j == 0) {
                s_mean[g] = mean;
                s_rstd[g] = rsqrtf(m2 / cnt + a.eps);
            }
        }
    }
    __syncthreads();
    // a thread owns a fixed 8-channel column: its scale / shift live in registers, rows are streamed four at a time
    const int cols = a.C / 8;
    const int p0 = chunk * a.ppc, p1 = min(a.HW, p0 + a.ppc);
    const int cstride = cols < GN_THREADS ? cols : GN_THREADS;
    const int rows_par = GN_THREADS / cstride;
    const int my_col = tid % cstride, my_row = tid / cstride;
    if (my_row >= rows_par) return;
    const long long rb = (long long)n * a.HW;
    for (int col = my_col; col < cols; col += cstride) {
        float A[8], Bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = col * 8 + e, g = c / a.cpg;
            A[e] = s_rstd[g] * a.gamma[c];
            Bv[e] = a.beta[c] - s_mean[g] * A[e];
        }
        auto emit = [&](const u32x4& v, long long row) {
            float f[8];
            unpack8(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = f[e] * A[e] + Bv[e];
                f[e] = a.silu ? silu_f(t) : t;
            }
            *(u32x4*)(a.out + row * a.ldo + col * 8) = pack8(f);
        };
        int p = p0 + my_row;
        for (; p + 3 * rows_par < p1; p += 4 * rows_par) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gn_load(a, rb + p + u * rows_par, col * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) emit(v[u], rb + p + u * rows_par);
        }
        for (; p < p1; p += rows_par) emit(gn_load(a, rb + p, col * 8), rb + p);
    }
}

void gn_plan(int N, int HW, int C, int& chunks, int& ppc) {
    // aim for >= ~1024 blocks overall but at least 16 pixels (and ~32 KB) per block
    int want = (1024 + N - 1) / N;
    int min_ppc = 16;
    while ((long long)min_ppc * C * 2 < 32768 && min_ppc < HW) min_ppc *= 2;
    int maxchunks = (HW + min_ppc - 1) / min_ppc;
    chunks = want < maxchunks ? want : maxchunks;
    if (chunks < 1) chunks = 1;
    ppc = (HW + chunks - 1) / chunks;
    chunks = (HW + ppc - 1) / ppc;
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
constexpr int LN_MAXCH = 4;  // 8-channel chunks per lane -> C <= 2048

// NCH = 8-channel chunks per lane, RW = rows per wave: all RW*NCH 16-B loads of a wave are issued before the first
// use, so a wave keeps RW*NCH KiB in flight (this pass is pure HBM streaming).
template <int NCH, int RW>
__global__ __launch_bounds__(256) void layernorm_kernel(const __bf16* x, int ldx, int rows, int C, float eps,
                                                        const float* gamma, const float* beta, __bf16* out,
                                                        int ldo) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= rows) return;
    const int cols = C / 8;
    u32x4 raw[RW][NCH];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int col = lane + k * 64;
            const int row = min(row0 + r, rows - 1);
            raw[r][k] = col < cols ? *(const u32x4*)(x + (long long)row * ldx + col * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    const float invC = 1.0f / (float)C;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float f[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            unpack8(raw[r][k], f[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[k][e];  // zero-filled beyond C
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (lane + k * 64 < cols) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        if (row0 + r < rows) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int col = lane + k * 64;
                if (col < cols) {
                    float o[8];
                    const f32x4 g0 = *(const f32x4*)(gamma + col * 8), g1 = *(const f32x4*)(gamma + col * 8 + 4);
                    const f32x4 b0 = *(const f32x4*)(beta + col * 8), b1 = *(const f32x4*)(beta + col * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (f[k][e] - mean) * rstd * g0[e] + b0[e];
                        o[4 + e] = (f[k][4 + e] - mean) * rstd * g1[e] + b1[e];
                    }
                    *(u32x4*)(out + (long long)(row0 + r) * ldo + col * 8) = pack8(o);
                }
            }
        }
    }
}

// Row statistics only (the LayerNorm whose normalisation is folded into the consuming GEMM, gemm_conv.hip): same loads and
// two-pass mean / variance as layernorm_kernel, 8 bytes written per row instead of the normalised row.
template <int NCH, int RW>
__global__ __launch_bounds__(256) void row_stats_kernel(const __bf16* x, int ldx, int rows, int C, float eps, float* stats) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= rows) return;
    const int cols = C / 8;
    u32x4 raw[RW][NCH];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int col = lane + k * 64;
            const int row = min(row0 + r, rows - 1);
            raw[r][k] = col < cols ? *(const u32x4*)(x + (long long)row * ldx + col * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    const float invC = 1.0f / (float)C;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float f[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            unpack8(raw[r][k], f[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[k][e];
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (lane + k * 64 < cols) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        if (lane == 0 && row0 + r < rows) *(f32x2*)(stats + 2ll * (row0 + r)) = f32x2{mean, rstd};
    }
}

template <int NCH, int RW>
void launch_row_stats(const __bf16* x, int ldx, int rows, int C, float eps, float* stats, hipStream_t st) {
    hipLaunchKernelGGL((row_stats_kernel<NCH, RW>), dim3((rows + 4 * RW - 1) / (4 * RW)), dim3(256), 0, st, x, ldx, rows, C,
                       eps, stats);
}

template <int NCH, int RW>
void launch_layernorm(const __bf16* x, int ldx, int rows, int C, float eps, const float* gamma, const float* beta,
                      __bf16* out, int ldo, hipStream_t st) {
    hipLaunchKernelGGL((layernorm_kernel<NCH, RW>), dim3((rows + 4 * RW - 1) / (4 * RW)), dim3(256), 0, st, x, ldx, rows,
                       C, eps, gamma, beta, out, ldo);
}

}  // namespace

extern "C" size_t ccnet_groupnorm_workspace_bytes(int N, int HW, int C, int groups) {
    if (N <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
    int chunks, ppc;
    gn_plan(N, HW, C, chunks, ppc);
    return (size_t)N * chunks * groups * 2 * sizeof(float);
}

// One-pass GroupNorm for the deeper levels: one block per (image, group) keeps its whole slice (HW pixels x cpg
// channels, <= GN1_MAXV 16-B vectors per thread) in registers -- one read and one write of the tensor (4 B/element)
// and one launch instead of the 6 B/element and two launches of the stats + apply pair.  Needs cpg % 8 == 0 (the
// slice of a pixel is whole 16-B vectors) and the block size a multiple of the vectors per pixel, so a thread always
// holds the same 8 channels (its gamma / beta live in registers).  The 32 groups of an image share every 128-B line
// of it, so consecutive blocks of an XCD (blockIdx % 8) walk the groups of ONE image: the lines come from HBM once
// per XCD.  Statistics are the two-pass mean / centred M2 over the registers, reduced in a fixed order.
// VW = dwords per vector (4 / 2 / 1: channels per group a multiple of 8 / 4 / 2), MAXV vectors per thread.
template <int VW, int MAXV>
__global__ __launch_bounds__(1024) void gn_onepass_kernel(const GnArgs a, int vpp) {
    constexpr int CH = 2 * VW;  // channels per vector
    __shared__ float s_red[32];
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    // XCD-aware (image, group) assignment: XCD x takes the images n with n % 8 == x, group-fastest
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int n, g;
    if ((gridDim.x & 7) == 0 && (a.N & 7) == 0) {
        n = (j / a.groups) * 8 + xcd; g = j % a.groups;
    } else {
        n = blockIdx.x / a.groups; g = blockIdx.x % a.groups;
    }
    const int c0 = g * a.cpg;
    const __bf16* src; int ld;
    if (c0 < a.C1) { src = a.x1 + (long long)n * a.HW * a.ldx1 + c0; ld = a.ldx1; }
    else { src = a.x2 + (long long)n * a.HW * a.ldx2 + (c0 - a.C1); ld = a.ldx2; }
    const int nvec = a.HW * vpp;
    const int jv = tid % vpp;  // this thread's channel column inside the group (T % vpp == 0)
    const int p0 = tid / vpp, pstep = T / vpp;  // vector k of this thread belongs to pixel p0 + k * pstep
    typedef __attribute__((ext_vector_type(VW))) unsigned int vec_t;
    vec_t v[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        vec_t z;
#pragma unroll
        for (int e = 0; e < VW; ++e) z[e] = 0u;
        v[k] = tid + k * T < nvec ? *(const vec_t*)(src + (long long)(p0 + k * pstep) * ld + jv * CH) : z;
    }
    auto block_sum = [&](float x) {
        x = wave_sum(x);
        __syncthreads();
        if (lane == 0) s_red[wave] = x;
        __syncthreads();
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += s_red[w];  // fixed order
        return t;
    };
    auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
    auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) s += lo(v[k][e]) + hi(v[k][e]);  // (vectors beyond nvec are zeros)
    const float cnt = (float)a.HW * (float)a.cpg;
    const float mean = block_sum(s) / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (tid + k * T < nvec) {
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float d0 = lo(v[k][e]) - mean, d1 = hi(v[k][e]) - mean;
                m2 += d0 * d0 + d1 * d1;
            }
        }
    }
    const float rstd = rsqrtf(block_sum(m2) / cnt + a.eps);
    float ga[CH], be[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        ga[e] = a.gamma[c0 + jv * CH + e] * rstd;
        be[e] = a.beta[c0 + jv * CH + e] - mean * ga[e];
    }
    __bf16* dst = a.out + (long long)n * a.HW * a.ldo + c0 + jv * CH;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (tid + k * T < nvec) {
            vec_t o;
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                float f0 = fmaf(lo(v[k][e]), ga[2 * e], be[2 * e]), f1 = fmaf(hi(v[k][e]), ga[2 * e + 1], be[2 * e + 1]);
                if (a.silu) { f0 = silu_f(f0); f1 = silu_f(f1); }
                o[e] = pack_bf16x2(f0, f1);
            }
            *(vec_t*)(dst + (long long)(p0 + k * pstep) * a.ldo) = o;
        }
    }
}

template <int VW, int MAXV>
bool launch_gn_onepass(const GnArgs& a, hipStream_t st) {
    const int vpp = a.cpg / (2 * VW);
    int unit = vpp;  // block size: a multiple of 64 and of the vectors per pixel
    while (unit % 64) unit += vpp;
    if (unit > 1024) return false;
    const long long nvec = (long long)a.HW * vpp;
    int T = unit;
    while (T + unit <= 1024 && (long long)T * MAXV < nvec) T += unit;
    if ((long long)T * MAXV < nvec) return false;
    hipLaunchKernelGGL((gn_onepass_kernel<VW, MAXV>), dim3(a.N * a.groups), dim3(T), 0, st, a, vpp);
    return true;
}

// One-pass GroupNorm where a group is not a whole number of 16-byte vectors (10, 20, 30 channels per group: the 64x64 and
// 32x32 levels of the 320 / 640 / 960-channel tensors): a block takes GQ adjacent groups whose channels together are whole
// vectors (GQ * cpg % 8 == 0) and keeps that slice of the image (HW pixels x vpp vectors) in registers.  A thread always holds
// the same vector column, whose 8 channels fall into at most two groups (cpg >= 8): it accumulates one partial per group,
// the block reduces them in a fixed order (wave g sums group g's partials), two-pass mean / centred M2 as the other paths.
// One read + one write of the tensor instead of the two reads + one write (and two launches) of the stats + apply pair.
template <int MAXV>
__global__ __launch_bounds__(1024) void gn_onepass_multi_kernel(const GnArgs a, int vpp, int GQ) {
    __shared__ float s_part[1024][2];
    __shared__ float s_grp[16];
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int bpi = a.groups / GQ;  // blocks per image
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int n, gq;
    if ((gridDim.x & 7) == 0 && (a.N & 7) == 0) { n = (j / bpi) * 8 + xcd; gq = j % bpi; }
    else { n = blockIdx.x / bpi; gq = blockIdx.x % bpi; }
    const int c0 = gq * GQ * a.cpg;  // first channel of this block's slice (a multiple of 8; the slice lies in one source)
    const __bf16* src; int ld;
    if (c0 < a.C1) { src = a.x1 + (long long)n * a.HW * a.ldx1 + c0; ld = a.ldx1; }
    else { src = a.x2 + (long long)n * a.HW * a.ldx2 + (c0 - a.C1); ld = a.ldx2; }
    const int nvec = a.HW * vpp;
    const int jv = tid % vpp, p0 = tid / vpp, pstep = T / vpp;
    // the vector's channels 8 jv .. 8 jv + 7 of the slice: the first `split` belong to local group gA, the rest to gA + 1
    const int gA = (8 * jv) / a.cpg;
    const int split = min(8, (gA + 1) * a.cpg - 8 * jv);
    // one address register for all MAXV loads: per-thread byte offset in a VGPR, the per-vector stride in an SGPR (the slice
    // itself fills the register file; 64-bit addresses per vector would not fit beside it)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(((long long)a.HW - 1) * ld * 2 + vpp * 16), 0x00020000);
    const unsigned voff = (unsigned)(p0 * ld + jv * 8) * 2u;
    const unsigned kstride = (unsigned)(pstep * ld) * 2u;
    u32x4 v[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k)  // (vectors beyond the slice fall outside the descriptor and read as zeros)
        v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tid + k * T < nvec ? voff : 0x80000000u),
                                                                               (int)(k * kstride), 0));
    // fixed-order block reduction of the per-thread (group gA, group gA + 1) partials: wave g < GQ sums group g
    auto group_sums = [&](float pa, float pb) {
        __syncthreads();
        s_part[tid][0] = pa; s_part[tid][1] = pb;
        __syncthreads();
        if (wave < GQ) {
            float t = 0.f;
            for (int u = lane; u < T; u += 64) {
                const int ga = (8 * (u % vpp)) / a.cpg;
                t += ga == wave ? s_part[u][0] : (ga + 1 == wave ? s_part[u][1] : 0.f);
            }
            t = wave_sum(t);
            if (lane == 0) s_grp[wave] = t;
        }
        __syncthreads();
    };
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        float f[8];
        unpack8(v[k], f);  // (vectors beyond nvec are zeros)
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa += e < split ? f[e] : 0.f; sb += e < split ? 0.f : f[e]; }
        __builtin_amdgcn_sched_barrier(0);  // one vector unpacked at a time
    }
    group_sums(sa, sb);
    // (the packed slice is made opaque between the passes: hipcc otherwise keeps every vector's eight unpacked floats alive
    // from one pass to the next, 176 registers instead of 88, and spills)
    auto opaque = [&]() {
#pragma unroll
        for (int k = 0; k < MAXV; ++k) asm volatile("" : "+v"(v[k]));
    };
    opaque();
    const float cnt = (float)a.HW * (float)a.cpg;
    const float meanA = s_grp[gA] / cnt, meanB = gA + 1 < GQ ? s_grp[gA + 1] / cnt : 0.f;
    sa = 0.f; sb = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (tid + k * T < nvec) {
            float f[8];
            unpack8(v[k], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = f[e] - (e < split ? meanA : meanB);
                sa += e < split ? d * d : 0.f;
                sb += e < split ? 0.f : d * d;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    group_sums(sa, sb);
    opaque();
    const float rstdA = rsqrtf(s_grp[gA] / cnt + a.eps), rstdB = gA + 1 < GQ ? rsqrtf(s_grp[gA + 1] / cnt + a.eps) : 0.f;
    // per-channel scale / shift of the slice go through LDS (the registers hold the slice itself): threads of pixel lane 0
    // write their vector column's 8 channels, every thread re-reads its column's two float4 pairs per vector (broadcast reads)
    __shared__ __attribute__((aligned(16))) float s_ga[128], s_be[128];
    if (p0 == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + jv * 8 + e;
            const float g_ = a.gamma[c] * (e < split ? rstdA : rstdB);
            s_ga[jv * 8 + e] = g_;
            s_be[jv * 8 + e] = a.beta[c] - (e < split ? meanA : meanB) * g_;
        }
    }
    __syncthreads();
    __bf16* dst = a.out + (long long)n * a.HW * a.ldo + c0;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (unsigned)(((long long)a.HW - 1) * a.ldo * 2 + vpp * 16), 0x00020000);
    const unsigned doff = (unsigned)(p0 * a.ldo + jv * 8) * 2u;
    const unsigned dstride = (unsigned)(pstep * a.ldo) * 2u;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        {
            float f[8];
            unpack8(v[k], f);
            const f32x4 g0 = *(const f32x4*)(s_ga + jv * 8), g1 = *(const f32x4*)(s_ga + jv * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(s_be + jv * 8), b1 = *(const f32x4*)(s_be + jv * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = fmaf(f[e], g0[e], b0[e]);
                f[4 + e] = fmaf(f[4 + e], g1[e], b1[e]);
            }
            if (a.silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
            }
            // (a vector beyond the slice is stored outside the descriptor, i.e. dropped)
            __builtin_amdgcn_raw_buffer_store_b128(pack8(f), rd, (int)(tid + k * T < nvec ? doff : 0x80000000u), (int)(k * dstride), 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // one vector at a time: the slice fills the register file
    }
}

template <int MAXV>
bool launch_gn_onepass_multi(const GnArgs& a, hipStream_t st) {
    // fewest groups per block whose channels are whole vectors; the slice must lie in one source and fit the registers
    int GQ = 1;
    while (GQ <= 16 && ((GQ * a.cpg) % 8 || (a.groups % GQ))) ++GQ;
    if (GQ > 16 || a.cpg < 8 || GQ * a.cpg > 128) return false;
    if (a.C2 > 0 && (a.C1 % (GQ * a.cpg))) return false;
    const int vpp = GQ * a.cpg / 8;
    int unit = vpp;
    while (unit % 64) unit += vpp;
    if (unit > 1024) return false;
    const long long nvec = (long long)a.HW * vpp;
    int T = unit;
    while (T + unit <= 1024 && (long long)T * MAXV < nvec) T += unit;
    if ((long long)T * MAXV < nvec) return false;
    // measured (tools/ab_gn.sh, 32 images): 67 -> 50 us at 320 channels / 64x64 (256 blocks = one round of the chip), but
    // 92 -> 111 us at 640 channels (512 blocks of 960 threads = two rounds, one block per CU): only when the blocks fill
    // most of ONE round (16 waves of <= 128 registers per CU); the stats + apply pair spreads any size over >= 1024 blocks
    const long long nblocks = (long long)a.N * (a.groups / GQ), slots = 256LL * (1024 / T);
    if (nblocks > slots || nblocks * 4 < slots * 3) return false;
    hipLaunchKernelGGL((gn_onepass_multi_kernel<MAXV>), dim3(a.N * (a.groups / GQ)), dim3(T), 0, st, a, vpp, GQ);
    return true;
}

int g_gn_onepass = 2;  // tuning (ccnet_set_option key 6): 0 = always the two-kernel path, 1 / 2 / 3 = narrowest vector 16 / 8 / 4 B

extern "C" int ccnet_groupnorm_bf16(const void* x1, int ldx1, int C1, const void* x2, int ldx2, int C2, int N,
                                    int HW, int groups, float eps, const float* gamma, const float* beta,
                                    int silu, void* out, int ldo, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    if (!x1 || !out || !gamma || !beta || !workspace) return CCNET_EINVAL;
    if (C2 < 0 || (C2 > 0 && !x2)) return CCNET_EINVAL;
    const int C = C1 + C2;
    if (N <= 0 || HW <= 0 || C1 <= 0 || groups <= 0 || groups > 64 || (C % groups)) return CCNET_EINVAL;
    if ((C1 % 8) || (C2 % 8) || (ldx1 % 8) || (ldo % 8) || (C2 > 0 && (ldx2 % 8))) return CCNET_EINVAL;
    if (C > GN_MAXC) return CCNET_EUNSUP;
    GnArgs a{};
    a.x1 = (const __bf16*)x1; a.x2 = (const __bf16*)x2; a.ldx1 = ldx1; a.ldx2 = ldx2; a.C1 = C1; a.C2 = C2; a.C = C;
    a.N = N; a.HW = HW; a.groups = groups; a.cpg = C / groups; a.eps = eps; a.gamma = gamma; a.beta = beta;
    a.silu = silu; a.out = (__bf16*)out; a.ldo = ldo; a.ws = (float*)workspace;
    gn_plan(N, HW, C, a.chunks, a.ppc);
    if (workspace_bytes < (size_t)N * a.chunks * groups * 2 * sizeof(float)) return CCNET_EWS;
    hipStream_t st = (hipStream_t)stream;
    if (g_gn_onepass && (C1 % a.cpg) == 0 && (long long)N * groups <= 0x7fffffffLL) {
        // widest vector the group width allows; g_gn_onepass = 1: 16-B vectors only, 2: also 8-B, 3: also 4-B
        bool done = false;
        if ((a.cpg % 8) == 0) done = launch_gn_onepass<4, 10>(a, st);
        else if ((a.cpg % 4) == 0 && g_gn_onepass >= 2 && (long long)HW * a.cpg <= 65536)  // (measured: slower beyond)
            done = launch_gn_onepass<2, 22>(a, st);
        else if ((a.cpg % 2) == 0 && g_gn_onepass >= 3) done = launch_gn_onepass<1, 32>(a, st);
        if (!done && (a.cpg % 8) != 0 && g_gn_onepass >= 2 && g_gn_onepass != 4) done = launch_gn_onepass_multi<22>(a, st);
        if (done) {
            CCNET_LAUNCH_CHECK();
            return CCNET_OK;
        }
    }
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.chunks, N), dim3(GN_THREADS), 0, st, a);
    CCNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_apply_kernel, dim3(a.chunks, N), dim3(GN_THREADS), 0, st, a);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_layernorm_bf16(const void* x, int ldx, int rows, int C, float eps, const float* gamma,
                                    const float* beta, void* out, int ldo, void* stream) {
    if (!x || !out || !gamma || !beta) return CCNET_EINVAL;
    if (rows <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldo % 8)) return CCNET_EINVAL;
    if (C > LN_MAXCH * 64 * 8) return CCNET_EUNSUP;
    const int nch = (C / 8 + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    const __bf16* xp = (const __bf16*)x;
    __bf16* op = (__bf16*)out;
    if (nch == 1) launch_layernorm<1, 4>(xp, ldx, rows, C, eps, gamma, beta, op, ldo, st);
    else if (nch == 2) launch_layernorm<2, 4>(xp, ldx, rows, C, eps, gamma, beta, op, ldo, st);
    else if (nch == 3) launch_layernorm<3, 2>(xp, ldx, rows, C, eps, gamma, beta, op, ldo, st);
    else launch_layernorm<4, 2>(xp, ldx, rows, C, eps, gamma, beta, op, ldo, st);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_row_stats_bf16(const void* x, int ldx, int rows, int C, float eps, float* stats, void* stream) {
    if (!x || !stats) return CCNET_EINVAL;
    if (rows <= 0 || C <= 0 || (C % 8) || (ldx % 8)) return CCNET_EINVAL;
    if (C > LN_MAXCH * 64 * 8) return CCNET_EUNSUP;
    const int nch = (C / 8 + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    const __bf16* xp = (const __bf16*)x;
    if (nch == 1) launch_row_stats<1, 4>(xp, ldx, rows, C, eps, stats, st);
    else if (nch == 2) launch_row_stats<2, 4>(xp, ldx, rows, C, eps, stats, st);
    else if (nch == 3) launch_row_stats<3, 2>(xp, ldx, rows, C, eps, stats, st);
    else launch_row_stats<4, 2>(xp, ldx, rows, C, eps, stats, st);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

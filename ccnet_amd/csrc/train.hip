// Backward-pass kernels of the ControlNet training step (reference:
// src/python/training/training_functions.py:888-1043, the autograd of monai-generative's ControlNet / DiffusionModelUNet).
//
// The two big contractions of a layer's backward run on the forward implicit-GEMM kernel (gemm_conv.hip):
//   * input gradient  = the same convolution over dY with the transposed, tap-flipped weights (packed by the host);
//   * weight gradient = dY^T . im2col(X), a GEMM whose contraction runs over the pixels: both operands are brought to
//     [channel][pixel] layout (ccnet_transpose_pad_bf16 for dY, im2col_t_kernel for X) and multiplied by that kernel.
// This file holds what is left: the transposing im2col, bias / time-embedding column sums, the backward of GroupNorm
// (+SiLU), LayerNorm, row softmax, GEGLU and SiLU, and the adjoints of nearest-x2 upsampling and of stride 2.
// All statistics and sums are fp32 and reduced in a fixed order (no float atomics): a training step is bit-reproducible.
#include "common.h"

namespace {

__device__ __forceinline__ float bf(const __bf16* p) { return bf16_bits_to_f32(*(const unsigned short*)p); }
__device__ __forceinline__ void st_bf(__bf16* p, float v) { *(unsigned short*)p = f32_to_bf16_bits(v); }
__device__ __forceinline__ float dsilu(float t) {  // d/dt [t sigmoid(t)]
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
    return s * (1.0f + t * (1.0f - s));
}

// block-wide sum in a fixed order; every thread gets the result.  `red` holds blockDim.x / 64 + 1 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// ------------------------------------------------------------------ transposing im2col (operand of the weight gradient)
struct Im2colArgs {
    const __bf16* x1; const __bf16* x2; int ldx1, ldx2, C1, C2, N, Hin, Win, Hout, Wout, R, stride, pad, ups, M;
    __bf16* out; long long ldo;
};

// out[(tap * (C1 + C2) + c)][m] = x[(n, iy, ix) of output pixel m and tap][c], zero for padding taps and for m >= M.
// grid (Mp / 64, (C1 + C2) / 64, R * R): a 64-pixel x 64-channel tile goes through LDS, read as 128-byte pixel rows and
// written as 32-byte runs of consecutive pixels.
__global__ __launch_bounds__(256) void im2col_t_kernel(const Im2colArgs a) {
    __shared__ unsigned short tile[64][72];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    const int kr = tap / a.R, ks = tap - kr * a.R;
    const int Hl = a.Hin * a.ups, Wl = a.Win * a.ups, ush = a.ups >> 1;
    const bool src2 = c0 >= a.C1;
    const __bf16* src = src2 ? a.x2 : a.x1;
    const int ld = src2 ? a.ldx2 : a.ldx1, cc = src2 ? c0 - a.C1 : c0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = tid + i * 256, p = ch >> 3, chunk = ch & 7;
        const int m = m0 + p;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m < a.M) {
            const int hw = a.Hout * a.Wout;
            const int n = m / hw, rem = m - n * hw, oy = rem / a.Wout, ox = rem - oy * a.Wout;
            const int iy = oy * a.stride - a.pad + kr, ix = ox * a.stride - a.pad + ks;
            if ((unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl)
                v = *(const u32x4*)(src + ((long long)(n * a.Hin + (iy >> ush)) * a.Win + (ix >> ush)) * ld + cc + chunk * 8);
        }
        *(u32x4*)(&tile[p][chunk * 8]) = v;
    }
    __syncthreads();
    const int c = tid >> 2, seg = tid & 3;
    unsigned short o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = tile[seg * 16 + e][c];
    __bf16* dst = a.out + ((long long)tap * (a.C1 + a.C2) + c0 + c) * a.ldo + m0 + seg * 16;
    u32x4 w0, w1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w0[e] = (unsigned)o[2 * e] | ((unsigned)o[2 * e + 1] << 16);
        w1[e] = (unsigned)o[8 + 2 * e] | ((unsigned)o[8 + 2 * e + 1] << 16);
    }
    *(u32x4*)dst = w0;
    *(u32x4*)(dst + 8) = w1;
}

// fp32 NCHW source with a handful of channels (conv_in of the trunk, first conv of the conditioning embedding; 3x3,
// pad 1): out[(tap * Cin + ci)][m] for the 9 * Cin real rows, zeros up to `rows_pad` rows; image n reads x[n % Nsrc].
__global__ __launch_bounds__(256) void im2col_t_nchw_kernel(const float* x, int Nsrc, int N, int Cin, int H, int W, int stride,
                                                            int Ho, int Wo, int M, __bf16* out, long long ldo, int rows_pad) {
    const long long total = (long long)rows_pad * ldo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int k = (int)(idx / ldo), m = (int)(idx - (long long)k * ldo);
        float v = 0.f;
        if (k < 9 * Cin && m < M) {
            const int tap = k / Cin, ci = k - tap * Cin, kr = tap / 3, ks = tap - kr * 3;
            const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo), oy = rem / Wo, ox = rem - oy * Wo;
            const int iy = oy * stride - 1 + kr, ix = ox * stride - 1 + ks;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = x[(((long long)(n % Nsrc) * Cin + ci) * H + iy) * W + ix];
        }
        st_bf(out + idx, v);
    }
}

// ------------------------------------------------------------------ column sums (bias, time-embedding row, LayerNorm gamma / beta)
// out[(split * nseg + seg)][c] = sum over the rows r of segment seg that fall to `split` of y[r][c]            (sum0)
// and, with x / rowstat, out2[...][c] = sum of y[r][c] * (x[r][c] - mean_r) * rstd_r                             (sum1)
// grid (C / 64 rounded up, nseg, nsplit); a block = 8 column chunks x 32 row lanes.
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* y, int ldy, int rows_per_seg, int C, const __bf16* x, int ldx,
                                                     const float* rowstat, float* out, float* out2) {
    __shared__ float red[2][32][64];
    const int tid = threadIdx.x, cch = tid & 7, rl = tid >> 3;
    const int c = blockIdx.x * 64 + cch * 8;
    const int seg = blockIdx.y, split = blockIdx.z, nsplit = gridDim.z;
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    if (c < C) {
        for (int r = split * 32 + rl; r < rows_per_seg; r += 32 * nsplit) {
            const long long row = (long long)seg * rows_per_seg + r;
            float f[8];
            unpack8(*(const u32x4*)(y + row * ldy + c), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s0[e] += f[e];
            if (x) {
                float g[8];
                unpack8(*(const u32x4*)(x + row * ldx + c), g);
                const float mean = rowstat[2 * row], rstd = rowstat[2 * row + 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[e] += f[e] * (g[e] - mean) * rstd;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][rl][cch * 8 + e] = s0[e]; red[1][rl][cch * 8 + e] = s1[e]; }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, col = tid & 63;
        if (blockIdx.x * 64 + col < C && (which == 0 || x)) {
            float t = 0.f;
            for (int r = 0; r < 32; ++r) t += red[which][r][col];
            float* o = which ? out2 : out;
            o[((long long)split * gridDim.y + seg) * C + blockIdx.x * 64 + col] = t;
        }
    }
}

// ------------------------------------------------------------------ GroupNorm (+SiLU) backward
struct GnBwdArgs {
    const __bf16* x1; const __bf16* x2; const __bf16* dy; __bf16* dx1; __bf16* dx2;
    int ldx1, ldx2, lddy, lddx1, lddx2, C1, C2, N, HW, groups, cpg, silu, nact;
    float eps; const float* gamma; const float* beta; float* dparam;  // dparam [N][C][2] = (dgamma, dbeta) of image n, or NULL
};

// One block per (image, group); the first nact = cpg * k threads work (a thread always on the same channel of the group), the
// rest of the last wave only takes part in the reductions (whole waves: a shuffle must not read an absent lane).
// y = silu?(xhat * gamma + beta): dt = dy * silu'(t); dxhat = dt * gamma;
// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), means over the (image, group) slice.
__global__ __launch_bounds__(256) void gn_bwd_kernel(const GnBwdArgs a) {
    __shared__ float red[8];
    __shared__ float pg[256], pb[256];
    const int n = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
    const int tid = threadIdx.x;
    const bool active = tid < a.nact;
    const int cl = tid % a.cpg, np = a.nact / a.cpg, pl = active ? tid / a.cpg : a.HW;  // idle threads: empty loops
    const int c = g * a.cpg + cl;
    const bool s2 = c >= a.C1;
    const __bf16* xs = s2 ? a.x2 + (c - a.C1) : a.x1 + c;
    const int ldx = s2 ? a.ldx2 : a.ldx1;
    __bf16* dxs = s2 ? a.dx2 + (c - a.C1) : a.dx1 + c;
    const int lddx = s2 ? a.lddx2 : a.lddx1;
    const long long r0 = (long long)n * a.HW;
    const float cnt = (float)a.HW * (float)a.cpg;
    float s = 0.f;
    for (int p = pl; p < a.HW; p += np) s += bf(xs + (r0 + p) * ldx);
    const float mean = block_sum(s, red) / cnt;
    float q = 0.f;
    for (int p = pl; p < a.HW; p += np) { const float d = bf(xs + (r0 + p) * ldx) - mean; q += d * d; }
    const float rstd = rsqrtf(block_sum(q, red) / cnt + a.eps);
    const float ga = a.gamma[c], be = a.beta[c];
    float s1 = 0.f, s2v = 0.f, dg = 0.f, db = 0.f;
    for (int p = pl; p < a.HW; p += np) {
        const float xh = (bf(xs + (r0 + p) * ldx) - mean) * rstd;
        float dt = bf(a.dy + (r0 + p) * a.lddy + c);
        if (a.silu) dt *= dsilu(fmaf(xh, ga, be));
        dg += dt * xh;
        db += dt;
        const float dxh = dt * ga;
        s1 += dxh;
        s2v += dxh * xh;
    }
    const float m1 = block_sum(s1, red) / cnt, m2 = block_sum(s2v, red) / cnt;
    if (a.dparam) {
        pg[tid] = dg; pb[tid] = db;
        __syncthreads();
        if (active && pl == 0) {
            float tg = 0.f, tb = 0.f;
            for (int k = 0; k < np; ++k) { tg += pg[k * a.cpg + cl]; tb += pb[k * a.cpg + cl]; }
            float* o = a.dparam + ((long long)n * (a.C1 + a.C2) + c) * 2;
            o[0] = tg; o[1] = tb;
        }
    }
    for (int p = pl; p < a.HW; p += np) {
        const float xh = (bf(xs + (r0 + p) * ldx) - mean) * rstd;
        float dt = bf(a.dy + (r0 + p) * a.lddy + c);
        if (a.silu) dt *= dsilu(fmaf(xh, ga, be));
        st_bf(dxs + (r0 + p) * lddx, rstd * (dt * ga - m1 - xh * m2));
    }
}

// ------------------------------------------------------------------ LayerNorm backward: one wave per row
// dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * gamma; rowstat[row] = (mean, rstd) for the
// parameter-gradient column sums (colsum_kernel).
__global__ __launch_bounds__(256) void ln_bwd_kernel(const __bf16* x, int ldx, const __bf16* dy, int lddy, int rows, int C, float eps,
                                                     const float* gamma, __bf16* dx, int lddx, float* rowstat) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const __bf16* xr = x + (long long)row * ldx;
    const __bf16* dr = dy + (long long)row * lddy;
    const float invC = 1.0f / (float)C;
    float s = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        float f[8];
        unpack8(*(const u32x4*)(xr + c), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
    }
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        float f[8];
        unpack8(*(const u32x4*)(xr + c), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        float f[8], d[8];
        unpack8(*(const u32x4*)(xr + c), f);
        unpack8(*(const u32x4*)(dr + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dyg = d[e] * gamma[c + e];
            s1 += dyg;
            s2 += dyg * (f[e] - mean) * rstd;
        }
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
    for (int c = lane * 8; c < C; c += 512) {
        float f[8], d[8], o[8];
        unpack8(*(const u32x4*)(xr + c), f);
        unpack8(*(const u32x4*)(dr + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (d[e] * gamma[c + e] - m1 - (f[e] - mean) * rstd * m2);
        *(u32x4*)(dx + (long long)row * lddx + c) = pack8(o);
    }
    if (lane == 0 && rowstat) { rowstat[2ll * row] = mean; rowstat[2ll * row + 1] = rstd; }
}

// ------------------------------------------------------------------ row softmax backward (attention scores)
// ds[r][c] = scale * p[r][c] * (dp[r][c] - sum_c' p[r][c'] dp[r][c']) for c < cols, 0 for cols <= c < cols_pad; in place over dp
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const __bf16* p, __bf16* dp, int ld, int rows, int cols, int cols_pad,
                                                               float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const __bf16* pr = p + (long long)row * ld;
    __bf16* dr = dp + (long long)row * ld;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += bf(pr + c) * bf(dr + c);
    s = wave_sum(s);
    for (int c = lane; c < cols; c += 64) st_bf(dr + c, scale * bf(pr + c) * (bf(dr + c) - s));
    for (int c = cols + lane; c < cols_pad; c += 64) st_bf(dr + c, 0.f);
}

// ------------------------------------------------------------------ element-wise backward passes
// GEGLU: h = (a | g) [rows][2C], y = a * gelu(g); dh = (dy * gelu(g) | dy * a * gelu'(g))
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const __bf16* h, int ldh, const __bf16* dy, int lddy, int rows, int C,
                                                        __bf16* dh, int lddh) {
    const int cols = C / 8;
    const long long total = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx % cols) * 8;
        float av[8], gv[8], d[8], da[8], dg[8];
        unpack8(*(const u32x4*)(h + (long long)r * ldh + c), av);
        unpack8(*(const u32x4*)(h + (long long)r * ldh + C + c), gv);
        unpack8(*(const u32x4*)(dy + (long long)r * lddy + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float cdf = 0.5f * (1.0f + erff(gv[e] * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * gv[e] * gv[e]);
            da[e] = d[e] * gv[e] * cdf;
            dg[e] = d[e] * av[e] * (cdf + gv[e] * pdf);
        }
        *(u32x4*)(dh + (long long)r * lddh + c) = pack8(da);
        *(u32x4*)(dh + (long long)r * lddh + C + c) = pack8(dg);
    }
}

// mode 0: out = silu(x); mode 1: out = dy * silu'(x)
__global__ __launch_bounds__(256) void silu_kernel(const __bf16* x, int ldx, const __bf16* dy, int lddy, int rows, int C, int mode,
                                                   __bf16* out, int ldo) {
    const int cols = C / 8;
    const long long total = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx % cols) * 8;
        float v[8], d[8];
        unpack8(*(const u32x4*)(x + (long long)r * ldx + c), v);
        if (mode == 1) {
            unpack8(*(const u32x4*)(dy + (long long)r * lddy + c), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = d[e] * dsilu(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
        *(u32x4*)(out + (long long)r * ldo + c) = pack8(v);
    }
}

// a[r][c] = alpha * a[r][c] + beta * b[r][c]  (gradient accumulation; fp32 arithmetic, bf16 storage)
__global__ __launch_bounds__(256) void axpby_bf16_kernel(__bf16* a, int lda, const __bf16* b, int ldb, int rows, int C, float alpha,
                                                         float beta) {
    const int cols = C / 8;
    const long long total = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx % cols) * 8;
        float x[8], y[8];
        unpack8(*(const u32x4*)(a + (long long)r * lda + c), x);
        unpack8(*(const u32x4*)(b + (long long)r * ldb + c), y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = alpha * x[e] + beta * y[e];
        *(u32x4*)(a + (long long)r * lda + c) = pack8(x);
    }
}

// mode 0 (adjoint of nearest x2 upsampling): out[n][y][x] = sum of in[n][2y + dy][2x + dx], in is [N][2H][2W][C]
// mode 1 (adjoint of taking every second pixel, the stride of a stride-2 convolution): out [N][H][W][C] is zero except
//         out[n][2y][2x] = in[n][y][x], in is [N][Hs][Ws][C]
__global__ __launch_bounds__(256) void resample_bwd_kernel(const __bf16* in, int ldi, __bf16* out, int ldo, int N, int H, int W,
                                                           int Hs, int Ws, int C, int mode) {
    const int cols = C / 8;
    const long long total = (long long)N * H * W * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % cols) * 8;
        const long long pix = idx / cols;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (mode == 0) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float f[8];
                    unpack8(*(const u32x4*)(in + (((long long)n * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * ldi + c), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += f[e];
                }
        } else if (!(y & 1) && !(x & 1) && (y >> 1) < Hs && (x >> 1) < Ws) {
            unpack8(*(const u32x4*)(in + (((long long)n * Hs + (y >> 1)) * Ws + (x >> 1)) * ldi + c), acc);
        }
        *(u32x4*)(out + pix * ldo + c) = pack8(acc);
    }
}

int grid_for(long long items, int cap = 4096) {
    const long long g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

}  // namespace

extern "C" int ccnet_im2col_t_bf16(const void* x1, int ldx1, int C1, const void* x2, int ldx2, int C2, int N, int Hin, int Win,
                                   int Hout, int Wout, int R, int stride, int pad, int ups, void* out, int64_t ldo,
                                   void* stream) {
    if (!x1 || !out || N <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return CCNET_EINVAL;
    if (C1 <= 0 || (C1 % 64) || C2 < 0 || (C2 % 64) || (C2 > 0 && !x2) || (ldx1 % 8) || (C2 > 0 && (ldx2 % 8))) return CCNET_EINVAL;
    if ((R != 1 && R != 3) || stride <= 0 || (ups != 1 && ups != 2)) return CCNET_EUNSUP;
    const long long M = (long long)N * Hout * Wout;
    if (ldo < M || (ldo % 64) || M > 0x7fffffffLL) return CCNET_EINVAL;
    Im2colArgs a{};
    a.x1 = (const __bf16*)x1; a.x2 = (const __bf16*)x2; a.ldx1 = ldx1; a.ldx2 = ldx2; a.C1 = C1; a.C2 = C2; a.N = N;
    a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.R = R; a.stride = stride; a.pad = pad; a.ups = ups; a.M = (int)M;
    a.out = (__bf16*)out; a.ldo = ldo;
    hipLaunchKernelGGL(im2col_t_kernel, dim3((unsigned)(ldo / 64), (C1 + C2) / 64, R * R), dim3(256), 0, (hipStream_t)stream, a);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_im2col_t_nchw_f32(const float* x, int Nsrc, int N, int Cin, int H, int W, int stride, void* out, int64_t ldo,
                                       int rows_pad, void* stream) {
    if (!x || !out || Nsrc <= 0 || N <= 0 || Cin <= 0 || Cin > 8 || H <= 0 || W <= 0 || stride <= 0) return CCNET_EINVAL;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long long M = (long long)N * Ho * Wo;
    if (ldo < M || rows_pad < 9 * Cin || M > 0x7fffffffLL) return CCNET_EINVAL;
    hipLaunchKernelGGL(im2col_t_nchw_kernel, dim3(grid_for((long long)rows_pad * ldo)), dim3(256), 0, (hipStream_t)stream, x, Nsrc,
                       N, Cin, H, W, stride, Ho, Wo, (int)M, (__bf16*)out, (long long)ldo, rows_pad);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_colsum_bf16(const void* y, int ldy, int nseg, int rows_per_seg, int C, const void* x, int ldx,
                                 const float* rowstat, int nsplit, float* out, float* out2, void* stream) {
    if (!y || !out || nseg <= 0 || rows_per_seg <= 0 || C <= 0 || (C % 8) || (ldy % 8) || nsplit <= 0 || nsplit > 65535 ||
        nseg > 65535)
        return CCNET_EINVAL;
    if (x && (!rowstat || !out2 || (ldx % 8))) return CCNET_EINVAL;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, nseg, nsplit), dim3(256), 0, (hipStream_t)stream, (const __bf16*)y, ldy,
                       rows_per_seg, C, (const __bf16*)x, ldx, rowstat, out, out2);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_groupnorm_bwd_bf16(const void* x1, int ldx1, int C1, const void* x2, int ldx2, int C2, const void* dy,
                                        int lddy, int N, int HW, int groups, float eps, const float* gamma, const float* beta,
                                        int silu, void* dx1, int lddx1, void* dx2, int lddx2, float* dparam, void* stream) {
    if (!x1 || !dy || !dx1 || !gamma || !beta) return CCNET_EINVAL;
    if (C2 < 0 || (C2 > 0 && (!x2 || !dx2))) return CCNET_EINVAL;
    const int C = C1 + C2;
    if (N <= 0 || HW <= 0 || C1 <= 0 || groups <= 0 || (C % groups)) return CCNET_EINVAL;
    const int cpg = C / groups;
    if (cpg > 256) return CCNET_EUNSUP;  // (a thread picks its source by channel: a group may straddle the two sources)
    if ((long long)N * groups > 0x7fffffffLL) return CCNET_EUNSUP;
    GnBwdArgs a{};
    a.x1 = (const __bf16*)x1; a.x2 = (const __bf16*)x2; a.dy = (const __bf16*)dy; a.dx1 = (__bf16*)dx1; a.dx2 = (__bf16*)dx2;
    a.ldx1 = ldx1; a.ldx2 = ldx2; a.lddy = lddy; a.lddx1 = lddx1; a.lddx2 = lddx2; a.C1 = C1; a.C2 = C2; a.N = N; a.HW = HW;
    a.groups = groups; a.cpg = cpg; a.silu = silu; a.eps = eps; a.gamma = gamma; a.beta = beta; a.dparam = dparam;
    a.nact = cpg * (256 / cpg);
    hipLaunchKernelGGL(gn_bwd_kernel, dim3(N * groups), dim3((a.nact + 63) / 64 * 64), 0, (hipStream_t)stream, a);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_layernorm_bwd_bf16(const void* x, int ldx, const void* dy, int lddy, int rows, int C, float eps,
                                        const float* gamma, void* dx, int lddx, float* rowstat, void* stream) {
    if (!x || !dy || !dx || !gamma || rows <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (lddy % 8) || (lddx % 8)) return CCNET_EINVAL;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ldx,
                       (const __bf16*)dy, lddy, rows, C, eps, gamma, (__bf16*)dx, lddx, rowstat);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_softmax_bwd_rows_bf16(const void* p, void* dp, int ld, int rows, int cols, int cols_pad, float scale,
                                           void* stream) {
    if (!p || !dp || rows <= 0 || cols <= 0 || cols_pad < cols || ld < cols_pad) return CCNET_EINVAL;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const __bf16*)p,
                       (__bf16*)dp, ld, rows, cols, cols_pad, scale);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_geglu_bwd_bf16(const void* h, int ldh, const void* dy, int lddy, int rows, int C, void* dh, int lddh,
                                    void* stream) {
    if (!h || !dy || !dh || rows <= 0 || C <= 0 || (C % 8) || (ldh % 8) || (lddy % 8) || (lddh % 8) || ldh < 2 * C || lddh < 2 * C)
        return CCNET_EINVAL;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for((long long)rows * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)h, ldh, (const __bf16*)dy, lddy, rows, C, (__bf16*)dh, lddh);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_silu_bf16(const void* x, int ldx, const void* dy, int lddy, int rows, int C, int mode, void* out, int ldo,
                               void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldo % 8) || mode < 0 || mode > 1) return CCNET_EINVAL;
    if (mode == 1 && (!dy || (lddy % 8))) return CCNET_EINVAL;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for((long long)rows * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x,
                       ldx, (const __bf16*)dy, lddy, rows, C, mode, (__bf16*)out, ldo);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_axpby_bf16(void* a, int lda, const void* b, int ldb, int rows, int C, float alpha, float beta, void* stream) {
    if (!a || !b || rows <= 0 || C <= 0 || (C % 8) || (lda % 8) || (ldb % 8)) return CCNET_EINVAL;
    hipLaunchKernelGGL(axpby_bf16_kernel, dim3(grid_for((long long)rows * (C / 8))), dim3(256), 0, (hipStream_t)stream, (__bf16*)a,
                       lda, (const __bf16*)b, ldb, rows, C, alpha, beta);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int ccnet_resample_bwd_bf16(const void* in, int ldi, void* out, int ldo, int N, int H, int W, int Hs, int Ws, int C,
                                       int mode, void* stream) {
    if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldi % 8) || (ldo % 8) || mode < 0 || mode > 1)
        return CCNET_EINVAL;
    if (mode == 1 && (Hs <= 0 || Ws <= 0 || 2 * (Hs - 1) >= H || 2 * (Ws - 1) >= W)) return CCNET_EINVAL;
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(grid_for((long long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)in, ldi, (__bf16*)out, ldo, N, H, W, Hs, Ws, C, mode);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

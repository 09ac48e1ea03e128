// K1/K2/K8/K10: implicit-GEMM convolution / linear layer on bf16 MFMA for gfx950.
//
// Data layout in HBM: activations NHWC bf16 (a "pixel row" of C channels is contiguous), weights
// [Cout][R*R*Cin] bf16 with K ordered (kr, ks, cin) so a 64-channel K-slice of one filter tap is one
// contiguous 128-byte run for both operands.  One workgroup of WM x WN waves computes a
// (32*TJ*WM pixels) x (32*TI*WN couts) tile; every wave owns TI x TJ accumulators of
// v_mfma_f32_32x32x16_bf16 with the weights as the A operand (rows = cout) and the pixels as the B operand
// (cols = pixel), so each lane ends up with 4 consecutive couts of one pixel per accumulator quad.
//
// K loop: one step = BK (64 or 32) channels of one tap.  Operand tiles go HBM/L2 -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write); the buffer range check returns 0 for the
// padding taps and for rows beyond M / Cout, so the gather has no branches.  The LDS image of a tile is
// [row][BK] bf16 with the 16-B chunk index XOR-swizzled by the row -- applied on the SOURCE address,
// because a DMA lands at wave-base + lane*16 -- which makes the ds_read_b128 fragment reads conflict-free.
// NS-stage ring (2..4 stages): the DMAs of steps k+1 .. k+NS-1 are in flight while step k's MFMAs run; a step
// ends with a COUNTED s_waitcnt vmcnt(N) + raw s_barrier, so the newest NS-2 steps stay in flight across it.
// Persistent instances walk tiles b, b + grid, ... and issue the next tile's first stages before their epilogue.
//
// Epilogue: straight from the accumulators (they start at bias + time-embedding row); v_permlane32_swap pairs
// give each lane 8 consecutive couts of its pixel, i.e. one 16-B store / residual load per 16 columns.
#include "common.h"
#include <atomic>
#include <type_traits>
#include <mutex>
#include <vector>

#ifndef CONV_DIAG
#define CONV_DIAG 0  // timing experiment: bit 0 = the plain epilogue of the tiled kernel loads no residual and stores nothing
#endif
namespace {

#ifdef CCNET_STAMP
// Diagnostic build (never the shipped library): wave 0 of every block stores s_memtime at phase boundaries into a
// buffer of its own; no output value depends on it.
#define STAMP(k)                                                                      \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        if (a.dbg && threadIdx.x == 0) {                                               \
            a.dbg[(long long)stamp_slot * 8 + (k)] = __builtin_amdgcn_s_memtime();        \
            if ((k) == 2 || (k) == 3) a.dbg[(long long)stamp_slot * 8 + 3 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

struct ConvArgs {
    const __bf16* x1; const __bf16* x2; const __bf16* w;
    const float* bias; const float* rowvec; const __bf16* res1; const __bf16* res2;
    __bf16* out; __bf16* out2;
    int N, Hin, Win, Hout, Wout, C1, C2, Cout, R, stride, pad, ups;
    int ldx1, ldx2, ldw, ldo, ldo2, ldres1, ldres2, ld_rowvec, act;
    float scale;
    int M, HWout, KT, CT1, CT, tilesN, tilesM, tiles, mfast;
    int korder;      // 1: K-steps walk (channel chunk, tap) instead of (tap, channel chunk); the weight image keeps its (tap, cin) order
    int winM, winN;  // mfast == 2: tiles are walked in windows of winM x winN = 32 tiles (one per CU of an XCD)
    long long sx1, sw, so;
    const float* rowstat; const float* colsum;  // LayerNorm folded into the GEMM (see the accumulator start)
    int rowstat_parts; float ln_eps;            // > 0: rowstat holds that many (mean, M2) partials per row (a producer's stat_out)
    float* stat_out; int stat_parts;            // STAT instances: per-row (mean, M2) of every 32*TI-column span of the output
    float* out32; int cout_real;  // SPLIT instances only: != nullptr -> final fp32 NCHW output (+ bias), Cout <= 8 edge convs
    int splitk;   // > 1: blockIdx.y selects a K slice, raw fp32 partial sums go to ws[slice][M][Cout]
    float* ws;
    unsigned x1_bytes, x2_bytes, w_bytes;
#ifdef CCNET_STAMP
    unsigned long long* dbg;  // diagnostic build only: per-block phase time stamps
#endif
};

#ifndef CCNET_DMA_FRONT
#define CCNET_DMA_FRONT 2   // MFMA k-steps of a K-step that carry its DMA issues (tuning builds: -DCCNET_DMA_FRONT=n)
#endif
constexpr unsigned OOB = 0x80000000u;  // beyond any num_records we accept (< 2^31 bytes)

__device__ __forceinline__ void buf_dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, unsigned voff) {
    // LDS-DMA: 64 lanes x 16 B land at lds_wave_base + lane*16 (wave-uniform base), no VGPR round trip
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0,
                                             0, 0);
}

// GEGLU = true: the packed weight rows interleave 8-row blocks of the "a" half and the "gate" half of an
// MLPBlock(act="GEGLU").linear1, so register quads (2h, 2h+1) of every accumulator hold a and gate of the same 4
// outputs in the same lane; the epilogue writes a * gelu_erf(gate) (Cout/2 columns, 16 per accumulator block).
// Any TI works (the 256x320 tile included); the interleave unit is the 8 couts a lane pair (hh = 0, 1) owns per quad.
// PERSIST = true: the grid is at most one block per CU slot and each block walks tiles b, b + grid, b + 2 grid, ...;
// the next tile's first NS-1 stages are issued BEFORE the current tile's epilogue, so a short-K layer's tile start-up
// (address set-up, HBM/L2 latency of the first stages) hides under the previous tile's stores.
// RESPRE = true: the residual (res1) tile is fetched into registers at the START of the tile's K loop instead of inside the
// epilogue.  A short-K layer with a residual is two phases that every CU runs in lockstep -- an MFMA-bound K loop during
// which HBM idles, then an epilogue that moves the tile's residual + output at the per-CU streaming rate (~10 B/clk/CU:
// 26k of the 35k epilogue cycles of a 256x320 tile, stamps in profiles/r02_stamp_clock.txt) -- so the memory time adds to the
// MFMA time.  Prefetching spreads the residual read over the K loop; the prefetched tile does not fit L2 (164 KB per CU x
// 32 CUs per XCD > 4 MB), so registers are the only place, and only the 80-accumulator 128x320 tile has 40 to spare.
// STAT = true: the epilogue also writes, for every output row and every span of 32*TI columns (a wave's share of the row),
// the span's (mean, M2) in fp32 to stat_out[row][span][2] -- the statistics of the LayerNorm that follows the layer, taken
// from the fp32 values on their way out (shifted sums around the lane's first value, Chan merge of the two lane halves), so
// the consumer (rowstat / rowstat_parts of the next GEMM) needs no pass over the tensor.
template <int WM, int WN, int TJ, int TI, bool GEGLU, int NS, int BK, bool SPLIT, bool PERSIST, bool RESPRE = false,
          bool STAT = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_kernel(const ConvArgs a) {
    static_assert(!(SPLIT && PERSIST), "split-K launches are never persistent");
    static_assert(!RESPRE || (!SPLIT && !GEGLU), "residual prefetch: plain epilogue only");
    static_assert(!STAT || (!SPLIT && !GEGLU), "row statistics: plain epilogue only");
    constexpr int NT = 64 * WM * WN, BM = 32 * TJ * WM, BN = 32 * TI * WN;
    static_assert(BK == 64 || BK == 32, "K-step of 64 or 32 channels");
    constexpr int CPW = BK / 8;            // 16-B chunks per tile row
    constexpr int RB = BK * 2;             // bytes per tile row
    constexpr int RPP = NT / CPW;          // tile rows covered by one pass of all threads
    constexpr int NX = BM / RPP, NW = BN / RPP;
    constexpr int NL = NX + NW;            // DMA instructions per thread per K-step
    constexpr int STAGE = (BM + BN) * RB;  // bytes per pipeline stage
    constexpr int KS = BK / 16;            // MFMA k-steps per K-step
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows per DMA pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef CCNET_STAMP
    int stamp_slot = blockIdx.x;  // tile (virtual block) the stamps belong to
#endif
    STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, hh = lane >> 5;

    // XCD-aware tile order: blocks b, b+8, ... share an L2, give them consecutive tiles (bijective form).
    // (persistent form: virtual block vb = b + i * grid keeps b's XCD because the grid is a multiple of 8)
    auto tile_of = [&](int vb) {
        const int nwg = a.tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    };
    // tile order inside an XCD's contiguous range: N-fastest keeps the gathered pixel panel hot in that L2 (big-M
    // layers); M-fastest (mfast) keeps a weight panel hot and lets each XCD stream only its share of the weights
    // from HBM (small-M, deep-K layers whose weights exceed the activations).
    int m0, n0;  // tile whose operands are being fetched
    const int z = blockIdx.z;

    const __bf16* x1 = a.x1 + (long long)z * a.sx1;
    const __bf16* wp = a.w + (long long)z * a.sw;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)x1, 0, a.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, a.w_bytes, 0x00020000);

    const int lrow = tid / CPW, lchunk = tid % CPW;
    // 16-B chunk swizzle (applied on the DMA source and on the fragment read): conflict-free ds_read_b128
    auto swz_of = [](int r) { return BK == 64 ? (r >> 1) & 7 : (r >> 2) & 3; };

    // per-thread gather rows: image index and top-left input coordinate of each output pixel
    int xn[NX], xiy0[NX], xix0[NX];
    unsigned xsw[NX], woff[NW];
#pragma unroll
    for (int i = 0; i < NX; ++i) xsw[i] = (unsigned)((lchunk ^ swz_of(lrow + i * RPP)) << 3);
    auto setup_tile = [&](int vb) {
        const int tile = tile_of(vb);
        if (a.mfast == 2) {
            // windowed order: the 32 tiles an XCD's CUs run together are winM pixel panels x winN cout tiles, so both the pixel
            // panels and the weight panels they share stay within that XCD's L2 (N-fastest shares one pixel panel and streams
            // 32 weight panels through it)
            const int w = tile >> 5, r = tile & 31, wnc = a.tilesN / a.winN;
            const int mw = w / wnc, nw = w - mw * wnc;
            const int rm = r / a.winN;
            m0 = (mw * a.winM + rm) * BM;
            n0 = (nw * a.winN + (r - rm * a.winN)) * BN;
        } else {
            m0 = (a.mfast ? tile % a.tilesM : tile / a.tilesN) * BM;
            n0 = (a.mfast ? tile / a.tilesM : tile % a.tilesN) * BN;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int m = m0 + lrow + i * RPP;
            if (m < a.M) {
                const int n = m / a.HWout, rem = m - n * a.HWout;
                const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
                xn[i] = n; xiy0[i] = oy * a.stride - a.pad; xix0[i] = ox * a.stride - a.pad;
            } else {
                xn[i] = -1; xiy0[i] = 0; xix0[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int r = lrow + i * RPP, n = n0 + r;
            woff[i] = n < a.Cout ? (unsigned)(n * a.ldw + ((lchunk ^ swz_of(r)) << 3)) * 2u : OOB;
        }
    };
    const int Hl = a.Hin * a.ups, Wl = a.Win * a.ups, ush = a.ups >> 1;

    int pix[NX];
    auto set_tap = [&](int tap) {
        const int kr = tap / a.R, ks = tap - kr * a.R;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int iy = xiy0[i] + kr, ix = xix0[i] + ks;
            const bool ok = xn[i] >= 0 && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
            pix[i] = ok ? (xn[i] * a.Hin + (iy >> ush)) * a.Win + (ix >> ush) : -1;
        }
    };

    // K-step being fetched: (tap_n, cc_n) and its flat index kt_n; slices [lo, hi) of its NL DMA instructions
    // split-K: this block accumulates K-steps [kt_lo, kt_lo + KTl) of the a.KT steps
    const int slice = SPLIT ? blockIdx.y : 0;
    const int kt_lo = SPLIT ? (int)((long long)a.KT * slice / a.splitk) : 0;
    const int KTl = SPLIT ? (int)((long long)a.KT * (slice + 1) / a.splitk) - kt_lo : a.KT;
    int tap_n, cc_n, kt_n;
    auto issue_slice = [&](int s, int lo, int hi) {
        char* xb = smem + s * STAGE + wave * 1024;
        char* wb = xb + BM * RB;
        const bool src2 = cc_n >= a.CT1;
        const unsigned co = (unsigned)((src2 ? cc_n - a.CT1 : cc_n) * BK);
        const unsigned ld = (unsigned)(src2 ? a.ldx2 : a.ldx1);
        const unsigned kb = (unsigned)kt_n * (unsigned)RB;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l < lo || l >= hi) continue;
            if (l < NX) {
                const unsigned vo = pix[l] < 0 ? OOB : ((unsigned)pix[l] * ld + co + xsw[l]) * 2u;
                if (!src2) buf_dma16(r1, xb + l * RPP * RB, vo);
                else buf_dma16(r2, xb + l * RPP * RB, vo);
            } else {
                const int i = l - NX;
                buf_dma16(rw, wb + i * RPP * RB, woff[i] == OOB ? OOB : woff[i] + kb);
            }
        }
    };
    auto advance = [&]() {
        ++kt_n;
        if (++cc_n == a.CT) { cc_n = 0; ++tap_n; }
    };

    const int swz = swz_of(l31);
    const int xrow_off = (wm * 32 * TJ + l31) * RB, wrow_off = (wn * 32 * TI + l31) * RB;

    // NS-stage ring, prefetch distance NS-1 K-steps: step kt's MFMAs run while the DMAs of steps kt+1 .. kt+NS-1 are in
    // flight.  The wait that ends a step is COUNTED (all but the newest NS-2 steps' DMAs retired) and the barrier is
    // a raw s_barrier, so loads stay in flight across it; the stage refilled during step kt is the one read during
    // step kt-1, which every wave has left before it passed that step's barrier.
    static_assert(NS >= 2 && NL * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
    // a tile's start: gather addresses, then its first NS-1 stages go out (every stage is free at that point)
    auto start_tile = [&](int vb) {
        setup_tile(vb);
        tap_n = kt_lo / a.CT; cc_n = kt_lo - tap_n * a.CT; kt_n = kt_lo;
        set_tap(tap_n);
#pragma unroll
        for (int p = 0; p < NS - 1; ++p) {
            if (p < KTl) {
                if (p > 0 && cc_n == 0) set_tap(tap_n);
                issue_slice(p, 0, NL);
                advance();
            }
        }
    };
    STAMP(1);
    int vb = blockIdx.x;
    start_tile(vb);
    for (;;) {  // one trip per tile (a single trip unless PERSIST)
#ifdef CCNET_STAMP
    stamp_slot = vb;
#endif
    const int cm0 = m0, cn0 = n0;  // tile being computed; (m0, n0) move on to the next tile before the epilogue
    __builtin_amdgcn_sched_barrier(0);  // the first stages' DMAs are in flight before the accumulator-init loads go out
    // Accumulators start at bias[cout] (+ the image's time-embedding row): register (i, j, 4q + e) of a lane is cout
    // n0 + wn*32*TI + 32 i + 8 q + 4 hh + e of pixel m0 + wm*32*TJ + 32 j + l31, so the epilogue has no column-indexed
    // operand left and these loads (issued after the first stages' DMAs) complete under the first tile's DMA latency.
    // LayerNorm prologue, folded (a.rowstat != nullptr): x is the UN-normalised row matrix, w = gamma o W and
    // bias = bias + W beta (packed so by the host), colsum[n] = sum_k w[n][k], rowstat[m] = (mean, rstd) of row m.
    // LN(x) W^T + b = rstd_m (sum_k x_mk w_nk - mean_m colsum_n) + bias_n, so the accumulators start at
    // bias_n / rstd_m - mean_m colsum_n and are scaled by rstd_m after the K loop: the normalised activations are never
    // written to, or re-read from, HBM.  Without it (sigma = 1, mean = 0) the start is bias_n as before.
    float ln_rstd[TJ], ln_sigma[TJ], ln_mean[TJ];
    const bool ln = !SPLIT && a.rowstat != nullptr;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
        f32x2 st = {0.f, 1.f};
        if (ln) {
            const long long mm = m < a.M ? m : a.M - 1;
            if (a.rowstat_parts == 0) {
                st = *(const f32x2*)(a.rowstat + 2ll * mm);
            } else {
                // (mean, M2) of P equal column spans of the row, written by the producing layer's epilogue: Chan merge.
                // (Two dependent rounds of loads at the head of every tile: measured +18 % on the GEGLU layer.  A single
                // round of 16-byte loads needs registers the 160-accumulator instance does not have: it spilled.)
                const int P = a.rowstat_parts;
                const float* rp = a.rowstat + 2ll * mm * P;
                const float span = (float)(a.C1 / P);
                float ms = 0.f;
                for (int q = 0; q < P; ++q) ms += rp[2 * q];
                const float mean = ms / (float)P;
                float m2 = 0.f;
                for (int q = 0; q < P; ++q) { const float d = rp[2 * q] - mean; m2 += rp[2 * q + 1] + d * d * span; }
                st = f32x2{mean, rsqrtf(m2 / (float)a.C1 + a.ln_eps)};
            }
        }
        ln_mean[j] = st[0];
        ln_rstd[j] = st[1];
        ln_sigma[j] = ln ? 1.0f / st[1] : 1.0f;
    }
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nb = cn0 + wn * 32 * TI + i * 32 + 8 * q + 4 * hh;
            f32x4 bq = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
            if (!SPLIT && a.bias && nb < a.Cout) bq = *(const f32x4*)(a.bias + nb);  // split-K: added by the reduce pass
            if (ln && nb < a.Cout) cs = *(const f32x4*)(a.colsum + nb);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = ln ? bq[e] * ln_sigma[j] - ln_mean[j] * cs[e] : bq[e];
        }
    if (!GEGLU && !SPLIT && a.rowvec) {  // (the host rejects act = 2 with a time-embedding row)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
            const float* rv = a.rowvec + (long long)((m < a.M ? m : a.M - 1) / a.HWout) * a.ld_rowvec;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nb = cn0 + wn * 32 * TI + i * 32 + 8 * q + 4 * hh;
                    if (nb < a.Cout) {
                        const f32x4 t = *(const f32x4*)(rv + nb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += t[e];
                    }
                }
            __builtin_amdgcn_sched_barrier(0);  // one pixel block's 4*TI loads in flight at a time
        }
    }

    // residual prefetch: the epilogue's layout (after the permlane32 swap a lane owns 8 consecutive couts of its pixel);
    // STAT instances add the residual before the swap, so they prefetch in the accumulator layout (quads of 4 couts)
    u32x4 rpre[(RESPRE && !STAT) ? TJ : 1][(RESPRE && !STAT) ? TI : 1][2];
    u32x2 rpre4[(RESPRE && STAT) ? TJ : 1][(RESPRE && STAT) ? TI : 1][4];
    if constexpr (RESPRE) {
        const bool has1 = a.res1 != nullptr;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
            const __bf16* r1row = a.res1 + (long long)m * a.ldres1;
#pragma unroll
            for (int ob = 0; ob < TI; ++ob) {
                if constexpr (STAT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ncol = cn0 + wn * 32 * TI + ob * 32 + 8 * q + 4 * hh;
                        rpre4[j][ob][q] = (has1 && m < a.M && ncol < a.Cout) ? *(const u32x2*)(r1row + ncol) : u32x2{0u, 0u};
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int ncol = cn0 + wn * 32 * TI + 8 * hh + ob * 32 + 16 * p;
                        rpre[j][ob][p] = (has1 && m < a.M && ncol < a.Cout) ? *(const u32x4*)(r1row + ncol) : u32x4{0u, 0u, 0u, 0u};
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // issued here, before the K loop; consumed after it
    }

    // (a later tile of a persistent block has the previous epilogue's stores in the same counter: drain it, its
    // stages were issued a whole epilogue ago)
    if (KTl >= NS - 1 && vb == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(2);
    int s = 0, sn = NS - 1;  // stage being read, stage being refilled
    for (int kt = 0; kt < KTl; ++kt) {
        const bool more = kt + NS - 1 < KTl;
        const char* xb = smem + s * STAGE;
        const char* wb = xb + BM * RB;
        if (more && cc_n == 0) set_tap(tap_n);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int co = ((2 * ks + hh) ^ swz) << 4;
            bf16x8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8*)(wb + wrow_off + i * 32 * RB + co);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *(const bf16x8*)(xb + xrow_off + j * 32 * RB + co);
            // The step's DMAs go out within its first two MFMA k-steps, not spread over all four: with two stages the wait
            // that ends the step follows the last issue by the rest of the step only, and a quarter of the stage issued in
            // the last k-step had ~600 cycles to land (measured: +3..7 % on every 64-deep layer, tools/bench_conv.py
            // --cold, profiles/r02_dma_front.txt; all in the first k-step is neutral: the burst delays that k-step's MFMAs).
            // Also measured there and dropped: the loop rotated so that wait + barrier sit in front of the step's last
            // k-step and the next stage's first fragments are read under its MFMAs (-9 % on the 256x320 tile, which has no
            // registers for it, -1 % elsewhere: the DMAs then have two k-steps instead of the whole step to land).
            constexpr int KF = CCNET_DMA_FRONT < KS ? CCNET_DMA_FRONT : KS;
            if (more && ks < KF) issue_slice(sn, (NL * ks) / KF, (NL * (ks + 1)) / KF);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            advance();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (NS - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s = s + 1 == NS ? 0 : s + 1;
        sn = sn + 1 == NS ? 0 : sn + 1;
    }

    STAMP(3);
    bool last = true;
    if constexpr (PERSIST) {
        // every wave is past the K loop's final barrier: all stages are free, the next tile's first stages go out now
        // and land while this tile's epilogue runs
        vb += gridDim.x;
        last = vb >= a.tiles;
        if (!last) start_tile(vb);
    }
    // ---------------------------------------------------------------- epilogue
    // Straight from the accumulators, no LDS: after v_permlane32_swap of the (q even, q odd) register pairs a lane
    // holds 8 consecutive couts of its pixel, i.e. one 16-B store (and one 16-B residual load) per 16 columns.
    constexpr int TIO = TI;                // output blocks per wave: one per accumulator block,
    constexpr int OBW = GEGLU ? 16 : 32;   // 16 columns wide under GEGLU (a and gate share the block), else 32
    constexpr int NP = GEGLU ? 1 : 2;      // 16-column store groups per output block
    __bf16* outp = a.out + (long long)z * a.so;
    const int cout_out = GEGLU ? a.Cout / 2 : a.Cout;
    const int ncol0 = (GEGLU ? cn0 / 2 + wn * 16 * TI : cn0 + wn * 32 * TI) + 8 * hh;
    if constexpr (SPLIT) {
        // raw fp32 partial sums, 16 B per lane (4 consecutive couts of its pixel); ccnet's reduce kernel sums the
        // slices in slice order and applies the epilogue, so results do not depend on the launch geometry
        if (a.out32) {
            // edge convolution with a handful of couts (UNet / AEKL output conv): fp32 NCHW, bias added here; lanes of a
            // store are consecutive pixels of one cout plane
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
                if (m < a.M) {
                    const int img = m / a.HWout, pix = m - img * a.HWout;
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int co = cn0 + wn * 32 * TI + i * 32 + 8 * q + 4 * hh + e;
                                if (co < a.cout_real)
                                    a.out32[((long long)img * a.cout_real + co) * a.HWout + pix] =
                                        acc[i][j][4 * q + e] + (a.bias ? a.bias[co] : 0.f);
                            }
                }
            }
            STAMP(4);
            return;
        }
        float* wsl = a.ws + (long long)slice * a.M * a.Cout;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
            if (m < a.M) {
                float* wrow = wsl + (long long)m * a.Cout + cn0 + wn * 32 * TI + 4 * hh;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (cn0 + wn * 32 * TI + i * 32 + 8 * q + 4 * hh < a.Cout)
                            *(f32x4*)(wrow + i * 32 + 8 * q) =
                                f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    }
            }
        }
        STAMP(4);
        return;
    }
    if constexpr (!SPLIT) {
    auto swap32 = [](unsigned& lo, unsigned& hi) {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    };
    // element-wise stages run in place over the accumulators, each under ONE wave-uniform branch with a straight-line
    // body (per-item branches inside the unrolled store loops make hipcc spill)
    if (a.rowstat) {  // folded LayerNorm: the row's 1 / sigma
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] *= ln_rstd[j];
    }
    if constexpr (GEGLU) {
        // quads (0, 1) -> outputs 4 hh + e, quads (2, 3) -> outputs 8 + 4 hh + e of the block's 16; the results land in
        // registers 0..7, i.e. in the layout the first 16-column store group of a plain block has
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x16& v = acc[i][j];
                        const f32x2 r = geglu_pk(f32x2{v[8 * h2 + e], v[8 * h2 + e + 1]}, f32x2{v[8 * h2 + 4 + e], v[8 * h2 + 5 + e]});
                        acc[i][j][4 * h2 + e] = r[0];
                        acc[i][j][4 * h2 + e + 1] = r[1];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // four GELU chains in flight at a time (register budget)
                }
            }
    }
    if (a.act == 1) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] = silu_f(acc[i][j][k]);
    }
    if (a.scale != 1.0f) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] *= a.scale;
    }
    if constexpr (STAT) {
        // Three passes over the accumulators, so that no pass carries another's state (the 160-accumulator tile has no
        // registers to spare): (1) the residual is added in the ACCUMULATOR layout -- a lane's register quad q of block i is
        // couts 32 i + 8 q + 4 hh .. + 3 of its pixel, an 8-byte load -- so nothing is exchanged between lanes yet; (2) the
        // row statistics of the finished values; (3) pack, permlane32 swap and store as the plain epilogue does.
        const bool has1 = a.res1 != nullptr;
        if (has1) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
                const __bf16* r1row = a.res1 + (long long)(m < a.M ? m : a.M - 1) * a.ldres1 + cn0 + wn * 32 * TI + 4 * hh;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    u32x2 rr[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (RESPRE) rr[q] = rpre4[j][i][q];
                        else rr[q] = cn0 + wn * 32 * TI + i * 32 + 8 * q + 4 * hh < a.Cout ? *(const u32x2*)(r1row + i * 32 + 8 * q) : u32x2{0u, 0u};
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[i][j][4 * q] += __uint_as_float(rr[q][0] << 16);
                        acc[i][j][4 * q + 1] += __uint_as_float(rr[q][0] & 0xffff0000u);
                        acc[i][j][4 * q + 2] += __uint_as_float(rr[q][1] << 16);
                        acc[i][j][4 * q + 3] += __uint_as_float(rr[q][1] & 0xffff0000u);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one block's residual loads in flight at a time (register budget)
                }
            }
        }
        if (a.stat_out) {
            // shifted sums around the lane's first value (robust against a large row mean), Chan merge with lane ^ 32
            constexpr float n = 16.0f * TI;  // values per lane and row; the other half of the span sits in lane ^ 32
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float c = acc[0][j][0];
                float ss = 0.f, sq = 0.f;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float d = acc[i][j][k] - c;
                        ss += d;
                        sq = fmaf(d, d, sq);
                    }
                const float mean = c + ss * (1.0f / n);
                const float m2 = fmaxf(sq - ss * ss * (1.0f / n), 0.f);
                const float mean_b = __shfl_xor(mean, 32), m2_b = __shfl_xor(m2, 32);
                const float d = mean - mean_b;
                const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
                if (hh == 0 && m < a.M)
                    *(f32x2*)(a.stat_out + ((long long)m * a.stat_parts + (cn0 / (32 * TI) + wn)) * 2) =
                        f32x2{0.5f * (mean + mean_b), m2 + m2_b + d * d * (0.5f * n)};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (STAT || (a.res1 == nullptr && a.out2 == nullptr)) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
            const bool mok = m < a.M;
            __bf16* orow = outp + (long long)m * a.ldo;
#pragma unroll
            for (int ob = 0; ob < TIO; ++ob)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x16& v = acc[ob][j];
                    unsigned d0 = pack_bf16x2(v[8 * p], v[8 * p + 1]), d1 = pack_bf16x2(v[8 * p + 2], v[8 * p + 3]);
                    unsigned d2 = pack_bf16x2(v[8 * p + 4], v[8 * p + 5]), d3 = pack_bf16x2(v[8 * p + 6], v[8 * p + 7]);
                    swap32(d0, d2);
                    swap32(d1, d3);
                    const int ncol = ncol0 + ob * OBW + 16 * p;
                    if (mok && ncol < cout_out) *(u32x4*)(orow + ncol) = u32x4{d0, d1, d2, d3};
                }
        }
    } else {
        const bool has1 = a.res1 != nullptr, has2 = a.out2 != nullptr;
        // the 160-accumulator persistent tile also carries the next tile's gather state through the epilogue
        constexpr int OBG = (PERSIST && TI * TJ >= 10) ? 1 : 2;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = cm0 + wm * 32 * TJ + j * 32 + l31;
            const bool mok = m < ((CONV_DIAG & 1) ? -a.M : a.M);
            __bf16* orow = outp + (long long)m * a.ldo;
            __bf16* orow2 = a.out2 + (long long)m * a.ldo2;
            const __bf16* r1row = a.res1 + (long long)m * a.ldres1;
            const __bf16* r2row = a.res2 + (long long)m * a.ldres2;
#pragma unroll
            for (int ob0 = 0; ob0 < TIO; ob0 += OBG) {
                // residual loads of up to OBG output blocks are issued together, ahead of the arithmetic; a missing
                // residual is predicated off per lane (no branch), its registers then hold zeros
                u32x4 rr1[OBG][NP], rr2[OBG][NP];
#pragma unroll
                for (int oo = 0; oo < OBG; ++oo)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int ncol = ncol0 + (ob0 + oo) * OBW + 16 * p;
                        const bool ok = ob0 + oo < TIO && mok && ncol < cout_out;
                        if constexpr (RESPRE) rr1[oo][p] = rpre[j][ob0 + oo < TIO ? ob0 + oo : 0][p];
                        else rr1[oo][p] = (ok && has1) ? *(const u32x4*)(r1row + ncol) : u32x4{0u, 0u, 0u, 0u};
                        rr2[oo][p] = (ok && has2) ? *(const u32x4*)(r2row + ncol) : u32x4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                for (int oo = 0; oo < OBG; ++oo) {
                    const int ob = ob0 + oo;
                    if (ob >= TIO) continue;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const f32x16& v = acc[ob][j];
                        float w[8], r[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned lo = __float_as_uint(v[8 * p + e]), hi = __float_as_uint(v[8 * p + 4 + e]);
                            swap32(lo, hi);
                            w[e] = __uint_as_float(lo); w[4 + e] = __uint_as_float(hi);
                        }
                        unpack8(rr1[oo][p], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] += r[e];
                        const int ncol = ncol0 + ob * OBW + 16 * p;
                        const bool ok = mok && ncol < cout_out;
                        if (ok) *(u32x4*)(orow + ncol) = pack8(w);
                        unpack8(rr2[oo][p], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] += r[e];
                        if (ok && has2) *(u32x4*)(orow2 + ncol) = pack8(w);
                    }
                }
            }
        }
    }
    }  // !SPLIT
    STAMP(4);
    if (last) break;
    }  // tiles
}


// ---------------------------------------------------------------- the same tile kernel on v_mfma_f32_16x16x32_bf16
// Same staging (LDS-DMA ring, swizzled tiles, counted waits), same tiles, same FLOPs and matrix-pipe cycles per K-step; the only
// difference is the MFMA shape: under load the chip holds a HIGHER CLOCK with the 16x16x32 shape than with 32x32x16
// (MI355X_MICROARCH.md, DVFS item 7; here, the K loop of the 32x32x16 kernel re-issued with this shape: -5 ... -10 % time on
// every deep-K layer, profiles/r03_mfma16_ab.txt).  Plain and split-K instances of the 64-deep K-step only; GEGLU, the
// LayerNorm / row-statistics forms and the 32-deep K-step stay on the kernel above.
//
// A 32 x 32 block of the wave's tile is four 16 x 16 sub-tiles (cs, ps): lane (p = lane % 16, g = lane / 16) holds couts
// 16 cs + 4 g + e of pixels 16 ps + p (e = 0..3).  Fragments: sub-tile row p, 16-byte chunk (4 ks + g) ^ swz(row) of the 64-deep
// stage (ks = 0, 1); swz depends on p alone ((row >> 1) & 7 with row = 32 i + 16 cs + p), and the 16 rows x 4 chunks that one
// ds_read_b128 phase covers fall into 16 different bank groups.  Epilogue: v_permlane16_swap of the (cs = 0, cs = 1) register
// pairs gives a lane 8 consecutive couts of one pixel -- couts 16 (g & 1) + 8 (g >> 1) .. + 7 -- i.e. one 16-byte store (and
// residual load) per pixel sub-tile and 32-cout block; a pixel's 64 bytes of a block are written by four adjacent lanes.
template <int WM, int WN, int TJ, int TI, int NS, bool SPLIT, bool PERSIST, bool KORD>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm16_kernel(const ConvArgs a) {
    static_assert(!(SPLIT && PERSIST), "split-K launches are never persistent");
    constexpr int BK = 64;
    constexpr int NT = 64 * WM * WN, BM = 32 * TJ * WM, BN = 32 * TI * WN;
    constexpr int CPW = BK / 8, RB = BK * 2, RPP = NT / CPW;
    constexpr int NX = BM / RPP, NW = BN / RPP, NL = NX + NW;
    constexpr int STAGE = (BM + BN) * RB;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows per DMA pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int p16 = lane & 15, g = lane >> 4;

    auto tile_of = [&](int vb) {
        const int nwg = a.tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    };
    int m0, n0;
    const int z = blockIdx.z;
    const __bf16* x1 = a.x1 + (long long)z * a.sx1;
    const __bf16* wp = a.w + (long long)z * a.sw;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)x1, 0, a.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, a.w_bytes, 0x00020000);

    const int lrow = tid / CPW, lchunk = tid % CPW;
    auto swz_of = [](int r) { return (r >> 1) & 7; };
    // KORD (3x3, no upsampling): K-steps walk (channel chunk, tap) instead of (tap, channel chunk) -- the nine taps of a 64-channel
    // chunk are consecutive steps, so the eight re-reads of a pixel's chunk hit L2 (the weight image keeps its (tap, cin) order:
    // only the walk changes).  The tap changes every step, so a tile keeps, per gathered row, the pixel index of tap (0, 0) and one
    // validity bit per tap: a step's gather addresses are one select + add per row.
    static_assert(!KORD || 9 * NX <= 64, "tap validity bits of the gathered rows fit one 64-bit mask");
    int xn[KORD ? 1 : NX], xiy0[KORD ? 1 : NX], xix0[KORD ? 1 : NX];
    unsigned xsw[NX], woff[NW];
    int pbase[KORD ? NX : 1];
    unsigned long long vmask = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) xsw[i] = (unsigned)((lchunk ^ swz_of(lrow + i * RPP)) << 3);
    auto setup_tile = [&](int vb) {
        const int tile = tile_of(vb);
        if (a.mfast == 2) {
            const int w = tile >> 5, r = tile & 31, wnc = a.tilesN / a.winN;
            const int mw = w / wnc, nw = w - mw * wnc;
            const int rm = r / a.winN;
            m0 = (mw * a.winM + rm) * BM;
            n0 = (nw * a.winN + (r - rm * a.winN)) * BN;
        } else {
            m0 = (a.mfast ? tile % a.tilesM : tile / a.tilesN) * BM;
            n0 = (a.mfast ? tile / a.tilesM : tile % a.tilesN) * BN;
        }
        if (KORD) vmask = 0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int m = m0 + lrow + i * RPP;
            int n = -1, iy0 = 0, ix0 = 0;
            if (m < a.M) {
                n = m / a.HWout;
                const int rem = m - n * a.HWout;
                const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
                iy0 = oy * a.stride - a.pad; ix0 = ox * a.stride - a.pad;
            }
            if constexpr (KORD) {
                pbase[i] = (n * a.Hin + iy0) * a.Win + ix0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                    const bool ok = n >= 0 && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
                    vmask |= (unsigned long long)ok << (t * NX + i);
                }
            } else {
                xn[i] = n; xiy0[i] = iy0; xix0[i] = ix0;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int r = lrow + i * RPP, n = n0 + r;
            woff[i] = n < a.Cout ? (unsigned)(n * a.ldw + ((lchunk ^ swz_of(r)) << 3)) * 2u : OOB;
        }
    };
    const int Hl = a.Hin * a.ups, Wl = a.Win * a.ups, ush = a.ups >> 1;
    int pix[NX];
    auto set_tap = [&](int tap) {
        if constexpr (KORD) {
            const int kr = (tap * 11) >> 5, ks = tap - 3 * kr, off = kr * a.Win + ks;
            const unsigned m = (unsigned)(vmask >> (tap * NX));
#pragma unroll
            for (int i = 0; i < NX; ++i) pix[i] = ((m >> i) & 1u) ? pbase[i] + off : -1;
        } else {
            const int kr = tap / a.R, ks = tap - kr * a.R;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int iy = xiy0[i] + kr, ix = xix0[i] + ks;
                const bool ok = xn[i] >= 0 && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
                pix[i] = ok ? (xn[i] * a.Hin + (iy >> ush)) * a.Win + (ix >> ush) : -1;
            }
        }
    };
    const int slice = SPLIT ? blockIdx.y : 0;
    const int kt_lo = SPLIT ? (int)((long long)a.KT * slice / a.splitk) : 0;
    const int KTl = SPLIT ? (int)((long long)a.KT * (slice + 1) / a.splitk) - kt_lo : a.KT;
    int tap_n, cc_n, kt_n;
    auto issue_slice = [&](int s, int lo, int hi) {
        char* xb = smem + s * STAGE + wave * 1024;
        char* wb = xb + BM * RB;
        const bool src2 = cc_n >= a.CT1;
        const unsigned co = (unsigned)((src2 ? cc_n - a.CT1 : cc_n) * BK);
        const unsigned ld = (unsigned)(src2 ? a.ldx2 : a.ldx1);
        const unsigned kb = (unsigned)kt_n * (unsigned)RB;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l < lo || l >= hi) continue;
            if (l < NX) {
                const unsigned vo = pix[l] < 0 ? OOB : ((unsigned)pix[l] * ld + co + xsw[l]) * 2u;
                if (!src2) buf_dma16(r1, xb + l * RPP * RB, vo);
                else buf_dma16(r2, xb + l * RPP * RB, vo);
            } else {
                const int i = l - NX;
                buf_dma16(rw, wb + i * RPP * RB, woff[i] == OOB ? OOB : woff[i] + kb);
            }
        }
    };
    auto advance = [&]() {
        if constexpr (KORD) {
            if (++tap_n == 9) { tap_n = 0; ++cc_n; }
            kt_n = tap_n * a.CT + cc_n;
        } else {
            ++kt_n;
            if (++cc_n == a.CT) { cc_n = 0; ++tap_n; }
        }
    };
    const int swz = (p16 >> 1) & 7;
    const int xrow_off = (wm * 32 * TJ + p16) * RB, wrow_off = (wn * 32 * TI + p16) * RB;
    static_assert(NS >= 2 && NL * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
    auto start_tile = [&](int vb) {
        setup_tile(vb);
        if constexpr (KORD) { cc_n = kt_lo / 9; tap_n = kt_lo - cc_n * 9; kt_n = tap_n * a.CT + cc_n; }
        else { tap_n = kt_lo / a.CT; cc_n = kt_lo - tap_n * a.CT; kt_n = kt_lo; }
        set_tap(tap_n);
#pragma unroll
        for (int pp = 0; pp < NS - 1; ++pp) {
            if (pp < KTl) {
                if (pp > 0 && (KORD || cc_n == 0)) set_tap(tap_n);
                issue_slice(pp, 0, NL);
                advance();
            }
        }
    };
    int vb = blockIdx.x;
    start_tile(vb);
    for (;;) {  // one trip per tile (a single trip unless PERSIST)
    const int cm0 = m0, cn0 = n0;
    __builtin_amdgcn_sched_barrier(0);
    // accumulators start at bias[cout] (+ the image's time-embedding row); sub-tile (cs, ps) of block (i, j), element e:
    // cout cn0 + wn 32 TI + 32 i + 16 cs + 4 g + e of pixel cm0 + wm 32 TJ + 32 j + 16 ps + p16
    f32x4 acc[TI][TJ][2][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int nb = cn0 + wn * 32 * TI + i * 32 + 16 * cs + 4 * g;
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (!SPLIT && a.bias && nb < a.Cout) bq = *(const f32x4*)(a.bias + nb);  // split-K: added by the reduce pass
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) acc[i][j][cs][ps] = bq;
        }
    if (!SPLIT && a.rowvec) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int m = cm0 + wm * 32 * TJ + j * 32 + 16 * ps + p16;
                const float* rv = a.rowvec + (long long)((m < a.M ? m : a.M - 1) / a.HWout) * a.ld_rowvec;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int cs = 0; cs < 2; ++cs) {
                        const int nb = cn0 + wn * 32 * TI + i * 32 + 16 * cs + 4 * g;
                        if (nb < a.Cout) {
                            const f32x4 t = *(const f32x4*)(rv + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][cs][ps][e] += t[e];
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if (KTl >= NS - 1 && vb == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int s = 0, sn = NS - 1;
    for (int kt = 0; kt < KTl; ++kt) {
        const bool more = kt + NS - 1 < KTl;
        const char* xb = smem + s * STAGE;
        const char* wb = xb + BM * RB;
        if (more && (KORD || cc_n == 0)) set_tap(tap_n);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // two 32-deep MFMA k-steps per 64-deep stage
            const int co = ((4 * ks + g) ^ swz) << 4;
            bf16x8 bfr[TJ][2];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) bfr[j][ps] = *(const bf16x8*)(xb + xrow_off + (j * 32 + 16 * ps) * RB + co);
#pragma unroll
            for (int cs = 0; cs < 2; ++cs) {
                bf16x8 af[TI];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8*)(wb + wrow_off + (i * 32 + 16 * cs) * RB + co);
                // the stage's DMAs go out within its first MFMA k-step (as in the 32x32x16 kernel: the first half of the stage)
                if (more && ks == 0) issue_slice(sn, (NL * cs) / 2, (NL * (cs + 1)) / 2);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps)
                            acc[i][j][cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j][ps], acc[i][j][cs][ps], 0, 0, 0);
            }
        }
        if (more) {
            advance();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (NS - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s = s + 1 == NS ? 0 : s + 1;
        sn = sn + 1 == NS ? 0 : sn + 1;
    }
    bool last = true;
    if constexpr (PERSIST) {
        vb += gridDim.x;
        last = vb >= a.tiles;
        if (!last) start_tile(vb);
    }
    // ---------------------------------------------------------------- epilogue
    __bf16* outp = a.out + (long long)z * a.so;
    if constexpr (SPLIT) {
        if (a.out32) {  // edge convolution with a handful of couts: fp32 NCHW, bias added here
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int m = cm0 + wm * 32 * TJ + j * 32 + 16 * ps + p16;
                    if (m < a.M) {
                        const int img = m / a.HWout, px = m - img * a.HWout;
#pragma unroll
                        for (int i = 0; i < TI; ++i)
#pragma unroll
                            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int co = cn0 + wn * 32 * TI + i * 32 + 16 * cs + 4 * g + e;
                                    if (co < a.cout_real)
                                        a.out32[((long long)img * a.cout_real + co) * a.HWout + px] =
                                            acc[i][j][cs][ps][e] + (a.bias ? a.bias[co] : 0.f);
                                }
                    }
                }
            return;
        }
        // raw fp32 partial sums, 16 B per lane (4 consecutive couts of its pixel); the reduce pass sums the slices in slice order
        float* wsl = a.ws + (long long)slice * a.M * a.Cout;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int m = cm0 + wm * 32 * TJ + j * 32 + 16 * ps + p16;
                if (m < a.M) {
                    float* wrow = wsl + (long long)m * a.Cout + cn0 + wn * 32 * TI + 4 * g;
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int cs = 0; cs < 2; ++cs)
                            if (cn0 + wn * 32 * TI + i * 32 + 16 * cs + 4 * g < a.Cout) *(f32x4*)(wrow + i * 32 + 16 * cs) = acc[i][j][cs][ps];
                }
            }
        return;
    }
    if constexpr (!SPLIT) {
    auto swap16 = [](unsigned& lo, unsigned& hi) {
        const auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    };
    if (a.act == 1) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][k >> 1][k & 1][e] = silu_f(acc[i][j][k >> 1][k & 1][e]);
    }
    if (a.scale != 1.0f) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][k >> 1][k & 1][e] *= a.scale;
    }
    const int ncol0 = cn0 + wn * 32 * TI + 16 * (g & 1) + 8 * (g >> 1);  // + 32 i: the lane's 8 couts of block i after the swap
    if (a.res1 == nullptr && a.out2 == nullptr) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int m = cm0 + wm * 32 * TJ + j * 32 + 16 * ps + p16;
                const bool mok = m < a.M;
                __bf16* orow = outp + (long long)m * a.ldo;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const f32x4 &v0 = acc[i][j][0][ps], &v1 = acc[i][j][1][ps];
                    unsigned d0 = pack_bf16x2(v0[0], v0[1]), d1 = pack_bf16x2(v0[2], v0[3]);
                    unsigned d2 = pack_bf16x2(v1[0], v1[1]), d3 = pack_bf16x2(v1[2], v1[3]);
                    swap16(d0, d2);
                    swap16(d1, d3);
                    const int ncol = ncol0 + 32 * i;
                    if (mok && ncol < a.Cout) *(u32x4*)(orow + ncol) = u32x4{d0, d1, d2, d3};
                }
            }
    } else {
        const bool has1 = a.res1 != nullptr, has2 = a.out2 != nullptr;
        constexpr int OBG = (PERSIST && TI * TJ >= 10) ? 1 : 2;  // output blocks whose residual loads are issued together
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int m = cm0 + wm * 32 * TJ + j * 32 + 16 * ps + p16;
                const bool mok = m < a.M;
                __bf16* orow = outp + (long long)m * a.ldo;
                __bf16* orow2 = a.out2 + (long long)m * a.ldo2;
                const __bf16* r1row = a.res1 + (long long)m * a.ldres1;
                const __bf16* r2row = a.res2 + (long long)m * a.ldres2;
#pragma unroll
                for (int ob0 = 0; ob0 < TI; ob0 += OBG) {
                    u32x4 rr1[OBG], rr2[OBG];
#pragma unroll
                    for (int oo = 0; oo < OBG; ++oo) {
                        const int ncol = ncol0 + 32 * (ob0 + oo);
                        const bool ok = ob0 + oo < TI && mok && ncol < a.Cout;
                        rr1[oo] = (ok && has1) ? *(const u32x4*)(r1row + ncol) : u32x4{0u, 0u, 0u, 0u};
                        rr2[oo] = (ok && has2) ? *(const u32x4*)(r2row + ncol) : u32x4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int oo = 0; oo < OBG; ++oo) {
                        const int i = ob0 + oo;
                        if (i >= TI) continue;
                        float w[8], r[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned lo = __float_as_uint(acc[i][j][0][ps][e]), hi = __float_as_uint(acc[i][j][1][ps][e]);
                            swap16(lo, hi);
                            w[e] = __uint_as_float(lo); w[4 + e] = __uint_as_float(hi);
                        }
                        unpack8(rr1[oo], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] += r[e];
                        const int ncol = ncol0 + 32 * i;
                        const bool ok = mok && ncol < a.Cout;
                        if (ok) *(u32x4*)(orow + ncol) = pack8(w);
                        unpack8(rr2[oo], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] += r[e];
                        if (ok && has2) *(u32x4*)(orow2 + ncol) = pack8(w);
                    }
                }
            }
    }
    }  // !SPLIT
    if (last) break;
    }  // tiles
}

// split-K second pass: out = scale * act(bias + time-embedding row + sum over slices) + res1 (and out2 = that + res2).
// 8 columns per thread, slices summed in index order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a) {
    const int cpr = a.Cout / 8;
    const long long total = (long long)a.M * cpr;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx / cpr), c = (int)(idx - (long long)m * cpr) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (a.bias) {
            const f32x4 b0 = *(const f32x4*)(a.bias + c), b1 = *(const f32x4*)(a.bias + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = b0[e]; v[4 + e] = b1[e]; }
        }
        if (a.rowvec) {
            const float* rv = a.rowvec + (long long)(m / a.HWout) * a.ld_rowvec + c;
            const f32x4 t0 = *(const f32x4*)rv, t1 = *(const f32x4*)(rv + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += t0[e]; v[4 + e] += t1[e]; }
        }
        for (int sl = 0; sl < a.splitk; ++sl) {
            const float* wp = a.ws + ((long long)sl * a.M + m) * a.Cout + c;
            const f32x4 p0 = *(const f32x4*)wp, p1 = *(const f32x4*)(wp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += p0[e]; v[4 + e] += p1[e]; }
        }
        if (a.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a.scale;
        if (a.res1) {
            float r[8];
            unpack8(*(const u32x4*)(a.res1 + (long long)m * a.ldres1 + c), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        *(u32x4*)(a.out + (long long)m * a.ldo + c) = pack8(v);
        if (a.out2) {
            float r[8];
            unpack8(*(const u32x4*)(a.res2 + (long long)m * a.ldres2 + c), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
            *(u32x4*)(a.out2 + (long long)m * a.ldo2 + c) = pack8(v);
        }
    }
}

// ---------------------------------------------------------------- pointwise layers with 320 / 640 input channels
// o-proj (+ residual), proj_in / proj_out (+ residual), attn2.to_q, qkv, ff.linear1 + GEGLU at the 64x64 (320 channels) and
// 32x32 (640 channels) levels: K = 320 gives 160 FLOP per byte, far below the machine balance, and in the tiled kernel a CU
// alternates between a K loop (HBM idle) and an epilogue (matrix core idle): 2.2-3.0 TB/s of algorithmic traffic.  Here the
// WEIGHTS never move: a block is WAVES waves, wave w keeps the A fragments of couts 32 w .. 32 w + 31 of the block's cout
// group for the whole K in registers (80 at K = 320, ten waves = 320 couts; 160 at K = 640, four waves = 128 couts), and the
// block streams its row range in chunks of 32 rows through a ring of NSTG LDS stages (LDS-DMA, NSTG - 1 chunks ahead).  Per
// chunk a wave does K / 16 MFMAs (B fragments = the chunk's rows, shared by all waves), adds the residual rows it prefetched
// one chunk earlier and stores 32 x 32 outputs -- loads, MFMAs and stores of neighbouring chunks overlap all the time, there
// is no tile epilogue.  Results are bit-identical to the tiled kernel (same k order, accumulators start at the bias).
// Cout = GW G (GW = 32 WAVES couts per block): block id = xcd + 8 g + 8 G rr handles cout group g of row range 8 rr + xcd, so
// the G blocks of a range sit on one XCD, are dispatched together and move at the same pace: the range's rows come from
// HBM once and from that XCD's L2 for the other G - 1 groups.
// GEGLU: the wave's 32 weight rows are 8-row value / gate blocks (ops.geglu_interleave); it writes 16 outputs per pixel.
constexpr int PW_ROWS = 32;
#ifndef PW_DIAG
#define PW_DIAG 0  // timing experiments (tools/pw_diag.sh): bit 0 no output stores, bit 1 no MFMAs, bit 2 the row DMAs fetch nothing
#endif
struct PwArgs {
    const __bf16* x; const __bf16* w; const float* bias; const __bf16* res; __bf16* out;
    int M, ldx, ldw, ldres, ldo, chunks, chunks_per_block;
    int G;  // cout groups (Cout <= 32 WAVES G): G blocks walk the same row range, one group each
    int per, total;  // per > 0: the one-round layout for a G that does not divide an XCD's CUs (see try_pw320)
    int Cout;  // (GEGLU: rows of w; the last group may be partial: its waves beyond Cout only stage rows and meet the barriers)
    unsigned x_bytes;
    const float* colsum; float ln_eps;  // LNF
};

// LNF: the LayerNorm in front of the layer, folded (ops.fold_layernorm: w = gamma o W, bias = bias + W beta, colsum = row sums
// of w) with the row statistics taken HERE, from fragments the waves read anyway: wave w sums (shifted by the row's first
// element) the two 16-channel k-steps 2 w, 2 w + 1 of the chunk's 32 rows into an LDS table, no cross-lane reduction; one
// iteration later (the barrier in between publishes the table) every lane adds the 2 x WAVES partials of its pixel and turns
// the raw products of that chunk into rstd (acc - mean colsum) + bias.  The normalised rows never exist in HBM.
template <int C, int WAVES, int NSTG, bool RES, bool GEGLU, bool LNF = false>
__global__ __launch_bounds__(64 * WAVES, 1) void pw_stream_kernel(const PwArgs a) {
    static_assert(!(RES && GEGLU), "GEGLU layers carry no residual");
    static_assert(!LNF || (!RES && C / 16 == 2 * WAVES), "LayerNorm-fed layers carry no residual; two k-steps per wave");
    static_assert(C == 320 || C == 640, "row swizzle below: 40 or 80 16-byte units per row");
    constexpr int ROWB = C * 2, CHUNKB = PW_ROWS * ROWB, UPR = C / 8;      // bytes per row / chunk, 16-byte units per row
    constexpr int PIECES = CHUNKB / 1024, PPW = PIECES / WAVES;             // 1-KiB DMA pieces per chunk / per wave
    static_assert(PIECES % WAVES == 0, "DMA pieces divide over the waves");
    constexpr int KS = C / 16, GW = 32 * WAVES;
    extern __shared__ __attribute__((aligned(1024))) char smem[];           // NSTG chunks
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    int bt = blockIdx.x >> 3, grp = bt % a.G, range = (bt / a.G) * 8 + (blockIdx.x & 7);
    if (a.per) {  // one round of floor(CUs / G) ranges: XCD x takes the (range, group) pairs x per .. x per + per - 1
        bt = (blockIdx.x & 7) * a.per + bt;
        if (bt >= a.total) return;  // (whole block: no barrier is pending)
        range = bt / a.G;
        grp = bt - range * a.G;
    }
    const int c_lo = range * a.chunks_per_block;
    const int c_hi = min(c_lo + a.chunks_per_block, a.chunks);
    if (c_lo >= c_hi) return;  // (whole block: no barrier is pending)
    const int n0 = GW * grp;   // first cout of the block's group
    const bool wact = n0 + 32 * wave < a.Cout;  // wave-uniform
    // LNF tables behind the ring: colsum[GW] | bias[GW] | partial sums [2][WAVES][64] (s, q) | first elements [2][32]
    float* cbs = (float*)(smem + NSTG * CHUNKB);
    float* ptab = cbs + 2 * GW;
    float* x0tab = ptab + 2 * WAVES * 64 * 2;
    if constexpr (LNF) {
        for (int t = tid; t < GW; t += 64 * WAVES) {
            cbs[t] = a.colsum[n0 + t];
            cbs[GW + t] = a.bias ? a.bias[n0 + t] : 0.f;
        }
    }

    // the wave's weight fragments: lane holds W[n0 + 32 wave + l31][16 ks + 8 hh .. + 7]
    bf16x8 wf[KS];
    {
        const __bf16* wp = a.w + (long long)(wact ? n0 + 32 * wave + l31 : 0) * a.ldw + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(wp + 16 * ks));
    }
    // accumulator start: bias of couts n0 + 32 wave + 8 q + 4 hh + e
    f32x16 bq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = (a.bias && wact && !LNF) ? *(const f32x4*)(a.bias + n0 + 32 * wave + 8 * q + 4 * hh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[4 * q + e] = t[e];
    }
    __builtin_amdgcn_sched_barrier(0);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    // Row swizzle: the unit at position p of row r is unit p ^ swz(r) of that row, swz(r) = (r >> 1) & 7 for 40 units per row
    // (rows alternate between the two halves of the 16 bank groups by themselves), + ((r & 1) << 3) for 80 (they do not):
    // the 16 rows that one ds_read_b128 phase covers hit 16 different bank groups.
    auto swz = [](int r) { return C == 320 ? ((r >> 1) & 7) : (((r >> 1) & 7) | ((r & 1) << 3)); };
    unsigned dvo[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int u = (PPW * wave + i) * 64 + lane, r = u / UPR, pp = u - r * UPR;
        dvo[i] = (unsigned)(r * a.ldx + (pp ^ swz(r)) * 8) * 2u;
    }
    const unsigned chunk_b = (unsigned)(PW_ROWS * a.ldx) * 2u;
    auto issue_chunk = [&](int c) {
        // (a chunk past the range, or rows past M: the descriptor's range check returns zeros, nothing is skipped, so every
        // iteration issues the same number of memory operations and the counted waits below hold)
        char* base = smem + (c % NSTG) * CHUNKB + wave * (PPW * 1024);
        const unsigned long long off = (unsigned long long)c * chunk_b;
        const bool ok = c < c_hi && off < 0x7fffffffull;
#pragma unroll
        for (int i = 0; i < PPW; ++i) buf_dma16(rx, base + i * 1024, (ok && !(PW_DIAG & 4)) ? dvo[i] + (unsigned)off : OOB);
    };
    // fragment read: row l31, unit (2 ks + hh) ^ swz(l31) = (2 ks) ^ (hh ^ swz(l31)): a lane constant XOR a literal
    const int xrow = l31 * ROWB, xsw = (hh ^ swz(l31)) << 4;

    // after the permlane32 swap a lane owns couts col0 + 16 p .. + 7 of its pixel (GEGLU: outputs col0 .. + 7, half as many)
    const int col0 = GEGLU ? (n0 + 32 * wave) / 2 + 8 * hh : n0 + 32 * wave + 8 * hh;
    auto res_load = [&](int c, u32x4* r) {
        const long long m = (long long)c * PW_ROWS + l31;
        const bool ok = RES && wact && c < c_hi && m < a.M;
        const __bf16* rp = a.res + (ok ? m : 0) * a.ldres + col0;
        r[0] = ok ? *(const u32x4*)rp : u32x4{0u, 0u, 0u, 0u};
        r[1] = ok ? *(const u32x4*)(rp + 16) : u32x4{0u, 0u, 0u, 0u};
    };
    auto swap32 = [](unsigned& lo, unsigned& hi) {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    };

    u32x4 rc[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    if constexpr (RES) res_load(c_lo, rc);
#pragma unroll
    for (int i = 0; i < NSTG - 1; ++i) issue_chunk(c_lo + i);
    // all landed: the counted wait of the loop assumes NSTG - 2 whole iterations of younger operations behind the DMA it waits for
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (lgkmcnt: the LNF tables written above)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // writes the 32 x 32 outputs of one chunk from their accumulators (values after bias / LayerNorm fold)
    auto store_chunk = [&](const f32x16& acc, int c) {
        const long long m = (long long)c * PW_ROWS + l31;
        __bf16* orow = a.out + m * a.ldo + col0;
        if constexpr (GEGLU) {
            // quads (0, 1) -> outputs 4 hh + e, quads (2, 3) -> outputs 8 + 4 hh + e of the wave's 16 (conv_gemm_kernel's form)
            float r8[8];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 r = geglu_pk(f32x2{acc[8 * h2 + e], acc[8 * h2 + e + 1]}, f32x2{acc[8 * h2 + 4 + e], acc[8 * h2 + 5 + e]});
                    r8[4 * h2 + e] = r[0];
                    r8[4 * h2 + e + 1] = r[1];
                }
            unsigned d0 = pack_bf16x2(r8[0], r8[1]), d1 = pack_bf16x2(r8[2], r8[3]);
            unsigned d2 = pack_bf16x2(r8[4], r8[5]), d3 = pack_bf16x2(r8[6], r8[7]);
            swap32(d0, d2);
            swap32(d1, d3);
            if (wact && m < ((PW_DIAG & 1) ? -a.M : a.M)) *(u32x4*)orow = u32x4{d0, d1, d2, d3};
        } else {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                float w8[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned lo = __float_as_uint(acc[8 * p2 + e]), hi = __float_as_uint(acc[8 * p2 + 4 + e]);
                    swap32(lo, hi);
                    w8[e] = __uint_as_float(lo); w8[4 + e] = __uint_as_float(hi);
                }
                if constexpr (RES) {
                    float r8[8];
                    unpack8(rc[p2], r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) w8[e] += r8[e];
                }
                if (wact && m < ((PW_DIAG & 1) ? -a.M : a.M)) *(u32x4*)(orow + 16 * p2) = pack8(w8);
            }
        }
    };
    // LNF: raw products of chunk c -> rstd (acc - mean colsum) + bias, then the stores
    auto finish_ln = [&](f32x16 acc, int c) {
        const float* pt = ptab + (c & 1) * (WAVES * 64 * 2);
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x2 p0 = *(const f32x2*)(pt + (w * 64 + l31) * 2), p1 = *(const f32x2*)(pt + (w * 64 + 32 + l31) * 2);
            sm += p0[0] + p1[0];
            sq += p0[1] + p1[1];
        }
        sm *= 1.0f / C;
        sq *= 1.0f / C;
        const float mean = x0tab[(c & 1) * PW_ROWS + l31] + sm;
        const float rstd = rsqrtf(fmaxf(sq - sm * sm, 0.f) + a.ln_eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 cs = *(const f32x4*)(cbs + 32 * wave + 8 * q + 4 * hh);
            const f32x4 bb = *(const f32x4*)(cbs + GW + 32 * wave + 8 * q + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = fmaf(rstd, acc[4 * q + e], fmaf(nmr, cs[e], bb[e]));
        }
        store_chunk(acc, c);
    };

    f32x16 acc_prev = {};
    for (int c = c_lo; c < c_hi; ++c) {
        // memory operations of an iteration, in issue order: residual rows of chunk c + 1 (RES), DMA of chunk c + NSTG - 1, stores
        // (LNF: of chunk c - 1).  The residual is issued first so that waiting for it (next iteration) never waits for the newest DMAs.
        u32x4 rn[2];
        if constexpr (RES) res_load(c + 1, rn);
        issue_chunk(c + NSTG - 1);
        const char* xb = smem + (c % NSTG) * CHUNKB + xrow;
        if constexpr (LNF) {
            // the previous chunk first: its accumulators die before this chunk's are born (168 registers at 10 waves)
            if (c > c_lo) finish_ln(acc_prev, c - 1);
            // shift = the row's first element (unit 0 sits at position swz(row)); this wave's share: k-steps 2 wave, 2 wave + 1
            const float x0 = bf16_bits_to_f32(*(const unsigned short*)(xb + (swz(l31) << 4)));
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float f[8];
                unpack8(*(const u32x4*)(xb + (((2 * (2 * wave + i)) << 4) ^ xsw)), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = f[e] - x0;
                    sm += d;
                    sq = fmaf(d, d, sq);
                }
            }
            *(f32x2*)(ptab + ((c & 1) * WAVES * 64 + wave * 64 + lane) * 2) = f32x2{sm, sq};
            if (wave == 0 && hh == 0) x0tab[(c & 1) * PW_ROWS + l31] = x0;
        }
        f32x16 acc = bq;
        if constexpr (LNF) acc = f32x16{};  // (the bias joins after the normalisation)
        if constexpr (C == 640 && !(PW_DIAG & 2)) {
            // 160 weight registers leave hipcc no room it would use for a read-ahead by itself: it reads each row fragment right
            // before its MFMA, and two waves per SIMD do not cover an LDS round trip per MFMA.  Pinned order: the fragment of
            // k-step ks + PF is read in the slot of MFMA ks.
            constexpr int PF = 4;
            bf16x8 xq[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) xq[i] = *(const bf16x8*)(xb + (((2 * i) << 4) ^ xsw));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xf = xq[ks % PF];
                if (ks + PF < KS) xq[ks % PF] = *(const bf16x8*)(xb + (((2 * (ks + PF)) << 4) ^ xsw));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xf = *(const bf16x8*)(xb + (((2 * ks) << 4) ^ xsw));
                if constexpr (PW_DIAG & 2) acc[ks & 15] += __builtin_bit_cast(f32x4, xf)[ks & 3];
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf, acc, 0, 0, 0);
            }
        }
        constexpr int ST = (PW_DIAG & 1) ? 0 : GEGLU ? 1 : 2, PER = (RES ? 2 : 0) + PPW + ST;
        if constexpr (LNF) {
            acc_prev = acc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this chunk's partial sums are in LDS before the barrier
            // (the first iterations carry fewer stores than the count below assumes)
            if (c < c_lo + NSTG - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST + (NSTG - 2) * PER) : "memory");
        } else {
            store_chunk(acc, c);
            if constexpr (RES) { rc[0] = rn[0]; rc[1] = rn[1]; }
            // chunk c + 1 must have landed.  Younger than its DMAs (issued NSTG - 2 iterations ago): that iteration's stores and
            // NSTG - 2 whole iterations of (residual loads +) DMAs + stores
            // (a wave beyond Cout in a partial last group stores nothing: its count is the DMAs alone; no such group with RES)
            if (wact) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST + (NSTG - 2) * PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * PPW) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if constexpr (LNF) finish_ln(acc_prev, c_hi - 1);
}

// ---------------------------------------------------------------- profiling hooks
struct ProfRec { hipEvent_t e0, e1; double flops; int M, Cout, K, R, flags; };
bool g_prof_on = false;       // benchmark hook (bench.py): while on, launches are serialised on g_prof_mu
std::vector<ProfRec> g_prof;
size_t g_prof_used = 0;
std::mutex g_prof_mu;

int g_force_cfg = -1;  // tuning: force a tile configuration (index into the dispatch below)
int g_force_splitk = 0;  // tuning: force the split-K factor (0 = heuristic)
int g_order = -1;     // tuning: force the tile order (0 N-fastest, 1 M-fastest, -1 = heuristic)
#ifdef CCNET_STAMP
unsigned long long* g_dbg = nullptr;
#endif
int g_dbg_zero = 0;    // timing experiments only: bit0 = x descriptors empty, bit1 = w descriptor empty

constexpr int kNumCUs = 256;
int g_persist = 1;  // tuning: 0 = persistent configurations launch one block per tile

int g_pw_ranges = 0;  // tuning (ccnet_set_option key 11): row ranges of the many-group streaming launches (0 = 64)
int g_pw320 = 31;  // tuning (ccnet_set_option key 8): 0 = these layers run on the tiled kernel; bit 1: several cout groups too;
                   // bit 2: GEGLU too; bit 3: 640 input channels too (8 waves x 160 weight registers, 256 couts per block);
                   // bit 4: LayerNorm statistics in the kernel (colsum without rowstat)

// CCNET_OK + *done = true if the problem is one of these layers and was launched here
template <int C, int WAVES, int NSTG, bool RES, bool GEGLU, bool LNF = false>
int launch_pw(const PwArgs& a, int grid, hipStream_t st) {
    constexpr int LDS = NSTG * PW_ROWS * C * 2 + (LNF ? (2 * 32 * WAVES + 2 * WAVES * 64 * 2 + 2 * PW_ROWS) * 4 : 0);
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)pw_stream_kernel<C, WAVES, NSTG, RES, GEGLU, LNF>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return CCNET_ELAUNCH;
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((pw_stream_kernel<C, WAVES, NSTG, RES, GEGLU, LNF>), dim3(grid), dim3(64 * WAVES), LDS, st, a);
    return hipGetLastError() == hipSuccess ? CCNET_OK : CCNET_ELAUNCH;
}

int try_pw320(const ccnet_conv_params* p, int M, hipStream_t st, bool* done) {
    *done = false;
    if (!g_pw320 || g_force_cfg >= 0 || g_force_splitk) return CCNET_OK;
    if (p->R != 1 || p->C2 != 0 || p->stride != 1 || p->ups != 1 || p->pad != 0) return CCNET_OK;
    // 640 input channels (32x32 level): eight waves of 160 weight registers = 256 couts per block, ring of three 40 KB stages.
    // Pays where many cout groups share the rows (qkv 1920: 101-109 -> 83 us, ff.linear1 + GEGLU 5120: 256-273 -> 226 us); for
    // the 640-cout layers it does not (proj 44 -> 47 us, o-proj + res 43 -> 42: 2.5 groups padded to 3, 32768 rows per 328 KB
    // weight prologue), nor did a four-wave form with 128 couts per block (one wave per SIMD): those stay on the tiled kernel.
    const bool c640 = p->C1 == 640 && (g_pw320 & 8) && p->Cout >= 1280 && ((p->Cout % 256) == 0 || ((p->Cout % 32) == 0 && !p->res1));
    if (p->C1 != 320 && !c640) return CCNET_OK;
    const int GW = c640 ? 256 : 320;  // couts per block
    if ((p->Cout % GW) && !c640) return CCNET_OK;  // (a partial last group: 640-channel instance without residual only)
    const int G = (p->Cout + GW - 1) / GW;
    if (G > 32 || (G > 1 && !(g_pw320 & 2))) return CCNET_OK;
    if ((p->act != 0 && p->act != 2) || (p->act == 2 && (p->res1 || !(g_pw320 & 4)))) return CCNET_OK;
    if (p->scale != 1.0f || p->rowvec || p->out2 || p->rowstat || p->stat_out || p->batch > 1) return CCNET_OK;
    const bool lnf = p->colsum != nullptr;  // (without rowstat: the row statistics are taken in the kernel)
    if (lnf && (p->res1 || p->act != 0 || c640 || !(g_pw320 & 16))) return CCNET_OK;
    if (p->Hin != p->Hout || p->Win != p->Wout || (long long)M * p->C1 < 65536ll * 320) return CCNET_OK;
    const long long xb = ((long long)M - 1) * p->ldx1 * 2 + p->C1 * 2;
    if (xb >= 0x7fffffffLL) return CCNET_OK;
    PwArgs a{};
    a.x = (const __bf16*)p->x1; a.w = (const __bf16*)p->w; a.bias = p->bias; a.res = (const __bf16*)p->res1; a.out = (__bf16*)p->out;
    a.M = M; a.ldx = p->ldx1; a.ldw = p->ldw; a.ldres = p->ldres1; a.ldo = p->ldo; a.G = G; a.Cout = p->Cout;
    a.chunks = (M + PW_ROWS - 1) / PW_ROWS;
    // row ranges: a multiple of 8 (one per XCD slot), ranges x G blocks fill one round of the chip
    // (one round of 8 floor(256 / 8 G) x G blocks measured 2-4 % faster than 256 or 128 ranges in three or four rounds: fewer
    // weight prologues)
    int ranges = 8 * (kNumCUs / (8 * G));
    int grid = ranges * G;
    if (ranges * G < 208) {
        // G does not fill the XCDs in units of 8 ranges (20 groups at ff.linear1 + GEGLU of the 32x32 level): still ONE round --
        // floor(CUs / G) ranges, the (range, group) pairs dealt to the XCDs in runs, so that a range's blocks sit on one XCD
        // (a run boundary splits every third range over two).  Rounds of 64 ranges cost a weight prologue per round and block:
        // 236 us (five rounds) / 200 us (24 ranges, two rounds) / 228 us on 160 CUs (8 ranges, one round) at 32768 x 5120 x 640.
        ranges = g_pw_ranges > 0 ? g_pw_ranges : kNumCUs / G;
        if (g_pw_ranges > 0 && (ranges % 8) == 0) {
            grid = ranges * G;  // (the multi-round form, for A/B)
        } else {
            a.total = ranges * G;
            a.per = (a.total + 7) / 8;
            grid = 8 * a.per;
        }
    }
    a.chunks_per_block = (a.chunks + ranges - 1) / ranges;
    a.x_bytes = (unsigned)xb;
    a.colsum = p->colsum; a.ln_eps = p->ln_eps;
    const int rc = lnf ? launch_pw<320, 10, 4, false, false, true>(a, grid, st)
                 : c640 ? (p->act == 2 ? launch_pw<640, 8, 3, false, true>(a, grid, st)
                           : p->res1 ? launch_pw<640, 8, 3, true, false>(a, grid, st) : launch_pw<640, 8, 3, false, false>(a, grid, st))
                 : p->act == 2 ? launch_pw<320, 10, 4, false, true>(a, grid, st)
                 : p->res1 ? launch_pw<320, 10, 4, true, false>(a, grid, st) : launch_pw<320, 10, 4, false, false>(a, grid, st);
    if (rc != CCNET_OK) return rc;
    *done = true;
    return CCNET_OK;
}


template <int WM, int WN, int TJ, int TI, bool GEGLU, int NS = 2, int BK = 64, bool SPLIT = false, bool PERSIST = false,
          bool RESPRE = false, bool STAT = false>
int launch(const ConvArgs& a, int batch, hipStream_t st) {
    constexpr int BM = 32 * TJ * WM, BN = 32 * TI * WN, NT = 64 * WM * WN;
    constexpr int LDS = NS * (BM + BN) * BK * 2;
    // once per instantiation and process, safe under concurrent first calls from several host threads (the attribute is a
    // property of the function, idempotent to set; the flag is only an optimisation)
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv_gemm_kernel<WM, WN, TJ, TI, GEGLU, NS, BK, SPLIT, PERSIST, RESPRE, STAT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return CCNET_ELAUNCH;
        attr_set.store(true, std::memory_order_release);
    }
    ConvArgs b = a;
    if (STAT) b.stat_parts = a.Cout / (32 * TI);
    b.CT1 = a.C1 / BK; b.CT = (a.C1 + a.C2) / BK; b.KT = a.R * a.R * b.CT;
    b.tilesN = (a.Cout + BN - 1) / BN;
    b.tilesM = (a.M + BM - 1) / BM;
    b.tiles = b.tilesM * b.tilesN;
    // windowed order wherever an XCD's 32 tiles would otherwise be 32 different weight panels (>= 8 cout tiles): measured on
    // ff.linear1 of the 16x16 level (8192 x 10240 x 1280) 940 -> 963 TFLOP/s, with GEGLU 1039 -> 1075 (tools/bench_conv.py --cold
    // --order 2 against --order -1, same box); g_order 0 / 1 force the N- / M-fastest walks
    if (g_order == 2 || (g_order < 0 && b.tilesN >= 8)) {
        b.winN = (b.tilesN % 8) == 0 ? 8 : (b.tilesN % 4) == 0 ? 4 : (b.tilesN % 2) == 0 ? 2 : 1;
        b.winM = 32 / b.winN;
        b.mfast = ((b.tilesM % b.winM) == 0 && (b.tiles % 256) == 0) ? 2 : 0;
    }
    // persistent: one resident block per CU (these configurations take > 80 KB of LDS), a multiple of 8 blocks
    // (measured: +5..9 % on the K <= 1280 GEGLU / 1x1 layers, nothing beyond noise on deeper K, where a tile's
    // start-up is a small share and the hardware's dynamic block placement balances ragged tiles better)
    const int kmax = GEGLU ? 1280 : 640;
    const int grid = PERSIST && g_persist && b.KT * BK <= (g_persist > 1 ? g_persist : kmax) && b.tiles > kNumCUs ? kNumCUs : b.tiles;
    hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, TJ, TI, GEGLU, NS, BK, SPLIT, PERSIST, RESPRE, STAT>), dim3(grid, SPLIT ? b.splitk : 1, batch),
                       dim3(NT), LDS, st, b);
    CCNET_LAUNCH_CHECK();
    if (SPLIT && !b.out32) {
        const long long items = (long long)b.M * (b.Cout / 8);
        const int grid = (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, b);
        CCNET_LAUNCH_CHECK();
    }
    return CCNET_OK;
}

template <int WM, int WN, int TJ, int TI, int NS = 2, bool SPLIT = false, bool PERSIST = false, bool KORD = false>
int launch16(const ConvArgs& a, int batch, hipStream_t st) {
    constexpr int BM = 32 * TJ * WM, BN = 32 * TI * WN, NT = 64 * WM * WN, BK = 64;
    constexpr int LDS = NS * (BM + BN) * BK * 2;
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv_gemm16_kernel<WM, WN, TJ, TI, NS, SPLIT, PERSIST, KORD>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return CCNET_ELAUNCH;
        attr_set.store(true, std::memory_order_release);
    }
    ConvArgs b = a;
    b.CT1 = a.C1 / BK; b.CT = (a.C1 + a.C2) / BK; b.KT = a.R * a.R * b.CT;
    b.tilesN = (a.Cout + BN - 1) / BN;
    b.tilesM = (a.M + BM - 1) / BM;
    b.tiles = b.tilesM * b.tilesN;
    if (g_order == 2 || (g_order < 0 && b.tilesN >= 8)) {  // windowed tile walk, as launch<>
        b.winN = (b.tilesN % 8) == 0 ? 8 : (b.tilesN % 4) == 0 ? 4 : (b.tilesN % 2) == 0 ? 2 : 1;
        b.winM = 32 / b.winN;
        b.mfast = ((b.tilesM % b.winM) == 0 && (b.tiles % 256) == 0) ? 2 : 0;
    }
    // split-K 3x3 layers of the 8x8 level (<= 8 pixel panels): M-fastest, so that an XCD streams its share of the weights instead of
    // all of them (the layer is weight-bound): 318 -> 207 MB per launch across the L2 boundary, -2 % time (profiles/r03_korder.txt)
    // (and the 16x16-level ones where they run the channel-major K walk, see conv_gemm_impl)
    if (SPLIT && g_order < 0 && a.R == 3 && (b.tilesM <= 8 || KORD)) b.mfast = 1;
    const int grid = PERSIST && g_persist && b.KT * BK <= (g_persist > 1 ? g_persist : 640) && b.tiles > kNumCUs ? kNumCUs : b.tiles;
    hipLaunchKernelGGL((conv_gemm16_kernel<WM, WN, TJ, TI, NS, SPLIT, PERSIST, KORD>), dim3(grid, SPLIT ? b.splitk : 1, batch), dim3(NT), LDS, st, b);
    CCNET_LAUNCH_CHECK();
    if (SPLIT && !b.out32) {
        const long long items = (long long)b.M * (b.Cout / 8);
        const int rgrid = (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, st, b);
        CCNET_LAUNCH_CHECK();
    }
    return CCNET_OK;
}

typedef int (*LaunchFn)(const ConvArgs&, int, hipStream_t);
struct TileCfg { int bm, bn; LaunchFn fn, fn_geglu, fn_split; LaunchFn fn_stat = nullptr; int stat_span = 0; };
// Indices are stable (tools/bench_conv.py --cfgs): entries that lost every measurement were removed and stay empty.
const TileCfg kCfgs[] = {
    {128, 128, launch<2, 2, 2, 2, false>, launch<2, 2, 2, 2, true>, nullptr},   // 0: 4 waves, two blocks per CU
    {128, 64, launch<2, 1, 2, 2, false>, launch<2, 1, 2, 2, true>, nullptr},    // 1: 2 waves
    {64, 64, launch<1, 1, 2, 2, false>, launch<1, 1, 2, 2, true>, nullptr},     // 2: 1 wave, tiny problems
    {128, 320, launch<2, 2, 2, 5, false>, nullptr, launch<2, 2, 2, 5, false, 2, 64, true>},  // 3: 4 waves, Cout = 320 without waste
    {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr},  // 4-6
    {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr},  // 7-9
    {256, 256, launch<4, 2, 2, 4, false>, nullptr, nullptr},                     // 10: 8 waves, 2 x 64-deep stages
    {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr},        // 11-12
    {128, 128, launch<2, 2, 2, 2, false, 3>, nullptr, nullptr},                  // 13: 4 waves, 3 stages
    {0, 0, nullptr, nullptr, nullptr},                                           // 14
    {256, 128, launch<4, 2, 2, 2, false, 3>, nullptr, launch<4, 2, 2, 2, false, 3, 64, true>},  // 15: 8 waves, 3 stages
    {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr},  // 16-18
    {128, 128, launch<4, 2, 1, 2, false, 3>, nullptr, launch<4, 2, 1, 2, false, 3, 64, true>},  // 19: 8 waves, 3 stages
    {256, 256, launch<4, 2, 2, 4, false, 4, 32, false, true>, launch<4, 2, 2, 4, true, 4, 32, false, true>, nullptr},    // 20: 8 waves, 4 x 32-deep
    {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr},        // 21-22
    {256, 320, launch<4, 2, 2, 5, false, 2, 64, false, true>, launch<4, 2, 2, 5, true, 2, 64, false, true>,
     launch<4, 2, 2, 5, false, 2, 64, true>, launch<4, 2, 2, 5, false, 2, 64, false, false, false, true>, 160},  // 23: 8 waves (the statistics instance is not persistent: no registers left for the next tile's gather state)
    {0, 0, nullptr, nullptr, nullptr}, {0, 0, nullptr, nullptr, nullptr},        // 24-25
    {128, 256, launch<2, 2, 2, 4, false, 3, 32>, launch<2, 2, 2, 4, true, 3, 32>, nullptr},    // 26: 4 waves, 72 KB
    {128, 320, launch<2, 2, 2, 5, false, 2, 32>, nullptr, launch<2, 2, 2, 5, false, 2, 32, true>},  // 27: 4 waves, 57 KB
    {128, 128, launch<2, 2, 2, 2, false, 3, 32>, nullptr, nullptr},              // 28: 48 KB: three blocks per CU
    {128, 320, launch<4, 2, 1, 5, false, 2, 64>, nullptr, nullptr,
     launch<4, 2, 1, 5, false, 2, 64, false, false, false, true>, 160},          // 29: 8 waves of 32 x 160, 114 KB
    {128, 320, launch<4, 2, 1, 5, false, 2, 64, false, true, true>, nullptr, nullptr,
     launch<4, 2, 1, 5, false, 2, 64, false, true, true, true>, 160},            // 30: 29, persistent, residual prefetch
#ifdef CCNET_EXP_TILES
    // diagnostic build only (tools/build_variant.sh exp "-DCCNET_EXP_TILES" gemm_conv): one wave per SIMD, 128 x 160 /
    // 128 x 128 wave tiles (0.45 / 0.5 LDS fragment reads per MFMA instead of 0.7 / 0.75); hipcc spills in their epilogues
    {256, 320, launch<2, 2, 4, 5, false, 2, 64>, nullptr, nullptr},             // 31
    {256, 256, launch<2, 2, 4, 4, false, 2, 64>, nullptr, nullptr},             // 32
#endif
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
// The v_mfma_f32_16x16x32_bf16 instances of the plain / split-K launches of a tile configuration (same tiles, stages and waves;
// conv_gemm16_kernel).  GEGLU, row-statistics and residual-prefetch launches, and the 32-deep-stage configurations, have none.
// fn_k / fn_split_k: the channel-major K walk (KORD) of a 3x3 convolution without upsampling, where the instance exists
// fn_np: the instance without the persistent tile loop, for launches that would run one tile per block anyway (K > 640): the
// next-tile state of the persistent instance costs the 256 x 320 tile its fragment prefetch (256 registers: hipcc then reads
// each cout fragment of the second MFMA k-step right before its four MFMAs)
struct Cfg16 { int cfg; LaunchFn fn, fn_split, fn_k = nullptr, fn_split_k = nullptr, fn_np = nullptr; };
const Cfg16 kCfg16[] = {
    {0, launch16<2, 2, 2, 2>, nullptr},
    {3, launch16<2, 2, 2, 5>, launch16<2, 2, 2, 5, 2, true>},
    {10, launch16<4, 2, 2, 4>, nullptr, launch16<4, 2, 2, 4, 2, false, false, true>},
    {13, launch16<2, 2, 2, 2, 3>, nullptr},
    {15, launch16<4, 2, 2, 2, 3>, launch16<4, 2, 2, 2, 3, true>, launch16<4, 2, 2, 2, 3, false, false, true>, launch16<4, 2, 2, 2, 3, true, false, true>},
    {19, launch16<4, 2, 1, 2, 3>, launch16<4, 2, 1, 2, 3, true>},
    {23, launch16<4, 2, 2, 5, 2, false, true>, launch16<4, 2, 2, 5, 2, true>, launch16<4, 2, 2, 5, 2, false, false, true>, launch16<4, 2, 2, 5, 2, true, false, true>,
     launch16<4, 2, 2, 5, 2, false, false>},
    {29, launch16<4, 2, 1, 5>, nullptr},
};
int g_korder = -1;  // tuning (ccnet_set_option key 10): K walk of R > 1 convolutions, 0 = (tap, channel chunk), 1 = (channel chunk, tap), -1 = per layer
int g_m16 = 1;  // tuning (ccnet_set_option key 9): 0 = every launch on the 32x32x16 kernel

// 256 pixels x 32 couts, 4 waves, two blocks per CU: the output convolutions (Cout <= 8); raw-accumulator epilogue
const LaunchFn kEdgeOutFn = launch<4, 1, 2, 1, false, 3, 64, true>;

// Dispatch = a few rules for the big layers (tuned in situ at CFG batch 32) + a cost model for everything else.
//
// The model prices a (tile configuration, split-K factor S) pair as
//     rounds * (a + b * KT / S)  [+ r0 + r1 * M * Cout * S for the reduce pass when S > 1],   rounds = ceil(tiles * S / resident blocks)
// with a (start-up + epilogue of one tile) and b (one 64-deep K step) fitted per configuration by tools/fit_dispatch.py to
// tools/sweep_small.sh timings of every UNet layer shape at 1, 2, 4, 8 and 16 images (cold weights; rms error 9 %; its pick is
// within 1 % of the best measured pair summed over the layers at each size).  Rounds are what the old tile-count thresholds
// approximated: a tile count just over a multiple of 256 costs a whole extra round.
struct CostRow { int cfg; float a_us, b_us; int per_cu; };
const CostRow kCost[] = {{19, 4.94f, 0.5370f, 1}, {15, 7.85f, 0.7451f, 1}, {0, 6.48f, 0.9108f, 2},
                         {3, 8.93f, 1.3780f, 1},  {20, 11.80f, 1.4901f, 1}, {23, 13.00f, 1.7703f, 1}};
constexpr float kReduceUs = 3.14f, kReduceUsPerElem = 1.165e-6f;
constexpr int kMaxSplit = 32, kMinSliceSteps = 13;  // every K slice keeps a real K loop

// Returns the tile configuration and sets *S_out (1 = no split-K).  `can_split`: the caller can run a reduce pass.
int plan_conv(int M, int Cout, int KT, int batch, bool geglu, bool can_split, int* S_out, bool res_only = false) {
    *S_out = 1;
    const bool forced = g_force_cfg >= 0 && g_force_cfg < kNumCfgs && (geglu ? kCfgs[g_force_cfg].fn_geglu : kCfgs[g_force_cfg].fn);
    auto tiles = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((Cout + bn - 1) / bn) * batch; };
    if (geglu) {
        if (forced) return g_force_cfg;
        // 256x320 (2 x 64-deep stages) where it fills whole rounds: ff.linear1 of every G1 level (2560 / 5120 / 10240 couts)
        if ((Cout % 320) == 0 && tiles(256, 320) >= 200) return 23;
        // 256x256 where it fills whole rounds or most of one; between them (257..399 tiles) the second round runs half empty
        const long long t = (Cout % 256) == 0 ? tiles(256, 256) : 0;
        if (t >= 400 || (t >= 128 && t <= 256)) return 20;
        const bool n128 = (Cout % 128) == 0 || Cout >= 512;
        if (n128 && tiles(128, 128) >= 160) return 0;
        return tiles(128, 64) >= 384 ? 1 : 2;
    }
    can_split = can_split && batch == 1;
    int s_lo = 1, s_hi = can_split ? kMaxSplit : 1;
    if (g_force_splitk > 0 && can_split) {
        s_lo = g_force_splitk < kMaxSplit ? g_force_splitk : kMaxSplit;
        while (s_lo > 1 && KT / s_lo < kMinSliceSteps) --s_lo;
        s_hi = s_lo;
    }
    if (!forced && s_lo == 1) {
        // Layers with whole rounds of the largest tiles (measured in situ: persistent 256x320 / 256x256 tiles, 45-50 % DMA duty,
        // beat every smaller tile there; the L2 -> LDS path delivers ~64 B/clk/CU, so (BM+BN)*128 B / 64 per K step must stay
        // well under the MFMA time BM*BN/32 clk)
        // 128x320 tiles (8 waves of 32 x 160; measured with tools/bench_conv.py --cold, profiles/r02_tile_experiments.log):
        //  * K <= 320 with a residual and nothing else in the epilogue (attention o-proj, proj_out at the 64x64 level): the
        //    layer is HBM-bound and its epilogue, not its K loop, sets the time -- persistent with the residual tile
        //    prefetched at the start of the K loop (RESPRE): 391 vs 341 TFLOP/s;
        //  * one round of them where 256x320 tiles would fill half the chip (M = 8192, Cout = K = 1280: 256 vs 128 tiles):
        //    760-830 vs 646-750 TFLOP/s; K beyond 5120 goes back to the big tile, whose K loop is the more efficient one
        if ((Cout % 320) == 0 && batch == 1) {
            const long long t128 = tiles(128, 320);
            if (res_only && KT <= 5 && t128 >= 512) return 30;
            // (K up to 5120: ff.linear2 of the 16x16 level 995 vs 936 TFLOP/s, with residual 964 vs 915; at K = 11520 the big tile wins 1144 vs 962)
            // (round 3, 16x16x32 instances, tools/sweep_plan.sh: at K = 5120 two K-slices of the big tile are 6-8 % faster, 982 / 947
            // (+ res) against 924 / 900 TFLOP/s, but write 84 MB of fp32 partials per launch where the output is 21 MB: not taken,
            // 0.25 % of the pipeline against 3 % more bytes across the L2 boundary over all launches)
            if (t128 > 128 && t128 <= 256 && tiles(256, 320) < 200 && KT <= 80 && KT >= 10) return 29;
        }
        // qkv of the 16x16 level (8192 x 3840 x 1280): 480 tiles of 256 x 256 fill two rounds better than 384 of 256 x 320 fill
        // one and a half: 1043-1052 against 918-924 TFLOP/s (same sweep; the bytes across the L2 boundary are the same either
        // way, ~250-300 MB per launch: the 9.8 MB of weights thrash a 4 MB L2 once per pixel panel under both tilings)
        if ((Cout % 256) == 0 && (Cout % 320) == 0 && batch == 1 && KT > 10 && KT <= 20 && tiles(256, 256) >= 400 &&
            tiles(256, 256) <= 512 && tiles(256, 320) < 512) return 10;
        if ((Cout % 320) == 0 && tiles(256, 320) >= 200) return 23;
        if ((Cout % 256) == 0 && (Cout % 320) != 0 && tiles(256, 256) >= 150) return Cout == 256 ? 10 : 20;
        if ((Cout % 128) == 0 && (Cout % 320) != 0 && tiles(256, 128) >= 200) return 15;
        if (Cout < 128) return tiles(128, 64) >= 384 ? 1 : 2;  // narrow outputs: not part of the fit
    }
    float best = 3.0e38f;
    int best_cfg = -1;
    for (const CostRow& r : kCost) {
        if (forced && r.cfg != g_force_cfg) continue;
        const TileCfg& c = kCfgs[r.cfg];
        const long long t = tiles(c.bm, c.bn);
        for (int S = s_lo; S <= s_hi; ++S) {
            if (S > 1 && (!c.fn_split || KT / S < kMinSliceSteps)) break;
            const long long slots = (long long)kNumCUs * r.per_cu;
            const float rounds = (float)((t * S + slots - 1) / slots);
            float us = rounds * (r.a_us + r.b_us * (float)KT / (float)S);
            if (S > 1) us += kReduceUs + kReduceUsPerElem * (float)M * (float)Cout * (float)S;
            if (us < best) { best = us; best_cfg = r.cfg; *S_out = S; }
        }
    }
    if (best_cfg < 0) { best_cfg = g_force_cfg; *S_out = 1; }  // a forced configuration the model has no row for
    return best_cfg;
}

}  // namespace

extern int g_gn_onepass;  // norm.hip
extern int g_attn_mode;   // attention.hip

extern "C" int ccnet_set_option(int key, int value) {
    if (key == 0) { g_force_cfg = value; return CCNET_OK; }  // force conv tile configuration (-1 = heuristic)
    if (key == 1) { g_dbg_zero = value; return CCNET_OK; }
    if (key == 2) { g_order = value; return CCNET_OK; }
    if (key == 3) { g_force_splitk = value; return CCNET_OK; }
    if (key == 4) { g_persist = value; return CCNET_OK; }
    if (key == 6) { g_gn_onepass = value; return CCNET_OK; }
    if (key == 7) { g_attn_mode = value; return CCNET_OK; }
    if (key == 8) { g_pw320 = value; return CCNET_OK; }
    if (key == 9) { g_m16 = value; return CCNET_OK; }
    if (key == 10) { g_korder = value; return CCNET_OK; }
    if (key == 11) { g_pw_ranges = value; return CCNET_OK; }
    return CCNET_EINVAL;
}

#ifdef CCNET_STAMP
extern "C" int ccnet_debug_stamp_buffer(void* p) { g_dbg = (unsigned long long*)p; return CCNET_OK; }
#endif

extern "C" int ccnet_prof_enable(int on) {
    g_prof_on = on != 0;
    g_prof_used = 0;
    return CCNET_OK;
}

extern "C" int ccnet_prof_collect(double* ms, double* flops, int64_t* launches) {
    double tms = 0, tf = 0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        if (hipEventSynchronize(g_prof[i].e1) != hipSuccess) return CCNET_ELAUNCH;
        float t = 0;
        if (hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1) != hipSuccess) return CCNET_ELAUNCH;
        tms += t;
        tf += g_prof[i].flops;
    }
    if (ms) *ms = tms;
    if (flops) *flops = tf;
    if (launches) *launches = (int64_t)g_prof_used;
    g_prof_used = 0;
    return CCNET_OK;
}

// Records of other conv / linear-class kernels of the library (fused_ff.hip) in the same table: open returns a handle (0: profiling
// off), close records the end event.
extern "C" __attribute__((visibility("hidden"))) int ccnet_prof_open(double flops, int M, int Cout, int K, int R, int flags, void* stream) {
    if (!g_prof_on) return 0;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof_used == g_prof.size()) {
        ProfRec r{};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return 0;
        g_prof.push_back(r);
    }
    ProfRec& rec = g_prof[g_prof_used++];
    rec.flops = flops; rec.M = M; rec.Cout = Cout; rec.K = K; rec.R = R; rec.flags = flags;
    (void)hipEventRecord(rec.e0, (hipStream_t)stream);
    return (int)g_prof_used;
}
extern "C" __attribute__((visibility("hidden"))) void ccnet_prof_close(int handle, void* stream) {
    if (handle <= 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if ((size_t)handle <= g_prof.size()) (void)hipEventRecord(g_prof[handle - 1].e1, (hipStream_t)stream);
}

extern "C" int ccnet_prof_shapes(double* table, int max_rows) {
    // rows of 7 doubles {M, Cout, K, R, flags, launches, ms}; flags: bit0 stride 2, bit1 upsample, bit2 two sources,
    // bit3 GEGLU, bit4 res1, bit5 out2, bit6 batched, bit7 fused feed-forward block (fused_ff.hip).  Call before ccnet_prof_collect (which resets the records).
    int n = 0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        const ProfRec& r = g_prof[i];
        if (hipEventSynchronize(r.e1) != hipSuccess) return CCNET_ELAUNCH;
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return CCNET_ELAUNCH;
        int j = 0;
        for (; j < n; ++j) {
            double* q = table + 7 * j;
            if (q[0] == r.M && q[1] == r.Cout && q[2] == r.K && q[3] == r.R && q[4] == r.flags) break;
        }
        if (j == n) {
            if (n == max_rows) continue;
            double* q = table + 7 * n++;
            q[0] = r.M; q[1] = r.Cout; q[2] = r.K; q[3] = r.R; q[4] = r.flags; q[5] = 0; q[6] = 0;
        }
        table[7 * j + 5] += 1;
        table[7 * j + 6] += t;
    }
    return n;
}

// Output convolution (3x3, stride 1, pad 1, Cin % 64 == 0, Cout <= 8) on the implicit-GEMM kernel: NHWC bf16 in, fp32 NCHW
// out.  Called by ccnet_conv_out_small (elementwise.hip); w is [Cout][9][Cin] bf16 -- the rows a 32-cout tile reads
// beyond Cout fall outside the weight descriptor and come back as zeros.
extern "C" __attribute__((visibility("hidden"))) int ccnet_conv_out_nchw(const void* x, int ldx, int N, int H, int W, int Cin, const void* w, const float* bias,
                                   int Cout, float* out, void* stream) {
    if (!x || !w || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 64) || (ldx % 8) || Cout <= 0 || Cout > 8)
        return CCNET_EINVAL;
    ConvArgs a{};
    a.x1 = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias;
    a.N = N; a.Hin = H; a.Win = W; a.Hout = H; a.Wout = W; a.C1 = Cin; a.C2 = 0; a.Cout = Cout; a.R = 3; a.stride = 1;
    a.pad = 1; a.ups = 1; a.ldx1 = ldx; a.ldw = 9 * Cin; a.act = 0; a.scale = 1.0f;
    a.M = N * H * W; a.HWout = H * W; a.splitk = 1; a.out32 = out; a.cout_real = Cout;
    const long long xb1 = ((long long)N * H * W - 1) * ldx * 2 + (long long)Cin * 2;
    const long long wb = (long long)Cout * 9 * Cin * 2;
    if (xb1 >= 0x7fffffffLL) return CCNET_EUNSUP;
    a.x1_bytes = (unsigned)xb1; a.w_bytes = (unsigned)wb;
    return kEdgeOutFn(a, 1, (hipStream_t)stream);
}

extern "C" size_t ccnet_conv_workspace_bytes(const ccnet_conv_params* p) {
    if (!p || p->C1 <= 0 || p->Cout <= 0 || p->N <= 0) return 0;
    const int KT = p->R * p->R * ((p->C1 + p->C2) / 64);
    const int M = p->N * p->Hout * p->Wout;
    int S = 1;
    (void)plan_conv(M, p->Cout, KT, p->batch > 1 ? p->batch : 1, p->act == 2, !p->rowstat, &S, p->res1 && !p->out2);
    return S > 1 ? (size_t)S * M * p->Cout * sizeof(float) : 0;
}

extern "C" int ccnet_conv_stat_parts(const ccnet_conv_params* p) {
    if (!p || p->C1 <= 0 || p->Cout <= 0 || p->N <= 0 || p->act == 2 || p->batch > 1) return 0;
    const int KT = p->R * p->R * ((p->C1 + p->C2) / 64);
    const int M = p->N * p->Hout * p->Wout;
    int S = 1;
    // the same plan ccnet_conv_gemm_bf16 will make for these parameters (a workspace is assumed present, as ops.conv brings one)
    const int cfg_i = plan_conv(M, p->Cout, KT, 1, false, !p->rowstat, &S, p->res1 && !p->out2);
    const TileCfg& cfg = kCfgs[cfg_i];
    if (S > 1 || !cfg.fn_stat || (p->Cout % cfg.stat_span)) return 0;
    return p->Cout / cfg.stat_span;
}

namespace {
// Shared body of ccnet_conv_gemm_bf16 (workspace from the caller, `ws_limit_code` = CCNET_EWS) and of
// ccnet_conv_gemm_bf16_noalloc (workspace = the library's static device array, no profiling records: those create events).
int conv_gemm_impl(const ccnet_conv_params* p, void* ws, int64_t ws_bytes, int ws_limit_code, bool allow_prof, void* stream) {
    if (!p || !p->x1 || !p->w || !p->out) return CCNET_EINVAL;
    if (p->R != 1 && p->R != 3) return CCNET_EUNSUP;
    if (p->ups != 1 && p->ups != 2) return CCNET_EUNSUP;
    if (p->C1 <= 0 || (p->C1 % 64) || (p->C2 % 64) || p->C2 < 0 || (p->C2 > 0 && !p->x2)) return CCNET_EINVAL;
    if ((p->Cout % 8) || p->Cout <= 0) return CCNET_EINVAL;
    if ((p->ldx1 % 8) || (p->ldw % 8) || (p->ldo % 8)) return CCNET_EINVAL;
    if (p->C2 > 0 && (p->ldx2 % 8)) return CCNET_EINVAL;
    if (p->res1 && (p->ldres1 % 8)) return CCNET_EINVAL;
    if ((p->res2 != nullptr) != (p->out2 != nullptr)) return CCNET_EINVAL;
    if (p->out2 && ((p->ldo2 % 8) || (p->ldres2 % 8))) return CCNET_EINVAL;
    if (p->rowvec && (p->ld_rowvec % 4)) return CCNET_EINVAL;
    if (p->N <= 0 || p->Hin <= 0 || p->Win <= 0 || p->Hout <= 0 || p->Wout <= 0 || p->stride <= 0) return CCNET_EINVAL;
    const int batch = p->batch > 1 ? p->batch : 1;
    if (batch > 1 && (p->R != 1 || p->C2 != 0)) return CCNET_EUNSUP;
    if (p->act < 0 || p->act > 2) return CCNET_EINVAL;
    if (p->act == 2 && (p->Cout % 64)) return CCNET_EINVAL;
    if (p->act == 2 && p->rowvec) return CCNET_EUNSUP;  // GEGLU: interleaved 32-row a / gate blocks
    if (p->rowstat && !p->colsum) return CCNET_EINVAL;   // (colsum alone: statistics in the kernel, streaming instance only)
    if (p->colsum && (p->R != 1 || p->C2 != 0 || p->rowvec || batch > 1 || p->stride != 1 || p->ups != 1)) return CCNET_EUNSUP;
    if (p->rowstat && p->rowstat_parts != 0 && p->rowstat_parts != 2 && p->rowstat_parts != 4 && p->rowstat_parts != 8) return CCNET_EUNSUP;
    if (p->rowstat && p->rowstat_parts > 0 && (p->C1 % p->rowstat_parts)) return CCNET_EINVAL;
    if (p->stat_out && (p->act == 2 || batch > 1)) return CCNET_EUNSUP;
    // every gathered tap must fall inside the (upsampled) input or be a padding tap: checked in-kernel by
    // coordinates; here make sure the implied geometry is self-consistent so no row can index past x1/x2.
    const long long Hl = (long long)p->Hin * p->ups, Wl = (long long)p->Win * p->ups;
    if ((long long)(p->Hout - 1) * p->stride - p->pad >= Hl || (long long)(p->Wout - 1) * p->stride - p->pad >= Wl)
        return CCNET_EINVAL;

    ConvArgs a{};
    a.x1 = (const __bf16*)p->x1; a.x2 = (const __bf16*)p->x2; a.w = (const __bf16*)p->w;
    a.bias = p->bias; a.rowvec = p->rowvec; a.res1 = (const __bf16*)p->res1; a.res2 = (const __bf16*)p->res2;
    a.rowstat = p->rowstat; a.colsum = p->colsum; a.rowstat_parts = p->rowstat_parts; a.ln_eps = p->ln_eps;
    a.stat_out = p->stat_out;
    a.out = (__bf16*)p->out; a.out2 = (__bf16*)p->out2;
    a.N = p->N; a.Hin = p->Hin; a.Win = p->Win; a.Hout = p->Hout; a.Wout = p->Wout;
    a.C1 = p->C1; a.C2 = p->C2; a.Cout = p->Cout; a.R = p->R; a.stride = p->stride; a.pad = p->pad; a.ups = p->ups;
    a.ldx1 = p->ldx1; a.ldx2 = p->ldx2; a.ldw = p->ldw; a.ldo = p->ldo; a.ldo2 = p->ldo2;
    a.ldres1 = p->ldres1; a.ldres2 = p->ldres2; a.ld_rowvec = p->ld_rowvec; a.act = p->act; a.scale = p->scale;
    a.M = p->N * p->Hout * p->Wout; a.HWout = p->Hout * p->Wout;
    a.CT1 = p->C1 / 64; a.CT = (p->C1 + p->C2) / 64; a.KT = p->R * p->R * a.CT;
    a.sx1 = p->sx1; a.sw = p->sw; a.so = p->so;
    a.mfast = g_order == 1;  // M-fastest tile order only on request: measured no better than N-fastest on any G1 layer (2: windowed, set per launch)
    const long long xb1 = ((long long)p->N * p->Hin * p->Win - 1) * p->ldx1 * 2 + (long long)p->C1 * 2;
    const long long xb2 = p->C2 ? ((long long)p->N * p->Hin * p->Win - 1) * p->ldx2 * 2 + (long long)p->C2 * 2 : 0;
    const long long wb = ((long long)p->Cout - 1) * p->ldw * 2 + (long long)a.KT * 128;
    if (xb1 >= 0x7fffffffLL || xb2 >= 0x7fffffffLL || wb >= 0x7fffffffLL) return CCNET_EUNSUP;
    a.x1_bytes = (unsigned)xb1; a.x2_bytes = (unsigned)xb2; a.w_bytes = (unsigned)wb;
    if (g_dbg_zero & 1) { a.x1_bytes = 0; a.x2_bytes = 0; }
    if (g_dbg_zero & 2) a.w_bytes = 0;
#ifdef CCNET_STAMP
    a.dbg = g_dbg;
#endif

    hipStream_t st = (hipStream_t)stream;
    ProfRec* rec = nullptr;
    std::unique_lock<std::mutex> prof_lock(g_prof_mu, std::defer_lock);
    if (g_prof_on && allow_prof) {
        prof_lock.lock();  // the record vector is shared; with profiling off no lock is taken and calls are independent
        if (g_prof_used == g_prof.size()) {
            ProfRec r{};
            if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return CCNET_ELAUNCH;
            g_prof.push_back(r);
        }
        rec = &g_prof[g_prof_used++];
        rec->flops = 2.0 * a.M * (double)p->Cout * (double)(a.KT * 64) * batch;
        rec->M = a.M; rec->Cout = p->Cout; rec->K = a.KT * 64; rec->R = p->R;
        rec->flags = (p->stride == 2) | ((p->ups == 2) << 1) | ((p->C2 > 0) << 2) | ((p->act == 2) << 3) |
                     ((p->res1 != nullptr) << 4) | ((p->out2 != nullptr) << 5) | ((batch > 1) << 6);
        (void)hipEventRecord(rec->e0, st);
    }
    {
        bool done = false;
        const int rc = try_pw320(p, a.M, st, &done);
        if (rc != CCNET_OK || done) {
            if (rec) (void)hipEventRecord(rec->e1, st);
            return rc;
        }
        if (p->colsum && !p->rowstat) {  // only the streaming instance takes the row statistics itself
            if (rec) --g_prof_used;
            return CCNET_EUNSUP;
        }
    }
    const bool geglu = p->act == 2;
    // a caller that brings no workspace gets the best unsplit plan (slower on deep-K, few-tile problems, never an error);
    // one that brings a workspace must bring what ccnet_conv_workspace_bytes() reports
    // (a folded LayerNorm is applied in the K-loop kernel's own epilogue: such a layer never splits along K)
    const int cfg_i = plan_conv(a.M, p->Cout, a.KT, batch, geglu, ws != nullptr && !p->rowstat, &a.splitk, p->res1 && !p->out2);
    if (a.splitk > 1) {
        if (!ws || ws_bytes < (int64_t)((size_t)a.splitk * a.M * p->Cout * sizeof(float))) return ws_limit_code;
        a.ws = (float*)ws;
    }
    const TileCfg& cfg = kCfgs[cfg_i];
    // channel-major K walk for the 3x3 convolutions on 256 x 320 tiles (profiles/r03_korder.txt): unsplit (the 64x64 / 32x32 levels)
    // -3 % time at 64x64, -1 % at 32x32 and 4 - 6 x fewer bytes across the L2 boundary; split along K (the 16x16 level) together
    // with the M-fastest tile walk, which alone would double the traffic (an XCD then holds 16 pixel panels, and only under this
    // walk do their tap re-reads hit L2): 459 -> 161 MB per launch, time within 1 %.  The 8x8-level layers (256 x 128 tiles) are
    // weight-bound and lose 6 % under it (a weight row is then read in 128-byte pieces nine chunks apart): (tap, channel) walk.
    a.korder = (p->R == 3 && p->ups == 1 && (g_korder > 0 || (g_korder < 0 && cfg_i == 23))) ? 1 : 0;
    // row statistics come from the K-loop kernel's epilogue: only unsplit plans on the tiles that have the instance
    if (p->stat_out && (a.splitk > 1 || !cfg.fn_stat || (p->Cout % cfg.stat_span))) {
        if (rec) --g_prof_used;  // (nothing was launched: drop the half-made profiling record)
        return CCNET_EUNSUP;
    }
    LaunchFn fn = a.splitk > 1 ? cfg.fn_split : geglu ? cfg.fn_geglu : p->stat_out ? cfg.fn_stat : cfg.fn;
    if (g_m16 && !geglu && !p->stat_out && !p->rowstat) {
        for (const Cfg16& c : kCfg16)
            if (c.cfg == cfg_i && (a.splitk > 1 ? c.fn_split : c.fn)) {
                fn = a.splitk > 1 ? c.fn_split : c.fn;
                const LaunchFn fk = a.splitk > 1 ? c.fn_split_k : c.fn_k;
                if (a.splitk == 1 && c.fn_np && !(g_persist && a.KT * 64 <= (g_persist > 1 ? g_persist : 640))) fn = c.fn_np;
                if (a.korder && fk) fn = fk;
            }
    }
    const int rc = fn(a, batch, st);
    if (rec) (void)hipEventRecord(rec->e1, st);
    return rc;
}
}  // namespace

extern "C" int ccnet_conv_gemm_bf16(const ccnet_conv_params* p, void* stream) {
    return conv_gemm_impl(p, p ? p->workspace : nullptr, p ? p->workspace_bytes : 0, CCNET_EWS, true, stream);
}

// ---------------------------------------------------------------- allocation-free entry point
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it):
// ccnet_conv_gemm_bf16 with the split-K partial sums in a static device array instead of a caller workspace.  Every GPU
// the library runs on holds its own copy of the array (CCNET_NOALLOC_SCRATCH_BYTES, until the process ends) at its own
// address; host code never uses the array's NAME as an address (that is a host-side stand-in): the device address is
// looked up once per device with hipGetSymbolAddress, with that device current, and kept.
static __device__ __align__(256) unsigned char scratch[CCNET_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// The array's address on device `dev` (looked up on first use, with that device current, then kept).
int scratch_for_device(int dev, void** out) {
    if (dev < 0 || dev >= kMaxDevices) return CCNET_ELAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return CCNET_ELAUNCH;
        if (cur != dev && hipSetDevice(dev) != hipSuccess) return CCNET_ELAUNCH;
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (cur != dev) (void)hipSetDevice(cur);
        if (e != hipSuccess || !ptr) return CCNET_ELAUNCH;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return CCNET_OK;
}
// The device a stream belongs to (the null stream: the current device).
int device_of_stream(void* stream, int* dev) {
    if (stream) {
        hipDevice_t d;
        if (hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess) { *dev = (int)d; return CCNET_OK; }
    }
    return hipGetDevice(dev) == hipSuccess ? CCNET_OK : CCNET_ELAUNCH;
}
}  // namespace

extern "C" int ccnet_noalloc_init(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return CCNET_ELAUNCH;
    void* ws = nullptr;
    return scratch_for_device(dev, &ws);
}

extern "C" int ccnet_conv_gemm_bf16_noalloc(const ccnet_conv_params* p, void* stream) {
    // the array of the GPU that OWNS the stream (each GPU holds its own copy at its own address), and launches need that
    // device current: a stream of another device than the current one is refused rather than handed a wrong pointer
    int dev = -1, cur = -1;
    if (device_of_stream(stream, &dev) != CCNET_OK || hipGetDevice(&cur) != hipSuccess) return CCNET_ELAUNCH;
    if (dev != cur) return CCNET_EINVAL;
    void* ws = nullptr;
    const int rc = scratch_for_device(dev, &ws);
    if (rc != CCNET_OK) return rc;
    // Nothing in the array has to start at zero: every K slice writes its whole [M][Cout] slab before the reduce pass
    // reads it, so the previous call's contents are dead.  Calls that overlap on one GPU share the array and race.
    return conv_gemm_impl(p, ws, (int64_t)CCNET_NOALLOC_SCRATCH_BYTES, CCNET_ESCRATCH, false, stream);
}

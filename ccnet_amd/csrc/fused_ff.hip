// Fused feed-forward block of a BasicTransformerBlock at 320 channels (the 64x64 level of the SD-2.1 geometry):
//     out = x + linear2( GEGLU( linear1( LayerNorm(x) ) ) )                     (norm3 + MLPBlock(act="GEGLU"))
// in ONE kernel: the normalised rows, the 2 x 4C linear1 output and the 4C hidden tensor never reach HBM (unfused: LayerNorm
// launch + ff.linear1 + GEGLU launch + ff.linear2 + residual launch; 335 MB of hidden tensor written and re-read per call at
// CFG batch 32).
//
// Shape of the work on a CU: one workgroup of four waves, ONE wave per SIMD (a 512-register budget per lane).  A wave owns 32
// token rows for the whole block: its rows' raw values (the residual) and their LayerNorm (bf16, the B operand of linear1) stay
// in registers, and so do the fp32 accumulators of ALL 320 outputs of its rows (160 registers).  The hidden dimension is walked
// in blocks of 16: linear1 for the block's 16 value + 16 gate rows (20 MFMAs 32x32x16, K = 320) -> GEGLU in registers -> the
// lane's 8 bf16 hidden values ARE the B operand of the linear2 MFMAs (10 of them, one per 32 outputs): the S -> P -> P.V chaining
// of the flash-attention kernel, with W2's columns permuted at pack time to the order in which a lane holds them.  The weights
// are the A operands and are shared by the four waves: they stream through LDS by LDS-DMA in stages of two hidden blocks
// (60 KB: per block a 32 x 320 image of W1, swizzled for conflict-free fragment reads, and a [2][320][8] image of W2), packed
// on the host in exactly that order (ccnet_amd.ops.ff_pack), so a stage is one linear copy.  The workgroup is persistent: it
// walks 128-row tiles b, b + grid, ... and the weight ring keeps running across tile boundaries.
//
// Arithmetic per 128 rows: 2400 MFMAs per wave (76.8k cycles) against 2.46 MB of weights through L2 -> LDS (38k cycles at the
// 64 B/clk/CU that path delivers) and 160 KB of HBM traffic.
#include "common.h"
#include <atomic>
#include <type_traits>

namespace {

struct FfArgs {
    const __bf16* x; const __bf16* wp; const float* b1; const float* b2; const float* gamma; const float* beta;
    __bf16* out;
    int M, ldx, ldo, tiles;
    float eps;
    unsigned wp_bytes;
};

constexpr int FF_C = 320, FF_KS = FF_C / 16, FF_HID = 4 * FF_C, FF_NBLK = FF_HID / 16;
constexpr int FF_W1B = 32 * FF_C * 2, FF_W2B = FF_C * 16 * 2, FF_BLKB = FF_W1B + FF_W2B;  // bytes per hidden block: 20480 + 10240
constexpr int FF_HB = 2;                                  // hidden blocks per stage
constexpr int FF_STAGEB = FF_HB * FF_BLKB, FF_NSTAGE = FF_NBLK / FF_HB, FF_NS = 2;
constexpr int FF_PPW = FF_STAGEB / 1024 / 4;              // 1-KiB DMA pieces per wave and stage
static_assert(FF_STAGEB % 4096 == 0, "a stage divides into whole pieces per wave");
constexpr int FF_TAB = (2 * FF_HID + 3 * FF_C) * 4;       // b1 | gamma | beta | b2 (fp32)
constexpr int FF_LDS = FF_NS * FF_STAGEB + FF_TAB;
#ifndef FF_DIAG
#define FF_DIAG 0  // timing experiments (never the shipped library): bit 0 no weight DMA in the steady state, bit 1 no GEGLU arithmetic, bit 2 one W1 fragment read per block, bit 3 no wait / barrier per stage, bit 4 no LayerNorm arithmetic and no residual
#endif
#ifndef FF_PF
#define FF_PF 4   // weight fragments are read this many MFMA slots ahead of their use
#endif

// a * gelu_erf(g), scalar form of common.h's geglu_pk (same polynomial, same rounding)
__device__ __forceinline__ float geglu_f(float a, float g) {
    const float ag = fabsf(g);
    float q = fmaf(ag, 2.9924784e-05f, -0.0007398792f);
    q = fmaf(q, ag, 0.007977478f);
    q = fmaf(q, ag, -0.053238206f);
    q = fmaf(q, ag, -0.45891568f);
    q = fmaf(q, ag, -1.1511471f);
    q = q * ag;
    const float e = __builtin_amdgcn_exp2f(q);
    return a * fmaf(ag * -0.5f, e, fmaxf(g, 0.0f));
}

__global__ __launch_bounds__(256, 1) void ff320_kernel(const FfArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    float* b1s = (float*)(smem + FF_NS * FF_STAGEB);
    float* gms = b1s + 2 * FF_HID;
    float* bts = gms + FF_C;
    float* b2s = bts + FF_C;
    for (int t = tid; t < 2 * FF_HID; t += 256) b1s[t] = a.b1[t];
    for (int t = tid; t < FF_C; t += 256) { gms[t] = a.gamma[t]; bts[t] = a.beta[t]; b2s[t] = a.b2[t]; }

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, a.wp_bytes, 0x00020000);
    const int my_tiles = ((int)blockIdx.x < a.tiles) ? (a.tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    // The stream is periodic over a tile's FF_NSTAGE stages; stage s holds, for its two hidden blocks j = 2 s, 2 s + 1, the W1 image
    // of block j and the W2 image of block j - 1 (block -1 = block 79 of the PREVIOUS tile): linear2 runs one block behind linear1,
    // so that GEGLU of block j - 1 (vector instructions) issues under the MFMAs of linear1 of block j.  One stage more than the
    // tiles need is loaded: its first W2 image closes the last tile.
    const int total_loads = my_tiles * FF_NSTAGE + 1;
    {
        const int s = 0;
        char* base = smem + wave * (FF_PPW * 1024);
        const unsigned off = (unsigned)s * FF_STAGEB + (unsigned)wave * (FF_PPW * 1024) + (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < FF_PPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + i * 1024), 16,
                                                     (int)(off + i * 1024), 0, 0, 0);
    }

    // W1 fragment of k-step ks = 4 m + r: row l31, 16-byte unit (2 ks + hh) ^ swz(l31) = 8 m + ((2 r) ^ (hh ^ swz)): four lane
    // bases + a literal offset
    const int c8 = hh ^ ((l31 >> 1) & 7);
    int w1off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w1off[r] = l31 * (FF_C * 2) + (((2 * r) ^ c8) << 4);
    const int w2off = hh * (FF_C * 16) + l31 * 16;

    auto swap32 = [](unsigned& lo, unsigned& hi) {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    };

    bf16x8 xh[FF_KS];            // the wave's 32 normalised rows: lane holds channels 16 ks + 8 hh .. + 7 of row l31
    f32x16 acc2[FF_C / 32];      // all 320 outputs of those rows: register (4 q + e) of block i is cout 32 i + 8 q + 4 hh + e
    f32x16 acc1a, acc1b;         // linear1 of the even / odd hidden blocks (bias included): one waits for its GEGLU while the other accumulates
#pragma unroll
    for (int k = 0; k < 16; ++k) acc1b[k] = 0.f;
#pragma unroll
    for (int i = 0; i < FF_C / 32; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc2[i][k] = 0.f;

    // One hidden block: linear1 of block j from the W1 image at w1b (20 MFMAs, one accumulation chain) with GEGLU of block j - 1
    // and the next stage's DMA issues between them, then linear2 of block j - 1 from the W2 image at w2b (10 MFMAs).
    // The issue order inside a block is written out slot by slot -- one MFMA per slot, followed by what should issue while it
    // executes (an MFMA holds the SIMD's issue port for 8 of its 32 cycles): the fragment read that is FF_PF slots ahead, one DMA
    // piece, a slice of the GEGLU chain -- with a scheduling barrier between slots, because with one wave per SIMD nothing else
    // hides a stall and hipcc's own order clusters the MFMAs and leaves the vector work serial behind them
    // (__builtin_amdgcn_sched_group_barrier pipelines were not honoured in this loop).
    auto phase = [&](const char* w1b, const char* w2b, const float* bias, const f32x16& acc1p, f32x16& acc1n, auto dma, char* dma_base, unsigned dma_off) {
        constexpr int PF = FF_PF;
        bf16x8 fr1[FF_KS], fr2[FF_C / 32];
        f32x4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *(const f32x4*)(bias + 8 * q);
#pragma unroll
        for (int k = 0; k < PF; ++k) fr1[k] = *(const bf16x8*)(w1b + w1off[k & 3] + (k >> 2) * 128);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1n[4 * q + e] = bv[q][e];
        __builtin_amdgcn_sched_barrier(0);
        // GEGLU of the previous block in 24 parts (8 elements x 3): quads (0, 1) -> hidden 4 hh + e, quads (2, 3) -> hidden
        // 8 + 4 hh + e of the block's 16 (value rows and gate rows interleave in 8-row blocks, ops.geglu_interleave): the lane's 8
        // results, in this order, are the k = 8 hh .. 8 hh + 7 entries of linear2's B operand (W2's image is permuted to match).
        // a * gelu_erf(g) = a * (max(g, 0) - |g| / 2 * exp2(|g| Q(|g|))), common.h's polynomial, scalar fp32 on purpose: packed
        // fp32 instructions cost several times their issue slot beside MFMAs.
        float ga[8], gg[8], gq[8], gx[8], r8[8];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ga[4 * h2 + e] = acc1p[8 * h2 + e]; gg[4 * h2 + e] = acc1p[8 * h2 + 4 + e]; }
        auto gpart = [&](int t) {
            const int e = t & 7, part = t >> 3;
            const float ag = fabsf(gg[e]);
            if (FF_DIAG & 2) { r8[e] = ga[e] + gg[e]; return; }
            if (part == 0) {
                float q = fmaf(ag, 2.9924784e-05f, -0.0007398792f);
                q = fmaf(q, ag, 0.007977478f);
                gq[e] = fmaf(q, ag, -0.053238206f);
            } else if (part == 1) {
                float q = fmaf(gq[e], ag, -0.45891568f);
                q = fmaf(q, ag, -1.1511471f);
                gx[e] = __builtin_amdgcn_exp2f(q * ag);
            } else {
                r8[e] = ga[e] * fmaf(ag * -0.5f, gx[e], fmaxf(gg[e], 0.0f));
            }
        };
        bf16x8 pf;
#pragma unroll
        for (int k = 0; k < FF_KS; ++k) {
            acc1n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr1[k], xh[k], acc1n, 0, 0, 0);
            if (k + PF < FF_KS) {
                const int kk = (FF_DIAG & 4) ? 0 : k + PF;
                fr1[k + PF] = *(const bf16x8*)(w1b + w1off[kk & 3] + (kk >> 2) * 128);
            } else {
                fr2[k + PF - FF_KS] = *(const bf16x8*)(w2b + w2off + (k + PF - FF_KS) * 512);
            }
            if constexpr (decltype(dma)::value && !(FF_DIAG & 1)) {
                // stage gs + 1 into the slot that the stage before this one used (every wave left it: the barrier at the top of this
                // stage); beyond the last stage the offset falls outside the descriptor: zeros land, nothing is fetched, no branch
                if (k >= 1 && k <= FF_PPW)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dma_base + (k - 1) * 1024), 16,
                                                             (int)(dma_off + (k - 1) * 1024), 0, 0, 0);
            }
#pragma unroll
            for (int t = (k * 24) / FF_KS; t < ((k + 1) * 24) / FF_KS; ++t) gpart(t);
            if (k == FF_KS - 1) pf = __builtin_bit_cast(bf16x8, pack8(r8));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < FF_C / 32; ++i) {
            acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr2[i], pf, acc2[i], 0, 0, 0);
            if (i + PF < FF_C / 32) fr2[i + PF] = *(const bf16x8*)(w2b + w2off + (i + PF) * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int gs = 0, prow = 0;
    bool prok = false;
    const __bf16* pxp = a.x;
    for (int ti = 0;; ++ti) {
        const bool live = ti < my_tiles;
        const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
        const int row = tile * 128 + wave * 32 + l31;
        const bool rok = live && row < a.M;
        const __bf16* xp = a.x + (long long)(rok ? row : a.M - 1) * a.ldx + 8 * hh;
        if (live) {
            // ---- the wave's 32 rows -> LayerNorm -> bf16 (two-pass statistics over the lane's 160 values + its lane ^ 32 partner's)
            u32x4 xr[FF_KS];
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) xr[ks] = *(const u32x4*)(xp + 16 * ks);
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) {
                float f[8];
                unpack8(xr[ks], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[e];
            }
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / FF_C);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) {
                float f[8];
                unpack8(xr[ks], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q = fmaf(d, d, q); }
            }
            q += __shfl_xor(q, 32);
            const float rstd = rsqrtf(q * (1.0f / FF_C) + a.eps);
            if (ti == 0) {  // the tables written above
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) {
                float f[8], o[8];
                unpack8(xr[ks], f);
                const f32x4 g0 = *(const f32x4*)(gms + 16 * ks + 8 * hh), g1 = *(const f32x4*)(gms + 16 * ks + 8 * hh + 4);
                const f32x4 t0 = *(const f32x4*)(bts + 16 * ks + 8 * hh), t1 = *(const f32x4*)(bts + 16 * ks + 8 * hh + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f[e] - mean) * rstd * g0[e] + t0[e];
                    o[4 + e] = (f[4 + e] - mean) * rstd * g1[e] + t1[e];
                }
                xh[ks] = (FF_DIAG & 16) ? __builtin_bit_cast(bf16x8, xr[ks]) : __builtin_bit_cast(bf16x8, pack8(o));
            }
        }
        // stage gs: wait until it has landed (this wave's pieces: vmcnt; the others': barrier; every wave has then also left stage
        // gs - 1, whose slot takes stage gs + 1), then its two hidden blocks
        auto stage_head = [&](int s, const char*& sb, char*& dma_base, unsigned& dma_off) {
            if (!(FF_DIAG & 8)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
            sb = smem + (gs % FF_NS) * FF_STAGEB;
            const int gnext = gs + 1;
            dma_base = smem + (gnext % FF_NS) * FF_STAGEB + wave * (FF_PPW * 1024);
            dma_off = (unsigned)(gnext % FF_NSTAGE) * FF_STAGEB + (unsigned)wave * (FF_PPW * 1024) + (unsigned)lane * 16u;
            dma_off |= gnext < total_loads ? 0u : 0x80000000u;
        };
        {   // ---- stage 0: between its two blocks the PREVIOUS tile ends
            const char* sb; char* dma_base; unsigned dma_off;
            stage_head(0, sb, dma_base, dma_off);
            __builtin_amdgcn_sched_barrier(0);
            phase(sb, sb + FF_W1B, b1s + 4 * hh, acc1b, acc1a, std::true_type{}, dma_base, dma_off);
            __builtin_amdgcn_sched_barrier(0);
            // linear2 of the previous tile's last hidden block has just been added: that tile's outputs leave.  After the
            // permlane32 swap a lane owns couts 32 i + 16 p + 8 hh .. + 7 of its row; the residual is the raw 16-byte unit
            // 2 i + p of the row (re-read: L2 / Infinity Cache), one 16-byte store each.
            if (ti > 0) {
                __bf16* orow = a.out + (long long)prow * a.ldo + 8 * hh;
                u32x4 rr[FF_C / 32][2];  // (requested in two bursts of five blocks: two L2 round trips instead of ten)
#pragma unroll
                for (int i = 0; i < FF_C / 32; ++i) {
                    if (i % 5 == 0) {
#pragma unroll
                        for (int i2 = i; i2 < i + 5; ++i2)
#pragma unroll
                            for (int p = 0; p < 2; ++p)
                                rr[i2][p] = (FF_DIAG & 16) ? u32x4{0u, 0u, 0u, 0u} : *(const u32x4*)(pxp + 32 * i2 + 16 * p);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        float w8[8], r8[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned lo = __float_as_uint(acc2[i][8 * p + e]), hi = __float_as_uint(acc2[i][8 * p + 4 + e]);
                            swap32(lo, hi);
                            w8[e] = __uint_as_float(lo); w8[4 + e] = __uint_as_float(hi);
                        }
                        unpack8(rr[i][p], r8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w8[e] += r8[e];
                        if (prok) *(u32x4*)(orow + 32 * i + 16 * p) = pack8(w8);
                    }
                }
            }
            if (!live) break;
            // this tile's output accumulators start at linear2's bias
#pragma unroll
            for (int i = 0; i < FF_C / 32; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t = *(const f32x4*)(b2s + 32 * i + 8 * q + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[i][4 * q + e] = t[e];
                }
            __builtin_amdgcn_sched_barrier(0);
            phase(sb + FF_BLKB, sb + FF_BLKB + FF_W1B, b1s + 32 + 4 * hh, acc1a, acc1b, std::false_type{}, nullptr, 0u);
            __builtin_amdgcn_sched_barrier(0);
            ++gs;
        }
#pragma unroll 1
        for (int s = 1; s < FF_NSTAGE; ++s, ++gs) {
            const char* sb; char* dma_base; unsigned dma_off;
            stage_head(s, sb, dma_base, dma_off);
            __builtin_amdgcn_sched_barrier(0);
            phase(sb, sb + FF_W1B, b1s + (2 * s) * 32 + 4 * hh, acc1b, acc1a, std::true_type{}, dma_base, dma_off);
            __builtin_amdgcn_sched_barrier(0);
            phase(sb + FF_BLKB, sb + FF_BLKB + FF_W1B, b1s + (2 * s + 1) * 32 + 4 * hh, acc1a, acc1b, std::false_type{}, nullptr, 0u);
            __builtin_amdgcn_sched_barrier(0);
        }
        prow = row; prok = rok; pxp = xp;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zero-fill DMAs of the closing stage: no LDS write outlives the workgroup)
}

}  // namespace

extern "C" int ccnet_prof_open(double flops, int M, int Cout, int K, int R, int flags, void* stream);  // gemm_conv.hip
extern "C" void ccnet_prof_close(int handle, void* stream);

extern "C" int ccnet_ff_geglu_fused_bf16(const ccnet_ff_params* p, void* stream) {
    if (!p || !p->x || !p->w_packed || !p->b1 || !p->b2 || !p->gamma || !p->beta || !p->out) return CCNET_EINVAL;
    if (p->M <= 0 || (p->ldx % 8) || (p->ldo % 8) || p->ldx < p->C || p->ldo < p->C) return CCNET_EINVAL;
    if (p->C != FF_C || p->hidden != FF_HID) return CCNET_EUNSUP;
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)ff320_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess)
            return CCNET_ELAUNCH;
        attr_set.store(true, std::memory_order_release);
    }
    FfArgs a{};
    a.x = (const __bf16*)p->x; a.wp = (const __bf16*)p->w_packed; a.b1 = p->b1; a.b2 = p->b2; a.gamma = p->gamma; a.beta = p->beta;
    a.out = (__bf16*)p->out; a.M = p->M; a.ldx = p->ldx; a.ldo = p->ldo; a.eps = p->eps;
    a.tiles = (p->M + 127) / 128;
    a.wp_bytes = (unsigned)(FF_NBLK * FF_BLKB);
    const int grid = a.tiles < 256 ? a.tiles : 256;
    // benchmark hook (ccnet_prof_enable): recorded as an M x C x 12 C product (linear1 C -> 8C + linear2 4C -> C), flag bit 7
    const int rec = ccnet_prof_open(2.0 * p->M * (double)FF_C * (12.0 * FF_C), p->M, FF_C, 12 * FF_C, 1, 128, stream);
    hipLaunchKernelGGL(ff320_kernel, dim3(grid), dim3(256), FF_LDS, (hipStream_t)stream, a);
    const bool ok = hipGetLastError() == hipSuccess;
    ccnet_prof_close(rec, stream);
    if (!ok) return CCNET_ELAUNCH;
    return CCNET_OK;
}

extern "C" int64_t ccnet_ff_packed_bytes(int C, int hidden) {
    if (C != FF_C || hidden != FF_HID) return 0;
    return (int64_t)FF_NBLK * FF_BLKB;
}

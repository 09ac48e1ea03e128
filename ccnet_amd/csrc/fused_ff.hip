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
    // stage gs of this workgroup's sequence (tile after tile, FF_NSTAGE stages each) -> ring slot gs % FF_NS
    auto issue_stage = [&](int gs) {
        const int s = gs % FF_NSTAGE;
        char* base = smem + (gs % FF_NS) * FF_STAGEB + wave * (FF_PPW * 1024);
        const unsigned off = (unsigned)s * FF_STAGEB + (unsigned)wave * (FF_PPW * 1024) + (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < FF_PPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + i * 1024), 16,
                                                     (int)(off + i * 1024), 0, 0, 0);
    };

    const int my_tiles = ((int)blockIdx.x < a.tiles) ? (a.tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    const int total_stages = my_tiles * FF_NSTAGE;
    issue_stage(0);

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

    int gs = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
        const int row = tile * 128 + wave * 32 + l31;
        const bool rok = row < a.M;
        // ---- the wave's 32 rows: raw (residual) and normalised (B operand of linear1); lane holds channels 16 ks + 8 hh .. + 7
        bf16x8 xh[FF_KS];
        const __bf16* xp = a.x + (long long)(rok ? row : a.M - 1) * a.ldx + 8 * hh;
        {
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
                xh[ks] = __builtin_bit_cast(bf16x8, pack8(o));
            }
        }
        // ---- output accumulators start at linear2's bias: register (4 q + e) of block i is cout 32 i + 8 q + 4 hh + e
        f32x16 acc2[FF_C / 32];
#pragma unroll
        for (int i = 0; i < FF_C / 32; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *(const f32x4*)(b2s + 32 * i + 8 * q + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[i][4 * q + e] = t[e];
            }

        for (int s = 0; s < FF_NSTAGE; ++s, ++gs) {
            // stage gs has landed (this wave's pieces: vmcnt; the others': barrier) and every wave has left stage gs - 1,
            // whose slot takes stage gs + 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (gs + 1 < total_stages) issue_stage(gs + 1);
            const char* sb = smem + (gs % FF_NS) * FF_STAGEB;
            f32x16 acc1[FF_HB];
#pragma unroll
            for (int hb = 0; hb < FF_HB; ++hb) {
                const char* w1b = sb + hb * FF_BLKB;
                const float* bb = b1s + (s * FF_HB + hb) * 32 + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t = *(const f32x4*)(bb + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc1[hb][4 * q + e] = t[e];
                }
#pragma unroll
                for (int ks = 0; ks < FF_KS; ++ks) {
                    const bf16x8 af = *(const bf16x8*)(w1b + w1off[ks & 3] + (ks >> 2) * 128);
                    acc1[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xh[ks], acc1[hb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int hb = 0; hb < FF_HB; ++hb) {
                // GEGLU: quads (0, 1) -> hidden 4 hh + e, quads (2, 3) -> hidden 8 + 4 hh + e of the block's 16 (value rows and
                // gate rows interleave in 8-row blocks, ops.geglu_interleave): the lane's 8 results, in this order, are the
                // k = 8 hh .. 8 hh + 7 entries of linear2's B operand (W2's image is permuted to match)
                float r8[8];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x16& v = acc1[hb];
                        const f32x2 r = geglu_pk(f32x2{v[8 * h2 + e], v[8 * h2 + e + 1]}, f32x2{v[8 * h2 + 4 + e], v[8 * h2 + 5 + e]});
                        r8[4 * h2 + e] = r[0];
                        r8[4 * h2 + e + 1] = r[1];
                    }
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pack8(r8));
                const char* w2b = sb + hb * FF_BLKB + FF_W1B + w2off;
#pragma unroll
                for (int i = 0; i < FF_C / 32; ++i) {
                    const bf16x8 af = *(const bf16x8*)(w2b + i * 512);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, acc2[i], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: after the permlane32 swap a lane owns couts 32 i + 16 p + 8 hh .. + 7 of its row: the channels of the
        // raw 16-byte unit 2 i + p of the row (the residual: re-read, L2 / Infinity Cache), one 16-byte store each
        __bf16* orow = a.out + (long long)row * a.ldo + 8 * hh;
#pragma unroll
        for (int i = 0; i < FF_C / 32; ++i) {
            u32x4 rr[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) rr[p] = *(const u32x4*)(xp + 32 * i + 16 * p);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float w8[8], r8[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned lo = __float_as_uint(acc2[i][8 * p + e]), hi = __float_as_uint(acc2[i][8 * p + 4 + e]);
                    swap32(lo, hi);
                    w8[e] = __uint_as_float(lo); w8[4 + e] = __uint_as_float(hi);
                }
                unpack8(rr[p], r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) w8[e] += r8[e];
                if (rok) *(u32x4*)(orow + 32 * i + 16 * p) = pack8(w8);
            }
        }
    }
}

}  // namespace

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
    hipLaunchKernelGGL(ff320_kernel, dim3(grid), dim3(256), FF_LDS, (hipStream_t)stream, a);
    CCNET_LAUNCH_CHECK();
    return CCNET_OK;
}

extern "C" int64_t ccnet_ff_packed_bytes(int C, int hidden) {
    if (C != FF_C || hidden != FF_HID) return 0;
    return (int64_t)FF_NBLK * FF_BLKB;
}

/* ccnet_hip.h -- C ABI of libccnet_hip.so (MI355X / gfx950 only).
 *
 * The reference (RichardObi/ccnet) is pure Python: it has NO native/FFI boundary.
 * Its hot path calls third-party torch modules (monai-generative / diffusers) from
 *   src/python/training/run_controlnet_inference.py:216-256 and
 *   src/python/training/util.py:934-962, 1043-1092.
 * This header is the boundary the build defines *beneath* those module calls
 * (SURVEY.md section 8(b)): every entry point cites the reference call (or the upstream
 * op sequence reached from it) that it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless a comment says "host";
 *   - activations are NHWC bf16 ("pixel rows" x channels), row stride given in elements;
 *   - every function enqueues on `stream` (a hipStream_t passed as void*), never
 *     synchronises, never allocates, and returns 0 or a negative CCNET_E* code
 *     (text via ccnet_strerror); nothing throws across the boundary;
 *   - the caller owns all memory, including workspaces (size queries provided);
 *   - compute entry points are re-entrant and may be called from several host threads on distinct streams; the
 *     exceptions are the benchmark hooks at the end of this file (ccnet_set_option, ccnet_prof_*: process-wide state,
 *     while profiling is enabled ccnet_conv_gemm_bf16 creates HIP events and serialises on a lock) and
 *     ccnet_conv_gemm_bf16_noalloc (one scratch array per GPU: see its comment).
 */
#ifndef CCNET_HIP_H
#define CCNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCNET_OK 0
#define CCNET_EINVAL (-1)   /* bad shape / alignment / null pointer          */
#define CCNET_EUNSUP (-2)   /* configuration the kernels do not cover        */
#define CCNET_ELAUNCH (-3)  /* hipLaunch failed (see hipGetLastError)        */
#define CCNET_EWS (-4)      /* workspace too small                           */
#define CCNET_ESCRATCH (-5) /* ccnet_conv_gemm_bf16_noalloc: the problem's split-K partial sums exceed the static scratch */

const char* ccnet_strerror(int code);
int ccnet_version(void);          /* ABI version, bumped on any struct change */
const char* ccnet_arch(void);     /* "gfx950" */

/* ------------------------------------------------------------------ K1/K2/K8/K10
 * Implicit-GEMM convolution / linear layer on bf16 MFMA (v_mfma_f32_32x32x16_bf16).
 *   out[m][n] = scale * act( sum_k im2col(x)[m][k] * w[n][k] + bias[n] + rowvec[img(m)][n] ) + res1[m][n]
 *   out2[m][n] = (same value before bf16 rounding) + res2[m][n]           (optional)
 * Replaces: monai Convolution(conv_only)=nn.Conv2d 3x3/1x1 in ResnetBlock / Downsample /
 * Upsample / SpatialTransformer.proj_in,proj_out; nn.Linear in CrossAttention.to_q/k/v/to_out and
 * MLPBlock.linear1/linear2; ControlNet zero-convs with `* conditioning_scale`; the UNet's
 * `res + additional_residual` (res2/out2) and `torch.cat([h, skip], 1)` (x1/x2) -- all reached from
 * util.py:944 (ControlNet.forward) and util.py:1062 (DiffusionModelUNet.forward).
 * F.interpolate(scale_factor=2,"nearest") before a conv is folded into the gather (ups=2).
 */
typedef struct ccnet_conv_params {
    const void* x1;       /* bf16 [N*Hin*Win][ldx1]                                             */
    const void* x2;       /* bf16 second source for channel concat, or NULL                     */
    const void* w;        /* bf16 [Cout][ldw], K order = (kr, ks, c over x1 then x2)            */
    const float* bias;    /* [Cout] or NULL                                                     */
    const float* rowvec;  /* [N][ld_rowvec] per-image additive vector (time-embedding) or NULL  */
    const void* res1;     /* bf16 [M][ldres1] or NULL                                           */
    const void* res2;     /* bf16 [M][ldres2] or NULL (requires out2)                           */
    void* out;            /* bf16 [M][ldo]                                                      */
    void* out2;           /* bf16 [M][ldo2] or NULL                                             */
    int32_t N, Hin, Win;  /* input images and their (pre-upsample) size                         */
    int32_t Hout, Wout;   /* output size; M = N*Hout*Wout                                       */
    int32_t C1, C2;       /* channels taken from x1 / x2 (multiples of 64; C2 = 0 if no x2)     */
    int32_t Cout;         /* multiple of 8                                                      */
    int32_t R;            /* kernel size, 1 or 3 (square)                                       */
    int32_t stride, pad;  /* pad_lo; the high side is implied by Hout/Wout                      */
    int32_t ups;          /* 1, or 2 = nearest-neighbour x2 upsample folded into the gather     */
    int32_t ldx1, ldx2, ldw, ldo, ldo2, ldres1, ldres2, ld_rowvec;  /* row strides in elements  */
    int32_t act;          /* 0 none, 1 SiLU, 2 GEGLU: w/bias rows interleave 8-row blocks of the value and
                             gate halves of MLPBlock.linear1; out has Cout/2 columns = a * gelu_erf(gate)      */
    float scale;
    int32_t batch;        /* >1: batched plain GEMM (R==1); z-th problem uses the strides below */
    int64_t sx1, sw, so;  /* element strides between batch problems for x1, w, out              */
    void* workspace;      /* fp32 scratch for split-K partial sums, ccnet_conv_workspace_bytes(p) bytes (may be 0) */
    int64_t workspace_bytes;
    /* LayerNorm folded into a linear layer (BasicTransformerBlock.norm1/2/3 -> to_q/k/v, ff.linear1): x1 holds the
     * UN-normalised rows, w = gamma o W (rounded to bf16), bias = bias + W beta, colsum[n] = sum_k w[n][k] (of the rounded
     * weights), rowstat[m] = {mean, rstd} of row m (ccnet_row_stats_bf16).  Then
     *   out[m][n] = act( rstd[m] * (sum_k x1[m][k] w[n][k] - mean[m] * colsum[n]) + bias[n] ) ...
     * = act(LayerNorm(x1)[m] . W[n] + bias[n]) without the normalised tensor ever existing in HBM.  Both NULL: plain GEMM.
     * Needs R = 1, one source, no rowvec, batch 1; never runs split along K.
     * colsum WITHOUT rowstat (ln_eps set): the kernel takes mean / rstd of every row itself, from the rows it stages.  Only the
     * weight-stationary streaming kernel does that (C1 = 320, Cout a multiple of 320 up to 10240, M >= 65536, act = 0, no
     * residual, no second output); any other problem returns CCNET_EUNSUP and the caller runs ccnet_layernorm_bf16. */
    const float* rowstat; /* fp32 [M][2] or NULL */
    const float* colsum;  /* fp32 [Cout] or NULL */
    /* rowstat_parts > 0: rowstat is instead fp32 [M][rowstat_parts][2] = (mean, M2) of that many equal column spans of
     * every row of x1, as the producing layer wrote them through stat_out (below); the kernel merges them (Chan) and uses
     * rstd = 1 / sqrt(M2 / C1 + ln_eps).  0: rowstat is the final {mean, rstd}. */
    int32_t rowstat_parts;
    float ln_eps;
    /* stat_out != NULL: also write (mean, M2) of every span of ccnet_conv_stat_parts(p) equal column spans of every OUTPUT
     * row (of `out`, after bias / activation / scale / res1), fp32 [M][parts][2] -- the statistics of the LayerNorm that
     * follows this layer (BasicTransformerBlock: proj_in -> norm1, attn1.to_out + x -> norm2, attn2.to_out + x -> norm3),
     * taken in the epilogue so that LayerNorm needs no pass of its own.  Only plans for which ccnet_conv_stat_parts
     * returns > 0 can do this; others return CCNET_EUNSUP. */
    float* stat_out;
} ccnet_conv_params;

/* Number of column spans per row that ccnet_conv_gemm_bf16 would write through stat_out for these parameters (the planned
 * tile's wave width divides Cout, the plan is not split along K, no GEGLU), or 0 if this plan cannot emit row statistics
 * (the caller then runs ccnet_row_stats_bf16 on the output). */
int ccnet_conv_stat_parts(const ccnet_conv_params* p /*host*/);

int ccnet_conv_gemm_bf16(const ccnet_conv_params* p /*host*/, void* stream);
/* Deep-K layers with few output tiles are split along K (partial sums in `workspace`, summed in slice order by a second
 * launch): bytes the caller must provide for this problem, 0 if it runs unsplit.  With workspace == NULL the problem runs
 * unsplit (correct, slower where a split was planned); a non-NULL workspace smaller than this returns CCNET_EWS. */
size_t ccnet_conv_workspace_bytes(const ccnet_conv_params* p /*host*/);

/* Allocation-free variant of ccnet_conv_gemm_bf16.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * arguments and results (bit-identical); p->workspace / p->workspace_bytes are ignored: the split-K partial sums live in
 * a static device array of CCNET_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.
 * A problem whose plan needs more than that returns CCNET_ESCRATCH (ccnet_conv_workspace_bytes(p) tells beforehand).
 * Calls no hipMalloc / hipFree / allocating routine; its first call on a device looks the array's address up once.
 * All calls on one GPU share the array: calls that overlap in time on one GPU (two streams) race -- serialise them.
 * `stream` must belong to the CURRENT device (its GPU's copy of the array is used; a stream of another device returns
 * CCNET_EINVAL).  ccnet_noalloc_init() looks the array's address up for the current device ahead of time -- call it once per
 * device OUTSIDE a stream capture (the first ccnet_conv_gemm_bf16_noalloc call on a device otherwise does the lookup itself,
 * which must not fall inside a capture). */
#define CCNET_NOALLOC_SCRATCH_BYTES (96u << 20) /* 96 MiB: SD-2.1 geometry up to 16 images per batch (CFG batch 32) */
int ccnet_noalloc_init(void);
int ccnet_conv_gemm_bf16_noalloc(const ccnet_conv_params* p /*host*/, void* stream);

/* Feed-forward block of a BasicTransformerBlock in one kernel:
 *   out[m] = x[m] + linear2( a * gelu_erf(g) ),  (a | g) = linear1( LayerNorm(x[m]; gamma, beta, eps) )
 * Replaces `x = ff(norm3(x)) + x` of monai-generative's BasicTransformerBlock = nn.LayerNorm -> MLPBlock(act="GEGLU")
 * (linear1: C -> 8C, chunk, a * gelu(g), linear2: 4C -> C) + the residual add, reached from util.py:944 (ControlNet.forward)
 * and util.py:1062 (DiffusionModelUNet.forward); three ccnet_* launches otherwise (ccnet_layernorm_bf16, ccnet_conv_gemm_bf16
 * with act = 2, ccnet_conv_gemm_bf16 with res1).  The normalised rows and the hidden tensor never reach HBM.
 * Covers C = 320, hidden = 1280 (the 64x64 level of the SD-2.1 geometry); other shapes return CCNET_EUNSUP and the caller
 * runs the three launches.  w_packed: both weight matrices in the kernel's staging order, ccnet_ff_packed_bytes(C, hidden)
 * bytes (0: shape not covered), per block j of 16 hidden units:
 *   [32 rows][320] bf16: rows (a_{16j..16j+7}, g_{16j..16j+7}, a_{16j+8..16j+15}, g_{16j+8..16j+15}) of linear1.weight, the
 *                        16-byte unit u of row r stored at position u ^ ((r >> 1) & 7);
 *   [2][320][8] bf16:    entry (hh, n, t) = linear2.weight[n][16 j' + (t < 4 ? 4 hh + t : 4 + 4 hh + t)], j' = (j - 1) mod (hidden / 16):
 *                        the kernel runs linear2 one hidden block behind linear1;
 * b1: linear1.bias in the same row order as the first image ([2 * hidden]); b2: linear2.bias [C]. */
typedef struct ccnet_ff_params {
    const void* x;        /* bf16 [M][ldx]: the block's input rows (also the residual)            */
    const void* w_packed; /* bf16, see above                                                      */
    const float* b1;      /* [2 * hidden], interleaved as the rows of the linear1 image           */
    const float* b2;      /* [C]                                                                  */
    const float* gamma;   /* LayerNorm weight [C]                                                 */
    const float* beta;    /* LayerNorm bias [C]                                                   */
    void* out;            /* bf16 [M][ldo]                                                        */
    int32_t M, C, hidden; /* rows, channels, hidden width (4 C)                                   */
    int32_t ldx, ldo;     /* row strides in elements (multiples of 8)                             */
    float eps;            /* LayerNorm epsilon                                                    */
} ccnet_ff_params;
int ccnet_ff_geglu_fused_bf16(const ccnet_ff_params* p /*host*/, void* stream);
int64_t ccnet_ff_packed_bytes(int C, int hidden);

/* Fused cross-attention branch of a transformer block: out = x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(d)) V) in one launch
 * (monai-generative BasicTransformerBlock.forward: `x = self.attn2(self.norm2(x), context=context) + x`, CrossAttention with
 * to_q / to_k / to_v / to_out; reached from util.py:944,1062).  K and V are the prompt's projections, which do not depend on the
 * timestep: the caller computes them once per prompt batch (to_k | to_v as one linear layer) and lays them out per image as
 *   [80 keys][C] bf16, keys >= `keys` zero, the 16-byte unit at position p of row r being unit p ^ ((r >> 1) & 7) of the key's K row,
 *   [5][2][C][8] bf16: entry (s, hh, c, t) = V[16 s + (t < 4 ? 4 hh + t : 4 + 4 hh + t)][c]
 * (ccnet_xattn_kv_image_bytes per image; ccnet_amd.ops.xattn_kv_image); rows [i * rows_per_image, (i + 1) * rows_per_image) of x
 * attend to image i.  Served for C = 320, 5 heads of 64 and <= 80 keys (the 64x64 level of the SD-2.1 geometry; other shapes:
 * CCNET_EUNSUP, run the layers one by one).
 * w_packed (ccnet_amd.ops.xattn_pack): 20 stages of 20480 bytes.  Stages 0 .. 9: rows 32 i .. 32 i + 31 of to_q.weight * log2(e) /
 * sqrt(64), rounded to bf16, with the rows of every 16-row group in the order (0..3, 8..11, 4..7, 12..15) and the 16-byte units of
 * row r swizzled as above.  Stages 10 + 2 h + j (head h, j = 0, 1): k-steps s = 2 j, 2 j + 1 of the head's to_out columns as
 * [2][C][8] each: entry (hh, n, t) = to_out.weight[n][64 h + 16 s + (t < 4 ? 4 hh + t : 4 + 4 hh + t)].  bq: to_q bias (if the
 * layer has one) scaled and reordered like the rows; bo: to_out.bias. */
typedef struct ccnet_xattn_params {
    const void* x;         /* bf16 [M][ldx]: the branch's input rows (also the residual)           */
    const void* w_packed;  /* bf16, see above                                                      */
    const float* bq;       /* [C] or NULL                                                          */
    const float* bo;       /* [C] or NULL                                                          */
    const float* gamma;    /* LayerNorm weight [C]                                                 */
    const float* beta;     /* LayerNorm bias [C]                                                   */
    const void* kv_image;  /* bf16 [images][ccnet_xattn_kv_image_bytes / 2]                        */
    void* out;             /* bf16 [M][ldo]                                                        */
    int32_t M, C, heads, keys;          /* rows, channels, heads, keys per image (<= 80)          */
    int32_t rows_per_image, images;     /* tokens per image (a multiple of 128), images in kv_image */
    int32_t ldx, ldo;      /* row strides in elements (multiples of 8)                             */
    float eps;             /* LayerNorm epsilon                                                    */
} ccnet_xattn_params;
int ccnet_xattn_fused_bf16(const ccnet_xattn_params* p /*host*/, void* stream);
int64_t ccnet_xattn_packed_bytes(int C, int heads);
int64_t ccnet_xattn_kv_image_bytes(int C, int heads);

/* Direct 3x3 convolution for tiny channel counts at the network edges.
 *   conv_in_small : x fp32 NCHW [Nsrc][Cin<=8][H][W] (image n reads x[n % Nsrc]) -> NHWC bf16 [N][Hout][Wout][ldo]
 *                   out = act(conv(x) + bias) + res[n % Nres]   (res NHWC bf16, optional)
 *   conv_out_small: x NHWC bf16 -> fp32 NCHW [N][Cout<=8][H][W], stride 1, pad 1
 * Replaces: conv_in (4->C0), controlnet_cond_embedding.conv_in (1->e0) and `h = conv_in(x) + cond_emb`,
 * the UNet `out.2` conv (C0->4) and the AEKL decoder's last conv (util.py:944,1062; run_controlnet_inference.py:256).
 */
int ccnet_conv_in_small(const float* x, int Nsrc, int N, int Cin, int H, int W,
                        const float* w /*[Cout][Cin][3][3] fp32*/, const float* bias, int Cout,
                        int stride, int act, const void* res, int Nres, int ldres,
                        void* out, int ldo, int Hout, int Wout, void* stream);
int ccnet_conv_out_small(const void* x, int ldx, int N, int H, int W, int Cin,
                         const void* w /*bf16 [Cout][9][Cin]*/, const float* bias, int Cout,
                         float* out, void* stream);

/* ------------------------------------------------------------------ K3
 * GroupNorm (+ optional SiLU) over NHWC bf16, fp32 statistics, optional two-source channel concat.
 * Replaces nn.GroupNorm(32,C,eps)+F.silu in ResnetBlock.norm1/norm2, `out.0/1`, SpatialTransformer.norm,
 * AEKL ResBlock norms; with x2 != NULL also the torch.cat before norm1 of every up-block ResnetBlock.
 * Two launches inside: partial (sum, sumsq) per (image, pixel-chunk, group) -> apply.
 */
size_t ccnet_groupnorm_workspace_bytes(int N, int HW, int C, int groups);
int ccnet_groupnorm_bf16(const void* x1, int ldx1, int C1, const void* x2, int ldx2, int C2,
                         int N, int HW, int groups, float eps, const float* gamma, const float* beta,
                         int silu, void* out, int ldo, void* workspace, size_t workspace_bytes, void* stream);

/* K6: nn.LayerNorm(C, eps) over rows of a [rows][ldx] bf16 matrix (BasicTransformerBlock.norm1/2/3). */
int ccnet_layernorm_bf16(const void* x, int ldx, int rows, int C, float eps, const float* gamma,
                         const float* beta, void* out, int ldo, void* stream);

/* Row statistics for the folded LayerNorm of ccnet_conv_gemm_bf16: stats[r] = {mean, 1 / sqrt(var + eps)} over the first C
 * columns of row r (biased variance, two-pass over registers, fp32), one read of the matrix and 8 bytes written per row. */
int ccnet_row_stats_bf16(const void* x, int ldx, int rows, int C, float eps, float* stats /*[rows][2]*/, void* stream);

/* K7: GEGLU, out[r][c] = a[r][c] * gelu_erf(g[r][c]) with a = x[:, :C], g = x[:, C:2C] (MLPBlock act="GEGLU"). */
int ccnet_geglu_bf16(const void* x, int ldx, int rows, int C, void* out, int ldo, void* stream);

/* ------------------------------------------------------------------ K4/K5
 * Flash-style attention, head dim 64, bf16 in/out, fp32 online softmax.
 *   O[b, i, h*64:(h+1)*64] = softmax_j( scale * Q[b,i,h] . K[b,j,h] ) V[b,j,h]
 * Q rows: b*Sq + i, K/V rows: b*Skv + j; ld* are row strides in elements, head h at column h*64.
 * Replaces CrossAttention._attention (baddbmm -> softmax -> bmm) for self (context = x) and
 * 77-token cross attention (context = prompt_embeds).
 */
int ccnet_attention_d64_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                             void* O, int ldo, int B, int H, int Sq, int Skv, float scale, void* stream);

/* The same product for a Q that already carries scale * log2(e) (the caller folded it into the query projection's weights
 * and bias, as ccnet_amd.nets does when it packs to_q: same reference call as above, CrossAttention._attention): the softmax is
 * then exp2 of the raw scores, and the long-key instance takes the row maximum off inside the matrix core. */
int ccnet_attention_d64_prescaled_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                       void* O, int ldo, int B, int H, int Sq, int Skv, void* stream);

/* Causal self-attention (key j <= query i), same layout and kernel: the CLIP text tower's masked attention
 * (transformers CLIPTextModel, called at run_controlnet_inference.py:149,205 as text_encoder(input_ids)). */
int ccnet_attention_d64_causal_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                    void* O, int ldo, int B, int H, int S, float scale, void* stream);

/* Text tower pieces (run_controlnet_inference.py:126,149): token + position embedding lookup (fp32 tables, bf16 rows out;
 * row t uses position t % seq) and the MLP activation (mode 0 = erf GELU, 1 = quick_gelu). */
int ccnet_embed_tokens_bf16(const int64_t* ids, int T, int seq, int vocab, const float* tok, const float* pos, int C,
                            void* out, int ldo, void* stream);
int ccnet_gelu_bf16(const void* x, int ldx, int rows, int C, int mode, void* out, int ldo, void* stream);

/* Generic-head-dim attention fallback (single head of 512 / 768 channels in the repo's YAML geometry,
 * configs/controlnet/controlnet_v1.yaml:11): scores by ccnet_conv_gemm_bf16 (batched, scale in the epilogue), this row
 * softmax (in place capable; columns [cols, cols_pad) are written as zeros), V^T by ccnet_transpose_pad_bf16
 * (out[z][c][r] = x[z][r][c], zero for rows <= r < rows_pad), then P.V by ccnet_conv_gemm_bf16 again.
 * Replaces CrossAttention._attention where the flash kernel's d = 64 does not apply. */
int ccnet_softmax_rows_bf16(const void* x, int ldx, int rows, int cols, int cols_pad, float scale, void* out, int ldo,
                            void* stream);
int ccnet_transpose_pad_bf16(const void* x, int ldx, int rows, int cols, int64_t sx, void* out, int ldo, int rows_pad,
                             int64_t so, int batch, void* stream);

/* ------------------------------------------------------------------ K11
 * fp32 embedding path. timestep_embedding: emb[b][0:half]=cos(t_b*freqs), [half:dim]=sin (cos first);
 * freqs = exp(-ln(10000)*arange(half)/half) is computed once on the host in fp32.
 * linear_f32: y[b][n] = act_out( sum_k act_in(x[b][k]) * w[n][k] + bias[n] ) + add[b][n]   (silu_out = 1), or
 *             y = silu( ... + bias + add )                                                  (silu_out = 2)
 * (time_embed, CCnet class_embedding MLP util.py:1128-1132, all time_emb_proj(silu(emb)) rows at once).
 */
int ccnet_timestep_embedding(const float* t /*[nt]*/, int nt, int B, int dim, const float* freqs /*[dim/2]*/,
                             float* out, void* stream);
int ccnet_linear_f32(const float* x, int ldx, int B, int K, const float* w, const float* bias, int Nout,
                     int silu_in, int silu_out, const float* add, int ldadd, float* out, int ldo, void* stream);

/* ------------------------------------------------------------------ K9
 * Classifier-free-guidance combine + scheduler step, fp32, one launch.
 *   model_out fp32 NCHW [2B or B][...]: uncond = first B images, text = last B (run_controlnet_inference.py:245).
 *   mode 0: pred=text (guidance None), 1: pred=uncond (guidance 0), 2: uncond + g*(text-uncond);
 *   mode 3: model_out holds B images only and is the prediction itself (no CFG batch: the loop of
 *           log_controlnet_samples, util.py:885-909).
 *   coef (device, 8 floats): {sqrt_acp_t, sqrt_1m_acp_t, c_x0, c_eps_or_xt, sigma, clip(0/1), pred_type(0 eps,1 sample,2 v), ddpm(0/1)}
 *   DDIM: prev = c_x0*x0 + c_eps*eps ; DDPM: prev = c_x0*x0 + c_xt*x + sigma*noise   (scheduler.step, :253)
 */
int ccnet_cfg_scheduler_step(const float* model_out, int cfg_mode, float guidance, const float* coef,
                             const float* noise /*or NULL*/, float* sample /*in/out*/, float* x0_out /*or NULL*/,
                             int B, int per_image, void* stream);

/* fp32 NCHW <-> bf16 NHWC helpers for the module-level API (ControlNet residual lists are NCHW tensors). */
int ccnet_nchw_f32_to_nhwc_bf16(const float* x, int N, int C, int HW, void* out, int ldo, void* stream);
int ccnet_nhwc_bf16_to_nchw_f32(const void* x, int ldx, int N, int C, int HW, float* out, void* stream);

/* AEKL 1x1 convs on <= 8 fp32 NCHW channels: quant_conv_mu / post_quant_conv (mode 0) and
 * z_sigma = exp(clamp(quant_conv_log_sigma(h), -30, 20) / 2) (mode 1); out = a + b*c is AutoencoderKL.sampling
 * (util.py:1490-1491). */
int ccnet_pointwise_nchw_f32(const float* x, int N, int Cin, int HW, const float* w /*[Cout][Cin]*/,
                             const float* bias, int Cout, int mode, float* out, void* stream);
int ccnet_axpy_f32(const float* a, const float* b, const float* c, long long n, float* out, void* stream);

/* Loader edge: LoadImaged -> Resized(mode="bicubic") -> ScaleIntensityRanged(0,255 -> -1,1, clip) (util.py:123-128).
 * src uint8 [N][H0][W0] (single channel), out fp32 [N][1][H][W]; transpose != 0 reads each image as its transpose (what
 * MONAI's PNG reader hands the network).  Bicubic arithmetic of F.interpolate(align_corners=False), A = -0.75. */
int ccnet_preprocess_u8_bicubic(const uint8_t* src, int N, int H0, int W0, int transpose, float* out, int H, int W,
                                void* stream);

/* interval_mapping(x, -1, 1, 0, 1) -> rint(* 255) -> uint8 of channel 0 (util.py:1162-1173, 1412-1416). x fp32 NCHW.
 * Integer output, so the arithmetic is the reference's, bit for bit: f32_scale = 0 is the synthetic slice's recipe
 * (`.astype(np.float64)` first, util.py:1402-1414: everything in float64, rint = ties to even); f32_scale = 1 is what the
 * comparison panels get (util.py:1432-1435: cond / images reach interval_mapping as float32 arrays, so (x + 1) / 2 is
 * evaluated in float32 and only its result is widened).  NaN -> 0. */
int ccnet_to_uint8(const float* x, int N, int C, int HW, int f32_scale, uint8_t* out, void* stream);

/* ------------------------------------------------------------------ profiling hooks for bench.py
 * When enabled, ccnet_conv_gemm_bf16 brackets each launch with hipEvents on `stream`;
 * ccnet_prof_collect synchronises the events and returns summed kernel ms / FLOPs / launches. */
/* Tuning switches for A/B timing (tools/, bench.py; CCNET_HIP_OPTIONS="key=value,..." sets them at load): key 0 forces a conv
 * tile configuration (-1 = planned), 2 the tile order, 3 the split-K factor, 4 persistent launches (0 off, > 1 = K limit),
 * 6 the one-pass GroupNorm paths, 7 the attention instance (0 by shape, 1 round-1 kernel, 2 long-key instance wherever it
 * applies), 8 the streaming pointwise kernel (0 off; bits: 1 on, 2 several cout groups, 4 GEGLU, 8 the 640-channel instance, 16 LayerNorm
 * statistics in the kernel: without it a colsum-only problem returns CCNET_EUNSUP), 9 the 16x16x32-MFMA instances of the tile
 * kernel (0: every launch on the 32x32x16 instances), 2 = 2 also selects the windowed tile walk, 10 the K walk of the 3x3
 * convolutions on those instances (0 (tap, channel chunk), 1 (channel chunk, tap), -1 per layer: the walk changes, the weight
 * image does not), 11 the row ranges of a streaming launch whose cout groups do not fill the XCDs in units of eight ranges
 * (0: one round of floor(CUs / groups) ranges; a multiple of 8: that many ranges in several rounds).  Results do not depend on any of them beyond the order of fp32 sums inside a tile (the streaming kernel is
 * bit-identical to the tiled one). */
int ccnet_set_option(int key, int value);
int ccnet_prof_enable(int on);
int ccnet_prof_collect(double* ms, double* flops, int64_t* launches);
/* per-shape table of the recorded launches: rows of 7 doubles {M, Cout, K, R, flags, launches, ms}; returns the row
 * count (or a negative code). Call before ccnet_prof_collect, which resets the records. */
int ccnet_prof_shapes(double* table /*host*/, int max_rows);

/* ---------------------------------------------------------------------------------------------------------------------
 * Training step of the ControlNet: backward kernels.  Reference: the autograd of
 * src/python/training/training_functions.py:888-1043 (train_epoch_controlnet: loss.backward() through monai-generative's
 * ControlNet and the frozen DiffusionModelUNet).  Input gradients of conv / linear layers are ccnet_conv_gemm_bf16 over dY
 * with transposed, tap-flipped weights; weight gradients are ccnet_conv_gemm_bf16 on [channel][pixel] operands
 * (dY^T by ccnet_transpose_pad_bf16, im2col(X)^T by ccnet_im2col_t_bf16): out[cout][tap * Cin + cin]. */

/* out[(tap * (C1 + C2) + c)][m] = x[(n, iy, ix) of output pixel m under tap][c] (the forward gather of ccnet_conv_gemm_bf16:
 * stride, pad, nearest x2 upsample, two-source concat), zeros for padding taps and for M <= m < ldo; ldo % 64 == 0. */
int ccnet_im2col_t_bf16(const void* x1, int ldx1, int C1, const void* x2, int ldx2, int C2, int N, int Hin, int Win,
                        int Hout, int Wout, int R, int stride, int pad, int ups, void* out /*bf16 [R*R*(C1+C2)][ldo]*/,
                        int64_t ldo, void* stream);
/* the same for the fp32 NCHW input of ccnet_conv_in_small (3x3, pad 1, Cin <= 8): rows 9 * Cin .. rows_pad are zeros */
int ccnet_im2col_t_nchw_f32(const float* x, int Nsrc, int N, int Cin, int H, int W, int stride, void* out, int64_t ldo,
                            int rows_pad, void* stream);
/* Column sums over row segments (bias gradient, per-image time-embedding-row gradient): out[(split * nseg + seg)][c] =
 * sum over the segment's rows that fall to `split` of y[r][c]; with x / rowstat ({mean, rstd} per row) also out2 = sum of
 * y * (x - mean) * rstd (LayerNorm gamma gradient).  The caller adds the nsplit partials. */
int ccnet_colsum_bf16(const void* y, int ldy, int nseg, int rows_per_seg, int C, const void* x, int ldx,
                      const float* rowstat, int nsplit, float* out, float* out2, void* stream);
/* Backward of ccnet_groupnorm_bf16 (same operands; statistics recomputed): dx (two sources as in the forward) and, when
 * dparam != NULL, dparam[n][c] = {dgamma, dbeta} of image n (the caller sums over n). */
int ccnet_groupnorm_bwd_bf16(const void* x1, int ldx1, int C1, const void* x2, int ldx2, int C2, const void* dy, int lddy,
                             int N, int HW, int groups, float eps, const float* gamma, const float* beta, int silu,
                             void* dx1, int lddx1, void* dx2, int lddx2, float* dparam, void* stream);
/* Backward of ccnet_layernorm_bf16 w.r.t. x; rowstat (optional) receives {mean, rstd} per row for ccnet_colsum_bf16. */
int ccnet_layernorm_bwd_bf16(const void* x, int ldx, const void* dy, int lddy, int rows, int C, float eps,
                             const float* gamma, void* dx, int lddx, float* rowstat, void* stream);
/* Flash attention (head dim 64) for the training step: the forward with the row log-sum-exp kept (lse[b][h][q], base 2, of the
 * scaled scores), and its backward -- P is recomputed from Q, K and lse tile by tile (no N x N tensor), dQ by a pass with the
 * query on the lane, dK / dV by a pass with the key on the lane, no atomics.  delta [B][H][Sq] is scratch the call fills
 * (rowsum(dO o O)).  dQ, or the pair dK / dV, may be NULL (cross-attention under a frozen projection). */
int ccnet_attention_d64_lse_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int B,
                                 int H, int Sq, int Skv, float scale, float* lse, void* stream);
int ccnet_attention_d64_bwd_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo,
                                 const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq, void* dK, int lddk,
                                 void* dV, int lddv, int B, int H, int Sq, int Skv, float scale, void* stream);
/* Row softmax backward, in place over dp: ds = scale * p * (dp - rowsum(p * dp)); columns [cols, cols_pad) become 0. */
int ccnet_softmax_bwd_rows_bf16(const void* p, void* dp, int ld, int rows, int cols, int cols_pad, float scale, void* stream);
/* Backward of ccnet_geglu_bf16: h = (a | g) [rows][2C], dy [rows][C] -> dh = (dy * gelu(g) | dy * a * gelu'(g)). */
int ccnet_geglu_bwd_bf16(const void* h, int ldh, const void* dy, int lddy, int rows, int C, void* dh, int lddh, void* stream);
/* mode 0: out = silu(x) (the SiLU the sampler fuses into conv epilogues, unfused so its input survives for the backward);
 * mode 1: out = dy * silu'(x). */
int ccnet_silu_bf16(const void* x, int ldx, const void* dy, int lddy, int rows, int C, int mode, void* out, int ldo, void* stream);
/* a = alpha * a + beta * b over bf16 rows (gradient accumulation where a tensor has two consumers). */
int ccnet_axpby_bf16(void* a, int lda, const void* b, int ldb, int rows, int C, float alpha, float beta, void* stream);
/* mode 0: adjoint of the nearest x2 upsample folded into a conv gather: out [N][H][W][C] = 2x2 sums of in [N][2H][2W][C];
 * mode 1: adjoint of a stride-2 convolution's subsampling: out [N][H][W][C] = in [N][Hs][Ws][C] at the even pixels, else 0. */
int ccnet_resample_bwd_bf16(const void* in, int ldi, void* out, int ldo, int N, int H, int W, int Hs, int Ws, int C, int mode,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif

/* grl_hip.h -- C ABI of the MI355X-native GRL hot path (libgrl_hip.so).
 *
 * The reference (ofsoundof/GRL-Image-Restoration) is pure Python: its "FFI" for this path is the
 * list of torch ops issued inside models/networks/grl.py::GRL.forward and the modules under
 * models/common/.  Each entry point below replaces one fused group of those ops
 * and is what a ctypes / cpp_extension binding on the reference side would call
 * (INTEGRATION.md shows the binding).  Conventions:
 *
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer owned by the caller;
 *   - no allocation, no host synchronisation; work is enqueued on `stream` (a hipStream_t,
 *     passed as void* so the header needs no HIP include; NULL = default stream);
 *   - return value: 0 on success, a hipError_t (> 0) from the launch, or a GRL_ERR_* (< 0);
 *   - activations are channels-last token matrices [B*H*W, Cpad] with Cpad = channels rounded
 *     up to a multiple of 32 and the pad channels held at zero.
 */
#ifndef GRL_HIP_H
#define GRL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRL_ERR_BAD_ARG (-1)
#define GRL_ERR_UNSUPPORTED (-2)
#define GRL_ABI_VERSION 22

/* element kinds of activation / weight buffers */
enum { GRL_DT_F32 = 0, GRL_DT_BF16 = 1, GRL_DT_F16 = 2 };

/* ---------------------------------------------------------------------------------------------
 * Token-wise linear layer with fused epilogue.
 *   replaces  QKVProjection.forward        models/common/mixed_attn_block.py:669-676
 *             AnchorLinear.forward         models/common/mixed_attn_block.py:727-736 (pool_df > 1)
 *             MixedAttention.proj + norm1  models/common/mixed_attn_block_efficient.py:379,543-548
 *             Mlp.forward + norm2          models/common/swin_v1_block.py:37-43, efficient.py:554
 * ------------------------------------------------------------------------------------------- */
enum { GRL_EPI_PLAIN = 0, GRL_EPI_GELU = 1, GRL_EPI_GROUPNORM = 2, GRL_EPI_LN_RES = 3,
       GRL_EPI_GELU_GRAD = 4 /* ABI 22: out = (A W^T + bias) * gelu'(resid[m][col]) -- the data gradient through Mlp's activation; fp32 out */ };

typedef struct GrlLinearArgs {
    const void* a;          /* [M, lda] activations: GRL_DT_F32 (converted in-kernel) or GRL_DT_F16 */
    int32_t a_dtype;
    int64_t lda;            /* elements per A row (multiple of 8, >= Kpad)                          */
    int32_t pool_df;        /* >1: A row m is the mean of a df x df block of rows of a [B,H,W,lda]  */
    int32_t pool_H, pool_W; /*     fp32 image (AnchorLinear avg-pool); M = B*(H/df)*(W/df)          */
    const void* w;          /* fp16 [Npad, Kpad], zero padded (row n = output channel n)            */
    const float* bias;      /* [Npad]                                                               */
    int32_t M, Npad, Kpad;
    int32_t epi;            /* GRL_EPI_*                                                            */
    const float* gscale;    /* GROUPNORM: [Npad/32]; g!=0: L2-normalise the 32-col group, times |g|;*/
                            /*            g==0: pass through; g<0: additionally write 1.0 into      */
                            /*            column 31 of the group (spare head-dim slot of a K plane: */
                            /*            carries the attention kernel's running softmax offset)    */
    const float* ln_g;      /* LN_RES: gamma/beta [Npad], n_real real channels, eps, residual scale */
    const float* ln_b;
    int32_t n_real;
    float ln_eps;
    float res_scale;
    const float* resid;     /* LN_RES: fp32 residual [M, ldr]                                       */
    int64_t ldr;
    const void* add2;       /* LN_RES: optional extra branch (CAB output, GRL_DT_F16) added after   */
    int32_t add2_dtype;     /*         the norm, multiplied by add2_scale (required with add2)      */
    int64_t ldadd2;
    const float* add2_scale; /* [B, Npad] per-image channel scale applied to add2 (SE gate)           */
    int32_t rows_per_image;  /*          image of row m = m / rows_per_image                          */
    int32_t a_split;        /* 1 (default, 0 means 1) or 3: split-precision operands.  With 3, Kpad = 3 * Ksrc  */
                            /* and the kernel stages A (fp32) as [hi(a) | lo(a) | hi(a)], hi = fp16(a),         */
                            /* lo = fp16(a - hi); the caller packs W as [hi(w) | hi(w) | lo(w)]: the product    */
                            /* carries ~22 mantissa bits at 3x the MFMA work (small, error-amplifying models)   */
    float a_scale;          /* 0 means 1: fp32 A is multiplied by this before its conversion to fp16 and the          */
    float out_scale;        /* accumulator (before bias / epilogue) by out_scale (0 means 1): the backward pass feeds  */
                            /* gradients as A, pre-scaled into fp16 range, and un-scales the product                  */
    void* out;              /* [M, ldo]: GRL_DT_F32 or GRL_DT_F16 (attention operands, planes)       */
    int32_t out_dtype;
    int64_t ldo;
    int64_t out_plane_stride; /* >0: write 32-column groups as planes: element (m, c) goes to          */
                              /* out[(c/32)*out_plane_stride + m*32 + c%32]  (ldo ignored)             */
    void* out_lo;             /* optional, GRL_DT_F16 outputs: the rounding residual v - fp16(v) of every output value, */
                              /* same layout as out (the low half of a split-precision attention operand)               */
    const void* w_regs;       /* optional, a_split == 3: the weights once more, as the register image of the            */
                              /* weights-stationary kernel (csrc/linear_split.hip; plain / GELU / GROUPNORM epilogues,  */
                              /* fp32 A without pooling): per slab of 192 output columns and compute wave w = 0..5      */
                              /* (columns 192 slab + 32 w ..) 2 x Ksrc/16 MFMA A fragments of 1 KiB, hi = fp16(W) first, */
                              /* then lo = fp16(W - hi); fragment s, lane l, element e = W[column 32 (6 slab + w) +      */
                              /* (l & 31)][16 s + 8 (l >> 5) + e]; columns >= Npad zero.  grl_linear_split_blob_bytes;   */
                              /* ops.pack_linear_split.  NULL, or a shape it does not take: the generic kernel on `w`.   */
    /* ABI 22 -- operands and results at their REAL widths (training path: no padded copies of activations / gradients):  */
    int32_t a_cols;         /* > 0 (fp32 A, no pooling, a_split <= 1): A has a_cols real columns (multiple of 4), lda >= a_cols, */
                            /* lda % 4 == 0; the columns a_cols .. Kpad-1 of the operand are read as 0 ...                        */
    int32_t a_one;          /* ... except column a_cols, read as 1.0, when a_one != 0 (it meets a zero weight column in a        */
                            /* forward launch; the weight-gradient contraction finds the bias gradient there)                     */
    int32_t n_store;        /* > 0 (fp32 output, PLAIN / GELU epilogue, no planes): only the columns < n_store (multiple of 4)    */
                            /* are stored; ldo >= n_store, ldo % 4 == 0                                                           */
    int32_t a_gelu;         /* != 0 (fp32 A, no pooling, a_split <= 1): the operand is gelu(A), taken on the way to fp16 -- fc2 of the Mlp reads    */
                            /* fc1's pre-activation (swin_v1_block.py:37-43); constants of a_cols / a_one are not passed through it              */
    void* a16_out;          /* optional (fp32 A, a_split <= 1): the fp16 operand the kernel contracts -- a_scale * A, pad columns     */
    int64_t lda16;          /* and the a_one column included -- written as [M, lda16] (lda16 >= Kpad, multiple of 8): the weight-     */
                            /* gradient GEMM of the same layer reads it instead of converting the fp32 matrix once per output tile     */
} GrlLinearArgs;

int grl_linear_fwd(void* stream, const GrlLinearArgs* args);
int64_t grl_linear_split_blob_bytes(int32_t Npad, int32_t Ksrc);

/* ---------------------------------------------------------------------------------------------
 * Fused transformer MLP:  out = x + res_scale * LayerNorm(fc2(GELU(fc1(x))))  in one pass over x.
 *   replaces  Mlp.forward             models/common/swin_v1_block.py:37-43
 *             norm2 + residual         models/common/mixed_attn_block_efficient.py:554
 * The hidden activations never reach HBM and x is read once.  The weights arrive as a chunk stream ("blob"),
 * one chunk per 32 hidden channels c = 0 .. Hpad/32-1.  A chunk is the LDS image the kernel DMAs into its ring:
 *     W1c  32 rows x (2*Cpad + 16) bytes   fp16 rows 32c .. 32c+31 of fc1.weight (zero padded), columns in
 *                                          k-slot order per 32-group, 16 pad bytes per row
 *     W2c  Cpad rows x (64 + 16) bytes     fp16 columns of fc2.weight for the chunk's hidden channels in
 *                                          k-slot order, 16 pad bytes per row
 *     b1c  32 fp32                         fc1.bias for the chunk
 *   padded to a multiple of 1024 bytes; k-slot order of a 32-group: slot 8g+e <-> element 4g+e (e < 4),
 *   16+4g+e-4 (e >= 4) -- the 16x16x32 MFMA accumulator fragment of one layer is then the operand fragment of
 *   the next.  grl_mlp_blob_bytes gives the total size; ops.pack_mlp builds it; the blob must be 16-B aligned.
 * ------------------------------------------------------------------------------------------- */
typedef struct GrlMlpArgs {
    const float* x;         /* [M, ldx] fp32 input, also the residual; pad channels (>= n_real) must be 0 */
    int64_t ldx;
    const void* blob;       /* weight chunk stream, see above                                       */
    int32_t M, Cpad, Hpad;  /* Cpad in {64, 128, 192}; Hpad % 32 == 0                               */
    const float* b2;        /* [Cpad] fc2.bias                                                      */
    const float* ln_g;      /* [Cpad] norm2 weight / bias; n_real real channels                     */
    const float* ln_b;
    int32_t n_real;
    float ln_eps;
    float res_scale;
    float* out;             /* [M, ldo] fp32, pad channels = those of x (0); must not alias x       */
    int64_t ldo;
} GrlMlpArgs;

int grl_mlp_fwd(void* stream, const GrlMlpArgs* args);
int64_t grl_mlp_blob_bytes(int32_t Cpad, int32_t Hpad);

/* ---------------------------------------------------------------------------------------------
 * Block tail: everything of an EfficientMixAttnTransformerBlock after the attention and CAB kernels, in one pass:
 *     r1  = x + res_scale * LayerNorm1(att . Wp^T + bp) + cab * gate[image]     (never written to HBM)
 *     out = r1 + res_scale * LayerNorm2(fc2(GELU(fc1(r1))))
 *   replaces  MixedAttention.proj                          models/common/mixed_attn_block_efficient.py:379
 *             norm1 + residual + CAB branch + norm2 + Mlp   :543-556, models/common/swin_v1_block.py:37-43
 * att: fp16 slotted attention output [M, >= Cpad] (K = Cpad, natural order); cab: fp16 raw CAB conv output;
 * gate: [images, Cpad] SE gate (grl_se_scale_fwd); pblob: projection weight stream = Cpad/32 chunks, each 32 rows
 * (output channels) x (2*Cpad + 16) bytes fp16 padded to 1024 bytes (grl_proj_blob_bytes); blob as in GrlMlpArgs.
 * rows_per_image must be >= 128 (GRL_ERR_UNSUPPORTED otherwise).
 *
 * Register-resident variant (round 4, csrc/tail_regs.hip; GRL-Base shape: Cpad 192, Hpad 384, M and rows_per_image multiples
 * of 32, 160 < n_real <= 192): when `rblob` is given and the shape qualifies, the three weight matrices stay in the register
 * file of a persistent 8-wave workgroup as MFMA A fragments and only activations move (`blob` / `pblob` are then unused but
 * still have to be valid: other shapes fall back to the streaming kernel).  `rblob` (grl_tail_regs_blob_bytes() bytes, 16-B
 * aligned): 8 waves x 48 fragments x 1 KiB -- fragment f of wave w = 64 lanes x 16 B, lane l = 8 fp16 of row 32 t + (l & 31),
 * columns 16 s + 8 (l >> 5) + [0..8) of tile t, k-step s of a matrix: wave w < 6: f 0..11 proj tile w, 12..23 fc1 tile w,
 * 24..47 fc2 tile w (K = Hpad, natural order); waves 6, 7: f 0..35 = fc1 tiles 6..8 / 9..11 -- followed by the Hpad fp32 of fc1.bias.
 * ------------------------------------------------------------------------------------------- */
typedef struct GrlTailArgs {
    const void* att;
    int64_t ldatt;
    const float* x;         /* [M, ldx] fp32 block input (residual)                                 */
    int64_t ldx;
    const void* cab;
    int64_t ldcab;
    const float* gate;
    int32_t rows_per_image;
    const void* pblob;
    const float* pb;        /* [Cpad] proj.bias, norm1 weight / bias                                */
    const float* n1_g;
    const float* n1_b;
    const void* blob;       /* MLP weight stream (GrlMlpArgs)                                       */
    int32_t M, Cpad, Hpad;
    const float* b2;        /* [Cpad] fc2.bias, norm2 weight / bias                                 */
    const float* n2_g;
    const float* n2_b;
    int32_t n_real;
    float ln_eps;
    float res_scale;
    float* out;             /* [M, ldo] fp32; must not alias x                                      */
    int64_t ldo;
    const void* rblob;      /* NULL or the register-resident weight image (see above)               */
} GrlTailArgs;

int grl_block_tail_fwd(void* stream, const GrlTailArgs* args);
int64_t grl_tail_regs_blob_bytes(void);
int64_t grl_proj_blob_bytes(int32_t Cpad);

/* ---------------------------------------------------------------------------------------------
 * Streaming QKV projection: head planes out[slot][m][0..31] (fp16) = groupnorm(x[m,:] . W_slot^T + b_slot),
 * one pass over x for any number of slots (the weights-resident grl_linear_fwd needs column slabs above 160 KB).
 *   replaces  QKVProjection.forward   models/common/mixed_attn_block.py:669-676
 *             F.normalize + logit scale of Attention.attn   models/common/mixed_attn_block_efficient.py:39,85-90
 * A slot = 32 output columns (one head of q, k or v; see grl_attention_fwd).  Weight stream ("blob"): chunks of
 * 2 slots (1 if nslots is odd), each slot image = 32 rows x (2*Cpad + 16) bytes fp16 (columns in the k-slot order
 * of GrlMlpArgs, 16 pad bytes per row) | bias 32 fp32 | gscale fp32 + 12 pad bytes; chunk padded to 1024 bytes.
 * gscale as in GRL_EPI_GROUPNORM: != 0 -> L2-normalise the slot and multiply by |gscale|, == 0 -> pass through,
 * < 0 -> additionally column 31 of the slot is written as 1.0 (K planes, see grl_attention_fwd).
 * ------------------------------------------------------------------------------------------- */
typedef struct GrlQkvArgs {
    const float* x;            /* [M, ldx] fp32 tokens                                               */
    int64_t ldx;
    const void* blob;          /* see above; 16-B aligned; grl_qkv_blob_bytes(Cpad, nslots) bytes    */
    int32_t M, Cpad, nslots;   /* Cpad in {64, 128, 192}                                             */
    void* out;                 /* fp16 planes: element (m, slot, c) at slot*out_plane_stride + m*32 + c */
    int64_t out_plane_stride;  /* >= M*32                                                            */
} GrlQkvArgs;

int grl_qkv_fwd(void* stream, const GrlQkvArgs* args);
int64_t grl_qkv_blob_bytes(int32_t Cpad, int32_t nslots);

/* ---------------------------------------------------------------------------------------------
 * QKV projection AND anchor projection in one pass over x (round 3; csrc/qkv_anchor.hip).
 *   replaces  QKVProjection.forward               models/common/mixed_attn_block.py:661-676
 *             AnchorProjection / AnchorLinear     models/common/mixed_attn_block.py:714-736,739-785
 *             (avg_pool2d(2) + Linear C -> C/2; the pool commutes with the linear map and is applied to its outputs)
 *             F.normalize / logit scale of Attention.attn   models/common/mixed_attn_block_efficient.py:85-90,:39
 * x is the token matrix of B images of H x W tokens (H even, W a multiple of 64; other shapes: GRL_ERR_UNSUPPORTED ->
 * grl_qkv_fwd + grl_linear_fwd with pooling).  Anchor tokens are the 2 x 2 pooling cells in row-major order.
 * Weight stream `blob`: chunks of 2 slots, padded to 1 KiB; a slot = 32 rows x (2*Cpad + 16) bytes fp16 (row n = output column
 * n of the slot, K in natural order) | 32 fp32 bias | 1 fp32 gscale (+12 B); slots 0..nslots-1 = q/k/v, then nanc anchor slots.
 * gscale as in grl_qkv_fwd.
 *
 * Split-precision variant (round 4; `lo_blob` != NULL, GRL-Base shape only: Cpad 192, 18 + 3 slots, otherwise
 * GRL_ERR_UNSUPPORTED): for the normalised slots (gscale != 0: q, k, anchors) the projection is evaluated as
 *     W_hi . x_hi  +  W_hi . x_lo  +  2^-e W_lo8 . x_8,
 * W_hi = fp16(W) from `blob`, x_hi = fp16(x), x_lo = fp16(x - x_hi), W_lo8 = e4m3((W - W_hi) 2^(e+4)), x_8 = e4m3(clamp(x_hi / 16, +-448)):
 * the fp16 rounding of BOTH operands is carried (~2^-15 relative instead of 2^-11) at the cost of one more fp16 MFMA term on
 * register-resident weights and one fp8 MFMA term whose weights live in LDS.  Needed for checkpoints whose logit scales sit
 * near the clamp (exp(min(logit_scale, ln 100)), mixed_attn_block_efficient.py:39): the q.k logits multiply the operand
 * rounding by up to 144.  `lo_blob`: float 2^-e | float 2^e | 8 pad bytes | one image per normalised slot in slot order,
 * 32 rows x 192 e4m3 bytes: byte 64 c + 32 h + 8 u + t of row j = channel 64 c + 16 u + 8 h + t (the order in which one
 * 32x32x64 fp8 MFMA consumes a lane half's 32 bytes), and the 16-byte segments of a row are stored XOR-swizzled, segment s
 * at position s ^ ((j >> 2) & 3) (the image is copied to LDS verbatim).  Pass-through slots (gscale == 0: v) are not split.
 * ------------------------------------------------------------------------------------------- */
typedef struct GrlQkvAnchorArgs {
    const float* x;            /* [B*H*W, ldx] fp32 tokens                                            */
    int64_t ldx;
    int32_t B, H, W;
    int32_t Cpad;              /* 64, 128 or 192                                                      */
    const void* blob;          /* 16-B aligned; grl_qkv_anchor_blob_bytes(Cpad, nslots, nanc) bytes   */
    int32_t nslots, nanc;
    void* out;                 /* fp16 planes: element (m, slot, c) at slot*out_plane_stride + m*32 + c */
    int64_t out_plane_stride;  /* >= B*H*W*32                                                         */
    void* anc;                 /* fp16 planes of the anchors: (a, slot, c) at slot*anc_plane_stride + a*32 + c (nanc > 0) */
    int64_t anc_plane_stride;  /* >= B*(H/2)*(W/2)*32                                                 */
    const void* lo_blob;       /* NULL: fp16 operands; else the fp8 low parts of the normalised slots' weights (see above), 16-B aligned */
} GrlQkvAnchorArgs;

int grl_qkv_anchor_fwd(void* stream, const GrlQkvAnchorArgs* args);
int64_t grl_qkv_anchor_blob_bytes(int32_t Cpad, int32_t nslots, int32_t nanc);
int64_t grl_qkv_anchor_lo_blob_bytes(int32_t Cpad, int32_t nsplit);   /* nsplit = number of slots with gscale != 0 */

/* ---------------------------------------------------------------------------------------------
 * Cosine window / anchored-stripe attention (one call = one softmax(QK^T)V over all windows).
 *   replaces  WindowAttention.forward        models/common/mixed_attn_block_efficient.py:128-165
 *             AnchorStripeAttention.forward  models/common/mixed_attn_block_efficient.py:215-270
 *             (called twice: anchors->window tokens, then window tokens->anchors)
 *             Attention.attn / AffineTransform  :77-94 / :36-58
 *             roll / window_partition / window_reverse / masks / relative index
 *                                            models/common/ops.py:36-157,352-375 (all as index math)
 * Operands are fp16 token tensors with one 32-wide slot per head (fp32 accumulation):
 *   q: already L2-normalised and multiplied by logit_scale*log2(e); k: L2-normalised;
 *   v: raw values, slot column `ones_col` (>= head_dim) holding 1.0 so that the row sum of the
 *      softmax weights falls out of the PV product (ones_col < 0: summed explicitly).
 * Softmax offset (keeps the fp16 weights in range for ANY logit scale up to the clamp exp(ln 100)): a running offset that
 * follows the row maximum.  32-aligned geometries evaluate it lazily (only when a weight reaches 2^14) and keep it in
 * head-dim slot 31 of q: K must then hold 1.0 in slot 31 (`k_one31`, head_dim <= 30) and `lazy_floor[head]` (an integer
 * valued lower bound of every unmasked logit, log2 domain) must be given; otherwise the generic kernel runs an ordinary
 * online softmax.
 * ------------------------------------------------------------------------------------------- */
typedef struct GrlTokenGrid {
    const void* ptr;      /* fp16 base (o: fp16 or fp32 per out_dtype)                              */
    int64_t ld;           /* elements between consecutive tokens                                    */
    int64_t hstride;      /* elements between the 32-wide slots of consecutive heads: 32 for a      */
                          /* token-major matrix [tokens, heads*32]; tokens*32 for head planes       */
                          /* [heads][tokens][32] (the layout the QKV projection writes: a key tile  */
                          /* of 32 consecutive tokens is then 2 KB contiguous)                      */
    int32_t col0;         /* element offset of head 0's slot                                        */
    int32_t Himg, Wimg;   /* token image size                                                       */
    int32_t wh, ww;       /* window (stripe) size on this grid                                      */
    int32_t shy, shx;     /* cyclic shift (rolled[y] = orig[(y+sh) % H]); 0 = none                  */
    int32_t transposed;   /* 0: token (y, x) of image b is row (b*Himg + y)*Wimg + x of the matrix.  */
                          /* 1: the grid is the TRANSPOSED VIEW of a (Wimg x Himg) row-major image:  */
                          /* token (y, x) is row (b*Wimg + x)*Himg + y.  Attention is invariant under */
                          /* a joint transposition of the image axes (with wh/ww, shy/shx swapped and */
                          /* the bias table transposed by the caller), which brings a 128x64 stripe   */
                          /* with 32x16 anchors onto the 32-aligned fast path as 64x128 / 16x32.      */
                          /* All four grids of a launch must agree.  Forward only.                    */
} GrlTokenGrid;

typedef struct GrlAttnArgs {
    GrlTokenGrid q, k, v, o; /* v shares k's grid geometry; o shares q's                            */
    int32_t B, nh;
    int32_t nwy, nwx;        /* windows per image (same on both grids)                              */
    const float* table;      /* [nh, tstride] fp32: bias*log2e,                                     */
                             /* stored REVERSED: entry trows-1-i is row i of the reference's         */
                             /* relative-position table (keys then read ascending addresses)         */
    int32_t trows;           /* (q.wh + k.wh - 1) * (q.ww + k.ww - 1)                               */
    int32_t tstride;         /* floats between heads: trows rounded up to a multiple of 4           */
    int32_t masked;          /* 1: apply the shifted-window region mask (-100)                      */
    int32_t ones_col;        /* see above                                                           */
    int32_t head_dim;        /* real head dim (<= 32)                                               */
    int32_t out_dtype;       /* GRL_DT_F16 or GRL_DT_F32                                            */
    int32_t k_one31;         /* 1: every K row holds 1.0 in head-dim slot 31 (lazy offset possible) */
    const float* lazy_floor; /* [nh] see above (may be NULL: generic kernel)                        */
    float* lse;              /* optional [nh][lse_stride] fp32: log2-sum-exp2 of the kernel-domain  */
    int64_t lse_stride;      /* logits per query token row (what the backward kernel re-normalises with) */
    const void* q_lo;        /* optional split-precision operands (precision "high"): fp16 residuals q - fp16(q), ... on   */
    const void* k_lo;        /* the grids of q, k, v (same ld / hstride / col0).  With any of them the generic kernel forms */
    const void* v_lo;        /* S = q_hi k_hi + q_lo k_hi + q_hi k_lo and O = P_hi v_hi + P_lo v_hi + P_hi v_lo (~22-bit operands; P_lo = p - fp16(p))            */
    void* o_lo;              /* optional, GRL_DT_F16 output: residual o - fp16(o) on o's grid (the next attention's v_lo)    */
    const float* lazy_ceil;  /* optional [nh]: upper bound of every logit of the head (log2 domain: ceil(scale*log2e) + max of its   */
                             /* table).  Once the running offsets of a wave's queries are within 13.5 of it no weight can reach     */
                             /* 2^14 any more and the row-streaming kernel stops testing for it (NULL: always test)                 */
} GrlAttnArgs;

int grl_attention_fwd(void* stream, const GrlAttnArgs* args);

/* 1 when grl_attention_fwd would serve these arguments with the row-streaming kernel (csrc/attention_rows.hip): 32-aligned
 * window widths, table window within its LDS buffer, 16-aligned shifts.  Reads the geometry fields, head_dim, ones_col and
 * masked only (pointers may be NULL): lets the host choose between a grid and its transposed view at plan time. */
int grl_attention_rows_geometry_ok(const GrlAttnArgs* args);

/* ---------------------------------------------------------------------------------------------
 * 3x3 convolution (stride 1, zero pad 1) on channels-last token matrices, fused epilogue.
 *   replaces  CAB.cab[0], GELU, CAB.cab[2]          models/common/mixed_attn_block.py:970-983
 *             TransformerStage.conv (+ residual)     models/networks/grl.py:137,168-170
 *             conv_first / conv_after_body / conv_before_upsample / Upsample convs + PixelShuffle /
 *             conv_last                              models/networks/grl.py:293,348,352-379,515-518
 *                                                    models/common/upsample.py:16-19,45-46
 * ------------------------------------------------------------------------------------------- */
typedef struct GrlConvArgs {
    const void* x;          /* [B*H*W, ldx] GRL_DT_F32 or GRL_DT_F16, channels-last                 */
    int32_t x_dtype;
    int64_t ldx;
    const void* w;          /* fp16 [9][*][CinP] (tap = ky*3+kx): first output channel of this call */
    int64_t w_tap_stride;   /* elements between taps (= full layer Cout_pad * CinP)                  */
    const float* bias;      /* [CoutP]                                                              */
    int32_t B, H, W;
    int32_t CinP, CoutP;    /* CinP % 32 == 0, CoutP % 16 == 0, CoutP <= 192 per call               */
    int32_t x_split;        /* 1 (0 means 1) or 3: split-precision operands as in GrlLinearArgs.a_split: CinP =     */
                            /* 3 * CinSrc, x (fp32, CinSrc wide) is staged as [hi | lo | hi], w packed [hi | hi | lo] */
    float x_scale;          /* 0 means 1: fp32 x is multiplied by this before its conversion to fp16, the accumulator */
    float out_scale;        /* (before bias) by out_scale (0 means 1) -- gradient inputs of the backward pass         */
    int32_t act;            /* 0 none, 1 exact GELU, 2 LeakyReLU(slope)                             */
    float slope;
    const float* resid;     /* optional fp32 [B*H*W, ldr] added after the activation                */
    int64_t ldr;
    float* pool_partial;    /* optional [grl_conv3x3_num_workgroups(B,H,W), pool_stride]: per-       */
                            /* workgroup channel sums of the output (two-stage global average pool) */
    int64_t pool_stride;    /* floats per workgroup row (>= CoutP; the full layer's channel count    */
                            /* when the layer is computed as several output-channel slabs)          */
    void* out;              /* [rows, ldo] GRL_DT_F32 or GRL_DT_F16.  16-bit outputs without residual /  */
                            /* pixel shuffle: channels CoutP .. round_up(CoutP, 32) are written as zeros  */
                            /* when ldo is at least that wide (the consumer's K is padded to 32)          */
    int32_t out_dtype;
    int64_t ldo;
    int32_t shuffle_r;      /* >1: PixelShuffle(r) store into a [B, H*r, W*r, shuffle_cg] matrix;    */
    int32_t shuffle_cg;     /*     output channels are packed in (i, j, c) order, this call's        */
    int32_t shuffle_ij0;    /*     channel 0 belongs to sub-pixel group shuffle_ij0                  */
    /* ABI 22 (see GrlLinearArgs.a_cols / n_store): */
    int32_t x_cols;         /* > 0 (fp32 x, x_split <= 1): x has x_cols real channels (multiple of 4), ldx >= x_cols,   */
                            /* ldx % 4 == 0; channels x_cols .. CinP-1 are read as 0                                    */
    int32_t n_store;        /* > 0 (fp32 output, no pixel shuffle): only channels < n_store (multiple of 4) are stored  */
} GrlConvArgs;

int grl_conv3x3_fwd(void* stream, const GrlConvArgs* args);
int grl_conv3x3_num_workgroups(int32_t B, int32_t H, int32_t W);

/* CAB.cab[2] of the GRL-Base shape with the filter bank resident in registers (csrc/cab_conv2.hip): the same result as
 * grl_conv3x3_fwd on a 16-bit input with <= 48 (padded) input and 192 (padded) output channels, bias, no activation,
 * 16-bit output + the partial channel sums of the SE pool.   models/common/mixed_attn_block.py:948-983
 *   blob: grl_cab_conv2_blob_bytes() bytes, fp16 [12 channel groups][14 k-steps][64 lanes][8]: lane (g4 = lane / 16,
 *         r = lane % 16) of (group G, k-step s) holds W[16 G + r][k = 32 s + 8 g4 .. + 7], k = (ky * 3 + kx) * 48 + cin,
 *         zero for cin >= Cin, k >= 432 and output channels >= Cout.
 *   pool_partial: optional [B * wgs_per_image, pool_stride] -- one row of channel sums per workgroup (grl_se_scale_fwd's input
 *         with this wgs_per_image).  wgs_per_image persistent workgroups share the 8 x 32 pixel tiles of an image. */
typedef struct GrlCabConv2Args {
    const void* x;          /* GRL_DT_F16 [B*H*W, ldx], ldx >= 56, channels >= Cin are zero                  */
    int64_t ldx;
    const void* blob;
    const float* bias;      /* [192], zero beyond Cout                                                       */
    int32_t B, H, W;
    int32_t wgs_per_image;
    void* out;              /* GRL_DT_F16 [B*H*W, ldo], ldo >= 192: all 192 channels are written            */
    int64_t ldo;
    float* pool_partial;
    int64_t pool_stride;    /* >= 192 */
    /* optional squeeze-excite gate (ChannelAttention.attention, mixed_attn_block.py:956-963) computed by the last workgroup   */
    /* of every image from the partial sums: gate[b][c] = sigmoid(W2 . relu(W1 . mean_b + b1) + b2), 0 for c >= se_c           */
    float* gate;            /* [B, 192] or NULL (then the fields below are ignored)                           */
    int32_t* se_counter;    /* [B] zero-initialised once; the kernel leaves it zero (one launch at a time per buffer) */
    const float* se_w1;     /* [se_mid, se_c] */
    const float* se_b1;     /* [se_mid]       */
    const float* se_w2;     /* [se_c, se_mid] */
    const float* se_b2;     /* [se_c]         */
    int32_t se_c, se_mid;   /* <= 192, <= 64  */
    float inv_hw;           /* 1 / (H * W)    */
} GrlCabConv2Args;

int grl_cab_conv2_fwd(void* stream, const GrlCabConv2Args* args);
int64_t grl_cab_conv2_blob_bytes(void);

/* Squeeze-excite gate from the pooled sums: scale[b,c] = sigmoid(W2 relu(W1 mean_b + b1) + b2)
 *   replaces  ChannelAttention.attention   models/common/mixed_attn_block.py:956-967            */
int grl_se_scale_fwd(void* stream, const float* pool_partial, int32_t B, int32_t wgs_per_image, int32_t CP, int32_t C,
                     int32_t Cmid, int32_t HW, const float* w1, const float* b1, const float* w2, const float* b2,
                     float* scale);

/* ---------------------------------------------------------------------------------------------
 * Row LayerNorm on a padded token matrix (norm_start / norm_end, models/networks/grl.py:494,501).
 * ------------------------------------------------------------------------------------------- */
int grl_layernorm_fwd(void* stream, const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma,
                      const float* beta, int32_t M, int32_t n_real, int32_t n_pad, float eps);

/* y = resid + res_scale * LayerNorm(x) (+ add2 * add2_scale[image]): the un-fused form of GRL_EPI_LN_RES
 *   replaces  norm1 / norm2 + residual (+ CAB branch)   models/common/mixed_attn_block_efficient.py:543-556
 * used by the split-precision path, whose slab-split projections cannot carry the row norm in their epilogue. */
typedef struct GrlLnResArgs {
    const float* x;         /* [M, ldx] fp32 projection output                                       */
    int64_t ldx;
    const float* resid;     /* [M, ldr] fp32                                                         */
    int64_t ldr;
    const float* gamma;     /* [n_pad]                                                               */
    const float* beta;
    const void* add2;       /* optional [M, ldadd2] GRL_DT_F32 or GRL_DT_F16, times add2_scale[image][n_pad] */
    int32_t add2_dtype;
    int64_t ldadd2;
    const float* add2_scale;
    int32_t rows_per_image;
    int32_t M, n_real, n_pad; /* n_pad <= 256, multiple of 4; pad channels written as 0              */
    float eps, res_scale;
    float* y;               /* [M, ldy] fp32                                                         */
    int64_t ldy;
} GrlLnResArgs;

int grl_layernorm_res_fwd(void* stream, const GrlLnResArgs* args);

/* ---------------------------------------------------------------------------------------------
 * Training path (BASELINE config 5; reference: engines/base.py:221-236, autograd through the modules above).
 *
 * The data gradients of the linears and convolutions re-use grl_linear_fwd / grl_conv3x3_fwd with transposed / flipped
 * weights (a_scale / x_scale bring the fp32 gradients into fp16 range).  The entry points below are the remaining
 * contractions of the backward pass and the optimizer step.
 * ------------------------------------------------------------------------------------------- */

/* Weight gradient: c[tap][n][k] += out_scale * sum_m (a_scale * a[m][n]) * b[row(m, tap)][k]   (fp32 atomics; zero c first)
 *   taps = 1: row = m (token-wise linear: dW = dY^T X)
 *   taps = 9: 3x3 convolution, a / b are [images*H*W, ld] channels-last pixel matrices, tap = (dy+1)*3 + (dx+1),
 *             row = pixel (y + dy, x + dx) of the same image, zero outside: dW[tap][co][ci]                          */
typedef struct GrlGemmTnArgs {
    const float* a;         /* [M, lda] fp32 (output gradient)                                      */
    int64_t lda;
    const void* b;          /* [M, ldb] GRL_DT_F32 or GRL_DT_F16 (the layer's input)                */
    int32_t b_dtype;
    int64_t ldb;
    int32_t M, N, K;        /* N, K multiples of 4 (of 8 for fp16 b)                                */
    int32_t taps, H, W;     /* 1, or 9 with the image size                                          */
    int32_t splits;         /* M is cut into this many slabs (one workgroup column each)            */
    float a_scale, out_scale;
    float* c;               /* [taps][N][ldc] fp32                                                  */
    int64_t ldc, c_tap_stride;
    int64_t* c_fix;         /* optional (ABI 21): deterministic accumulation.  The slabs' partial tiles are added as 64-bit  */
                            /* fixed point (2^30 x the a_scale-d sums; integer atomics commute, fp32 atomics do not) into     */
                            /* this zeroed [taps][N][ldc] array instead of `c`; the caller converts:                          */
                            /* c = c_fix * 2^-30 * out_scale.  Bit-identical results run to run.                              */
    /* ABI 22: a / b at their real widths -- N, K multiples of 4 (fp32 operands), lda >= N, ldb >= K, nothing is read beyond   */
    /* column N / K of a row -- and the bias gradient without a ones column in memory:                                          */
    int32_t b_ones;         /* != 0: b has a VIRTUAL column K that holds 1.0 (taps = 9: inside the image); its products, the   */
    int32_t reserved0;      /* column sums of a, go to c_bias[n] (taps = 9: of the centre tap) instead of a column of c         */
                            /* (reserved0: ignored -- the entry point uses the slot to tell the kernel its tile order)          */
    float* c_bias;          /* [N] fp32, zeroed by the caller (required with b_ones unless c_bias_fix)                          */
    int64_t* c_bias_fix;    /* [N], the deterministic counterpart (with c_fix)                                                  */
    int32_t a_dtype;        /* 0 / GRL_DT_F32: a is fp32 (the field `a`); GRL_DT_F16: a points at fp16 values that are ALREADY     */
    int32_t reserved1;      /* multiplied by a_scale (GrlLinearArgs.a16_out of the data-gradient launch).  16-bit operands: lda /  */
                            /* ldb multiples of 8 and at least N / K rounded up to 8 (the pad columns are read)                    */
} GrlGemmTnArgs;

int grl_gemm_tn(void* stream, const GrlGemmTnArgs* args);

/* Cosine attention backward (flash-style recompute from q, k, v, the forward's output and its log2-sum-exp2):
 *   replaces autograd through Attention.attn / AffineTransform  models/common/mixed_attn_block_efficient.py:36-58,77-94
 * Inputs as grl_attention_fwd (`fwd`: the same argument block, o = the forward OUTPUT (GRL_DT_F32), lse required) plus the
 * output gradient d_o (fp32, o's grid).  Outputs (fp32, gradients w.r.t. the kernel's own operands -- the normalisation,
 * the logit scale and the CPB-MLP are differentiated by the caller): d_q, d_k, d_v on the grids of q, k, v (same
 * ld / hstride / col0 in ELEMENTS as the fp16 operands), d_table [nh, tstride] (accumulated with atomics: zero it first).
 * g_scale brings d_o into fp16 range for the contractions (results are un-scaled). */
typedef struct GrlAttnBwdArgs {
    GrlAttnArgs fwd;
    const float* d_o;
    float* d_q;
    float* d_k;
    float* d_v;
    float* d_table;
    float g_scale;
    int64_t* d_table_fix;   /* optional (ABI 21): deterministic mode.  Non-null: no split launches (their partial dQ / dK / dV  */
                            /* meet in fp32 atomics), the dq kernel runs one wave per workgroup (its LDS histogram is then      */
                            /* filled in program order) and the workgroups' tables are added as 64-bit fixed point (2^32 x the   */
                            /* g_scale-d sums) into this zeroed [nh, tstride] array instead of `d_table`; the caller converts:   */
                            /* d_table = d_table_fix * 2^-32 / g_scale.  Bit-identical gradients run to run.                     */
    int64_t d_o_ld;         /* optional (ABI 22): row stride of d_o in floats when it differs from o's (0 = o's): d_o as a column  */
                            /* block of a wider gradient matrix -- the two branches' outputs are concatenated along the channels     */
                            /* before the projection (mixed_attn_block_efficient.py:374-379), so their gradients arrive that way     */
} GrlAttnBwdArgs;

int grl_attention_bwd(void* stream, const GrlAttnBwdArgs* args);

/* torch.optim.AdamW over a list of fp32 tensors in one launch (config/optimizer/adamw.yaml):
 *   w *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  w -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 * Host-built work list: chunk c covers elements [chunk_offset[c], +4096) of tensor chunk_tensor[c].  All pointer arrays
 * live in DEVICE memory. */
typedef struct GrlAdamWArgs {
    void* const* params;
    const void* const* grads;
    void* const* exp_avg;
    void* const* exp_avg_sq;
    const int64_t* numel;
    const int32_t* weight_decay_flags;   /* optional per tensor: 0 = no decay                        */
    const int32_t* chunk_tensor;
    const int64_t* chunk_offset;
    int32_t num_chunks;
    float lr, beta1, beta2, eps, weight_decay;
    float bias_correction1, bias_correction2_sqrt;
    float grad_scale;                    /* gradients are multiplied by this (1 / loss scale, DDP averaging) */
    const float* bias_corrections_dev;   /* optional: {bias_correction1, bias_correction2_sqrt} in DEVICE memory, computed on the */
                                         /* device from a device-side step count; the two host fields above are then ignored --   */
                                         /* a launch captured in a HIP graph stays correct when it is replayed                    */
    const float* hyper_dev;              /* optional (ABI 20): {lr, weight_decay} in DEVICE memory; the host fields `lr` and           */
                                         /* `weight_decay` are then ignored -- an LR scheduler (the reference trains with MultiStepLR / */
                                         /* cosine / warm-up schedules, config/lr_scheduler) keeps working across replays of a captured  */
                                         /* launch: the caller refreshes the two floats before each replay                              */
} GrlAdamWArgs;

int grl_adamw_step(void* stream, const GrlAdamWArgs* args);

/* Row LayerNorm of the training path (ABI 21): forward with the row statistics kept, and backward.
 *   replaces nn.LayerNorm on token matrices and autograd through it: norm1 / norm2  models/common/mixed_attn_block_efficient.py:543-556,
 *   norm_start / norm_end  models/networks/grl.py:494,501.  n <= 256 channels, a multiple of 4; row strides in elements, multiples of 4.
 * forward: y, mean[M], rstd[M];  backward: dx, and dgamma / dbeta ACCUMULATED with atomics into arrays the caller zeroes. */
typedef struct GrlLnTrainArgs {
    const float* x;  int64_t ldx;
    const float* gamma;
    const float* beta;       /* forward only */
    float* y;        int64_t ldy;      /* forward only */
    float* mean;             /* [M] written by the forward, read by the backward */
    float* rstd;
    const float* dy; int64_t lddy;     /* backward only */
    float* dx;       int64_t lddx;
    float* dgamma;
    float* dbeta;
    int32_t M, n;
    float eps;
    /* ABI 22 -- the post-norm residual of a block in the same pass (mixed_attn_block_efficient.py:543-556: x + res_scale *    */
    /* DropPath(norm(f(x)))):  forward, resid != NULL: y = resid + c_row * LayerNorm(x);  backward, alpha != 0: dy is dL/dy of   */
    /* that sum and is multiplied by c_row first (dL/dresid = dy is the caller's).  c_row = alpha * row_scale[row /              */
    /* rows_per_image] (row_scale: one keep-mask entry per image, NULL = 1).                                                     */
    const float* resid; int64_t ldr;
    const float* row_scale;
    int32_t rows_per_image;
    float alpha;
    int32_t stat_replicas;   /* backward: dgamma / dbeta are [stat_replicas][n] arrays each (0 = 1); workgroup b adds into replica      */
    int32_t reserved0;       /* b % stat_replicas and the caller sums the replicas: a few thousand atomics on ONE address serialise      */
                             /* (~36 ns each on MI355X: 2048 workgroups on 180 + 180 addresses were 75 us of a 20-us kernel)              */
} GrlLnTrainArgs;

int grl_layernorm_train_fwd(void* stream, const GrlLnTrainArgs* args);
int grl_layernorm_bwd(void* stream, const GrlLnTrainArgs* args);

/* Weight / bias packing of a 3x3 convolution for grl_conv3x3_fwd in one launch (ABI 21, training path: the weights move every step):
 *   w [Cout][Cin][3][3] fp32 (nn.Conv2d.weight: grl.py:137,293,348; mixed_attn_block.py:970-983) -> out_w fp16 [9][rows_pad][cols_pad],
 *   tap = ky*3+kx, zero padded; flip_t = 0: rows = output channels, columns = input channels (the forward operand);
 *   flip_t = 1: the data-gradient operand out_w[tap][ci][co] = w[co][ci][2-ky][2-kx] (rows = Cin side, columns = Cout side).
 *   out_b (optional): fp32 [rows_pad] = b padded with zeros (b may be NULL: all zeros). */
int grl_pack_conv3x3(void* stream, const float* w, const float* b, void* out_w, float* out_b, int32_t Cout, int32_t Cin,
                     int32_t rows_pad, int32_t cols_pad, int32_t flip_t);

/* nn.Linear weight for grl_linear_fwd and for its data-gradient launch in one launch (ABI 22, training path):
 *   w [N][K] fp32 (mixed_attn_block.py:669-676, 727-736; mixed_attn_block_efficient.py:379; swin_v1_block.py:29-33)
 *   -> out_w fp16 [Np][Kp] zero padded, (optional) out_wt fp16 [Kp][Np], its transpose, and (optional) out_b fp32 [Np] = the bias b [N]
 *   zero padded (b NULL: zeros). */
int grl_pack_linear(void* stream, const float* w, const float* b, void* out_w, void* out_wt, float* out_b, int32_t N, int32_t K, int32_t Np,
                    int32_t Kp);

/* out[i] = a[i] + b[i] (+ c[i]) (+ d[i]) on flat fp32 arrays (ABI 22; 16-byte aligned, n a multiple of 4; c, d optional): the gradient of a
 * tensor with several consumers in one launch (a block's input x feeds QKVProjection, AnchorLinear, the CAB and the residual:
 * mixed_attn_block_efficient.py:539-556) instead of autograd's pairwise adds. */
int grl_sum4(void* stream, const float* a, const float* b, const float* c, const float* d, float* out, int64_t n);

/* Head planes of the training path (ABI 21): projection output x [T, S_in, nh, d] (fp32) -> the attention operands, fp32 planes
 * out32 [S_out][nh][T][32] and their fp16 copy out16, forward; dx and the scale gradients, backward.
 *   replaces F.normalize(q) * exp(min(logit_scale, ln 100)), F.normalize(k) and the head reshape / permute of
 *   models/common/mixed_attn_block_efficient.py:36-47,85-90,147-150,240-250 and autograd through them.
 * Output slot s reads input slot src[s];  raw[s] != 0: copied as it is (v), else y = x * scale[s][h] / max(|x|, 1e-12);
 * one_col[s] >= 0: that column of the plane holds 1.0 (pad columns are otherwise 0).  d even, <= 32; S_in, S_out, nh <= 8.
 * Backward: dy[s] = gradient of output slot s as an fp32 plane [nh][T][32] (NULL: none); dx [T, S_in, nh, d] is written, dscale
 * [S_out][nh] ACCUMULATED (zero it first) for the slots with want_dscale[s] != 0. */
typedef struct GrlPlanesArgs {
    const float* x;
    const float* scale;      /* [S_out][nh] (ignored for raw slots) */
    float* out32;            /* forward; optional (NULL: only out16 is written) */
    void* out16;             /* forward: fp16 [S_out][nh][T][32] */
    const float* dy[8];      /* backward */
    float* dx;
    float* dscale;
    int32_t T, S_in, S_out, nh, d;
    int32_t src[8], raw[8], one_col[8], want_dscale[8];
    int32_t dscale_replicas; /* backward (ABI 22): dscale is [dscale_replicas][S_out][nh] (0 = 1), workgroup b adds into replica b % replicas,   */
    int32_t reserved0;       /* the caller sums them (see GrlLnTrainArgs.stat_replicas)                                                          */
} GrlPlanesArgs;

int grl_head_planes_fwd(void* stream, const GrlPlanesArgs* args);
int grl_head_planes_bwd(void* stream, const GrlPlanesArgs* args);

/* Squeeze-excite gate MLP of the CAB in a training step (ABI 22), forward and backward:
 *   replaces  ChannelAttention.attention[1..4]  models/common/mixed_attn_block.py:956-963  (Conv2d(C, C/r, 1) -> ReLU -> Conv2d(C/r, C, 1)
 *   -> Sigmoid on the pooled [B, C] means) and autograd through it.  C <= 256, Cmid <= 64; all arrays fp32, contiguous.
 * forward: hidden [B, Cmid] = relu(w1 pool + b1), gate [B, C] = sigmoid(w2 hidden + b2);  backward: d_pool [B, C] and the parameter
 * gradients d_w1 [Cmid, C], d_b1 [Cmid], d_w2 [C, Cmid], d_b2 [C] (see `parallel`). */
typedef struct GrlSeMlpArgs {
    const float* pool;       /* [B, C] */
    const float* w1;         /* [Cmid, C] */
    const float* b1;
    const float* w2;         /* [C, Cmid] */
    const float* b2;
    float* gate;             /* forward output / backward input */
    float* hidden;           /* forward output / backward input */
    const float* d_gate;     /* backward */
    float* d_pool;
    float* d_w1;
    float* d_b1;
    float* d_w2;
    float* d_b2;
    int32_t B, C, Cmid;
    int32_t parallel;        /* backward: != 0: one workgroup per image that ADDS the parameter gradients with atomics into              */
                             /* arrays the caller zeroed; 0: one workgroup walks the batch and writes them (bit-reproducible, serial)    */
} GrlSeMlpArgs;

int grl_se_mlp_fwd(void* stream, const GrlSeMlpArgs* args);
int grl_se_mlp_bwd(void* stream, const GrlSeMlpArgs* args);

/* The two passes over the token matrix around that MLP (ABI 22; [M, ld] fp32 matrices, rows 16-byte aligned, C <= 256 a multiple of 4,
 * image of row m = m / rows_per_image):
 *   grl_se_colsum: out[b][c] += k * sum over the rows of image b of a[row][c] (* f[row][c] when f != NULL); out [M / rows_per_image, C],
 *                  zeroed by the caller (atomics).  AdaptiveAvgPool2d(1) of ChannelAttention (mixed_attn_block.py:956-958) with k = 1 / HW;
 *                  backward: the gate's gradient sum_rows dy * u.
 *   grl_se_apply : out[row][c] = a[row][c] * g[b][c] (+ f[row][c]) (+ k * h[b][c]).  Forward x + cab(x) * gate (:965-967 and the residual
 *                  of mixed_attn_block_efficient.py:548); backward d_u = dy * gate + d_pool / HW. */
typedef struct GrlSeRowsArgs {
    const float* a; int64_t lda;
    const float* f; int64_t ldf;     /* optional */
    const float* g;                  /* apply: [images, C] */
    const float* h;                  /* apply, optional: [images, C] */
    float* out;     int64_t ldo;     /* colsum: [images, C] (ldo ignored); apply: [M, ldo] */
    float k;
    int32_t M, C, rows_per_image;
} GrlSeRowsArgs;

int grl_se_colsum(void* stream, const GrlSeRowsArgs* args);
int grl_se_apply(void* stream, const GrlSeRowsArgs* args);

/* Relative-position bias tables for MANY AffineTransforms at once, forward and backward (ABI 21, training path):
 *   replaces  16 * sigmoid(cpb_mlp(relative_coords_table))  models/common/mixed_attn_block_efficient.py:23-34,49-58  (cpb_mlp =
 *   Linear(2, 512, bias) -> ReLU -> Linear(512, nh, no bias)) and autograd through it, without the [G, rows, 512] hidden layer in
 *   memory (1.5 GB for the 80 stripe transforms of GRL-Base at the checkpoint geometry).
 * out[g][n][i] = 16 log2(e) sigmoid(pre[g][n][rows-1-i]) for i < rows -- the REVERSED, exp2-domain layout grl_attention_fwd reads --
 * and the value of source row 0 in the pad entries i = rows .. rows4-1.  Backward: d_w1 / d_b1 / d_w2 must be zeroed by the caller
 * (accumulated with atomics).  nh in {1, 2, 3, 4, 6, 8}, hidden = 512. */
typedef struct GrlCpbArgs {
    const float* coords;    /* [rows, 2]   relative coordinates table (tables.coords_table), shared by all transforms */
    const float* w1;        /* [G, hidden, 2]   cpb_mlp.0.weight  */
    const float* b1;        /* [G, hidden]      cpb_mlp.0.bias    */
    const float* w2;        /* [G, nh, hidden]  cpb_mlp.2.weight  */
    float* out;             /* [G, nh, rows4]   (forward)         */
    const float* d_out;     /* [G, nh, rows4]   (backward)        */
    float* d_w1;            /* [G, hidden, 2], d_b1 [G, hidden], d_w2 [G, nh, hidden]  (backward; zeroed by the caller) */
    float* d_b1;
    float* d_w2;
    int32_t G, rows, rows4, nh, hidden;
} GrlCpbArgs;

int grl_cpb_table_fwd(void* stream, const GrlCpbArgs* args);
int grl_cpb_table_bwd(void* stream, const GrlCpbArgs* args);

/* Debug aid (ABI 21; no reference counterpart): fills the LDS of every CU with 0xFF bytes (fp32 / fp16 NaN) by a launch on
 * `stream`.  LDS keeps what the previous workgroup left in it; a kernel that reads LDS it has not written is otherwise right or
 * wrong depending on what ran before it.  The Python wrappers call this before every launch when GRL_DIRTY_LDS=1 (tests). */
int grl_debug_dirty_lds(void* stream);

/* Library self-description (used by the loader to refuse a stale build). */
int grl_abi_version(void);
const char* grl_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* GRL_HIP_H */

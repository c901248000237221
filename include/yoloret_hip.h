/* yoloret_hip.h - C-ABI of libyoloret_hip.so: the MI355X (gfx950) detection-forward
 * runtime behind yoloret_amd's Python surface.
 *
 * The reference (prakharg24/yoloret) has NO native / FFI interface: its path is
 * Python calling TensorFlow kernels.  Each entry point below therefore cites
 * the reference Python function (file:line under /root/reference/code) whose TF
 * kernels it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all tensors NHWC.  Element type (yr_dtype): float32, or - activations between the ops of a
 *     reduced-precision plan (BASELINE.json configs 3 and 5) - bfloat16 / float16.  Images in, the
 *     three logit outputs, BatchNorm scale/shift, depthwise weights, SE vectors and everything
 *     behind the logits (decode, NMS) are float32 in every plan; all accumulation is float32.
 *   - `ld` = channel stride in ELEMENTS (>= channels; a multiple of 4 for float32 tensors and of 8
 *     for 16-bit tensors unless stated, i.e. every pixel row is 16-byte aligned); every device
 *     pointer 16-byte aligned.
 *   - every function returns 0 on success, a negative yr_status otherwise, and
 *     never throws; the message is available from yr_last_error().
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it
 *     asynchronously, nothing synchronises.
 *   - a handle is bound to the device that was current at yr_create and is not
 *     thread-safe.
 */
#ifndef YOLORET_HIP_H
#define YOLORET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YR_ABI_VERSION 9   /* 9: float32 POINTWISE se_reduced bit 18 = the pixel-stationary form (pointwise_stream.hip: `wgt` holds the weights' float16 planes in fragment order) and bit 19 = its two-output form (gate_out / se_hidden / reserved0 = a second conv of the same source); plans of ABI 8 run unchanged - only the version word of a serialised plan differs; 8: YR_OP_MBR k bit 6 = the weight-streaming form of the fused block (mbk.hip; plans of ABI 7 run unchanged - only the version word of a serialised plan differs); 7: squeeze-excite finished by its producer (yr_op.gate_out / se_w / se_hidden / sync: the "SE tail"), op kind HEAD (gathered 1x1 conv -> depthwise 3x3 -> SE of a detection-head block in one launch), yr_workspace_bytes includes the arrival counters; 6: split forms - k bit 7 of MBR / MBE (float16-plane fragments), se_reduced bit 16 of a float32 POINTWISE op (= keep the float32 MFMA); float32 POINTWISE ops at least 16 channels deep otherwise run on the 16-bit matrix pipe with two float16 planes per operand (pointwise_split.hip; same results to float32 rounding, |x|, |w| < 65504); 5: op kinds MBR / MBE (float32 blocks on the matrix pipe, register-chained), MBCONV removed; 4: YR_U8 images (uint8 network entry); 3: op kind MBX, pair-packed STEM weights, BN scale folded into the packed taps of STEMBLOCK / MBLANE */
#define YR_MAX_SRC 4

typedef enum {
    YR_OK = 0,
    YR_ERR_ARG = -1,     /* bad argument / unsupported shape */
    YR_ERR_HIP = -2,     /* a HIP runtime call failed */
    YR_ERR_STATE = -3    /* e.g. weights not loaded */
} yr_status;

/* Element type of an activation tensor / of the POINTWISE weight matrix.  The 16-bit types are storage formats:
 * every kernel converts to float32 on load (bf16: exact widening) and rounds to nearest-even on store; only the
 * POINTWISE GEMM consumes them natively (v_mfma_f32_16x16x32_bf16 / _f16, float32 accumulate). */
typedef enum { YR_F32 = 0, YR_BF16 = 1, YR_F16 = 2,
               YR_U8 = 3   /* ABI 4: uint8 - ONLY as the element type of the image (external slot 0) read by a STEM / STEMBLOCK op: the
                              decoded bytes as they are, x / 255 (code/yolo.py:106) applied inside the kernel; the float32 batch
                              (4x the bytes) is then never written or read */
} yr_dtype;

typedef enum { YR_ACT_NONE = 0, YR_ACT_RELU6 = 1, YR_ACT_SWISH = 2, YR_ACT_SIGMOID = 3, YR_ACT_LEAKY = 4 } yr_act;

/* How a source tensor is read by a consumer (folds Keras UpSampling2D /
 * MaxPooling2D / Concatenate - model.py:139-144,157,164-166,253-255,307-308 -
 * into the consumer's loads). */
typedef enum {
    YR_X_IDENTITY = 0, YR_X_UP2 = 1, YR_X_MAXPOOL2 = 2, YR_X_MAXPOOL4 = 3,
    /* POINTWISE only, last source only: NOT part of the k space.  Its `cout` channels, gathered with nearest 2x
     * upsampling, are added to the accumulator before BatchNorm.  A 1x1 convolution commutes with nearest
     * upsampling (W.[up2(a); b] = up2(Wa.a) + Wb.b), so the compiler computes the upsampled share of a concat conv
     * at the source resolution and hands it in this way (model.py:253-255,273-275: UpSampling2D + Concatenate feeding
     * the first 1x1 conv of the top-down heads). */
    YR_X_UP2_ADD = 4,
    /* POINTWISE only, the single source: the source is read THROUGH DepthwiseConv2D 3x3 (TF 'SAME') + BatchNorm +
     * activation, i.e. the conv's input channel k at an output pixel is act(BN(sum over the 3x3 taps of
     * dw_w[tap][k] * src[pixel*stride + tap][k])) - the depthwise stage of an inverted-residual block folded into
     * the loads of its projection (MobileNetV2 block_* [3P] depthwise + project; efficientnet.py:501-533).  The
     * depthwise result (the block's widest tensor besides the expand output) never reaches HBM; values are
     * bit-identical to a DEPTHWISE op followed by the same POINTWISE op.  Op fields: src[0] = the depthwise INPUT
     * (its h, w), op.h/op.w = the depthwise OUTPUT dims, se_reduced = stride | (yr_act of the depthwise stage << 8),
     * wgt2 = depthwise weights [9][ld], b1 / b2 = its folded BN scale / shift [ld], ld = round_up(src.c, 4). */
    YR_X_DW3 = 5
} yr_xform;

/* One concatenated input segment.  (h,w) are the SOURCE's spatial dims; the
 * consumer's dims follow from xform.  In plan ops `ptr` is unused and `buf`
 * indexes the plan's buffer table; in yr_op_* calls `ptr` is the device pointer. */
typedef struct {
    const void* ptr;
    int32_t buf;
    int32_t h, w;
    int32_t c;       /* channels taken from this source */
    int32_t ld;      /* channel stride of the source buffer, in elements */
    int32_t xform;   /* yr_xform */
    int32_t dtype;   /* yr_dtype of the source buffer: must equal the op's `dtype`, except for sources that are float32
                        in every plan (the image of STEM / STEMBLOCK, a YR_X_UP2_ADD source, the pooled vector of SE_FC) */
} yr_src;

typedef enum {
    YR_OP_STEM = 1,      /* Conv2D 3x3 s2 Cin=3 + BN + act            (MobileNetV2 Conv1 [3P]; efficientnet.py:636-645).  Optional
                            wgt2 = the same parameters packed per channel PAIR, [round_up(cout,4)/2][27 taps x 2, times the BN
                            scale | 1 1 | BN shift 2] (STEMBLOCK's stem layout): selects the scalar-operand kernel */
    YR_OP_POINTWISE = 2, /* Conv2D 1x1 (+bias)(+BN)(+act)(+residual)  (model.py:25-30,98-114,152-155,243-251; efficientnet.py:485-496,517-533)
                            se_reduced flag bits (no depthwise-folded source): float32 ops, bit 16 = keep the float32 MFMA (not the
                            float16-plane split form); bit 17 (round 5, no layout change) = the k-split form of the split kernel - a
                            workgroup is one 16 x 16 output tile, its four waves split the k range (pointwise_split.hip: pwk_kernel; 16-bit ops:
                            one 16 x 32 tile, pointwise_h.hip: pwkh_kernel);
                            what the compiler's plan for one or two images asks of maps up to 32 x 32.  The form groups the sums by wave: it
                            belongs to the plan, not to the tuner (yr_op.k is not looked at); ignored below 64 input channels.
                            bit 18 (ABI 9, float32 ops without bit 16 / 17; identity | up2 sources, no residual / up2_add, act none | ReLU6,
                            k space <= 384 channels): the PIXEL-STATIONARY form (pointwise_stream.hip) - `wgt` is NOT Wt[cout][kp] but its float16
                            planes in MFMA fragment order, [ceil(cout / 16)][NK][2 planes h | m][64 lanes][8 halves] as float32 words with
                            NK = yr_pwt_chunks(kp) chunks of 32 channels (lane (m, g) of tile t, chunk c: W[16 t + m][32 c + 8 g + i], zero
                            beyond cout / kp; h = f16(w), m = f16((w - h) 2^11): compiler.head_pack).  Results are bit-identical to the tiled
                            split kernel's.  bit 19 (with bit 18): TWO outputs - a second 1x1 conv of the same (gated) single identity source in
                            the same launch: gate_out / gate_out_buf / gate_out_ld = its output (float32), se_hidden = its couts,
                            reserved0 = its yr_act | pooled << 8 (MaxPooling2D(2) of its result, like stride = 2 for the first); its cout tiles
                            follow the first output's in `wgt`, and scale / shift are [16 (tiles of the first + tiles of the second)]
                            floats, each output's values padded to a multiple of 16. */
    YR_OP_DEPTHWISE = 3, /* DepthwiseConv2D k3/k5 s1/s2 SAME + BN + act (model.py:20-24; efficientnet.py:501-510).  With `gate` set (SE
                            form): the kernel ALSO writes per-workgroup channel sums of its output to
                            gate = float32 [B][se_reduced rows][gate_ld] (rows: compiler.dw_se_geometry == depthwise.hip) - the squeeze of squeeze-excite (efficientnet.py:417) as an
                            epilogue; an SE_FC op with k = H*W adds the rows up instead of re-reading the map */
    YR_OP_SE_MEAN = 4,   /* Mean over H,W                              (efficientnet.py:391-403,417) */
    YR_OP_SE_FC = 5,     /* 1x1+bias -> Swish -> 1x1+bias -> sigmoid   (efficientnet.py:419-434).  Source: the pooled vector (h*w == 1,
                            float32), or a map to pool first (SE_MEAN merged in), or - k > 0 - float32 rows of partial channel
                            sums [B][h*w][ld] to add up and divide by k (the depthwise SE form above) */
    YR_OP_WSUM = 6,      /* WeightedSum of 4 gathered sources          (model.py:117-137,157) */
    YR_OP_GATHER = 7,    /* materialise upsample/maxpool/concat        (standalone K5; testing / unfused use) */
    YR_OP_MBCONV = 8,    /* (removed in ABI 5: the float32 LDS-tiled block kernel; no shipped plan selected it since MBLANE, and MBR / MBE
                            now take the blocks it was the fallback for.  The number stays reserved; yr_op_run returns YR_ERR_ARG) */
    YR_OP_STEMBLOCK = 9, /* fused network entry: stem Conv2D 3x3 s2 (Cin=3)+BN+act -> DW3x3 s1+BN+act -> project 1x1+BN
                            (MobileNetV2 Conv1 + expanded_conv block [3P]); se_reduced = stem width C1.  Parameters are packed
                            per channel PAIR (CP = round_up(C1,4)/2, COP = round_up(cout,8), zero padded; scale/shift unused):
                            wgt  = stem      [CP][27 taps (ky,kx,ci) x 2, times the BN scale | 1 1 | BN shift 2],
                            wgt2 = depthwise [CP][ 9 taps (ky,kx)    x 2, times the BN scale | 1 1 | BN shift 2],
                            b1   = project W[2*CP][COP] (input-channel major), b2 = project BN scale[COP] ++ shift[COP].
                            Built for (CP,COP) in {(12,16),(16,16),(16,24),(20,24),(24,16),(24,24)}; others: YR_ERR_ARG.
                            Without b1 (no projection; cout == C1, CP in {16,20,24}): stem + depthwise only - the depthwise
                            map is stored and, if `gate` is set, gate = OUTPUT float32 [B][ceil(h/14)*ceil(w/14)][gate_ld]
                            per-tile channel sums of the stored values (the squeeze of the first SE block).
                            MATRIX-PIPE LAYOUT (16-bit plans, round 3; selected by scale != NULL; C1 <= 64, cout <= 32, even image
                            sizes; C1P = round_up(C1,32), COP = round_up(cout,16), zero padded; stemblock_h.hip):
                            wgt = stem kernel as dtype [C1P][32], k order: image rows 0..2 x values 0..7 of the row's 9 (kx,ci) |
                            value 8 of rows 0..2 | 0 x 5;  scale / shift = stem BN [C1P];  wgt2 = float32 [10][C1P]: nine depthwise
                            taps times the BN scale | BN shift;  b1 = project Wt[COP][C1P] as dtype;  b2 = project BN [2][COP] */
    YR_OP_MBH = 11,      /* the MBCONV block on 16-bit activations, both 1x1 convs on bf16 / f16 MFMA, depthwise K = 3 | 5 from an
                            LDS tile (dtype must be YR_BF16 / YR_F16; cin, cout <= 128).  se_reduced = Cexp; k = K, optionally
                            | th << 8 | tw << 16 to force the output tile.  CexpP = round_up(Cexp,32), KP = round_up(cin,32),
                            zero padded: wgt = expand Wt[CexpP][KP] (16-bit); wgt2 = [K*K + 4][CexpP] float32: depthwise taps |
                            depthwise BN scale | shift | expand BN scale | shift;
                            b1 = project Wt[cout][CexpP] (16-bit); b2 = project BN scale ++ shift, [round_up(cout,8)] each;
                            res (optional) = the block input (stride 1, cin == cout).  The library runs K = 3, stride 2, cin <= 32
                            (cin % 8 == 0), Cexp <= 192, cout <= 32, no residual, no forced tile on mbn_h.hip (whole expanded
                            halo tile on chip in the 16-bit type: same op fields) */
    YR_OP_MBLANE = 10,   /* the MBCONV block (same layers, same op fields) in the lane-per-pixel formulation for narrow
                            block inputs (Cin <= 32): packed-fp32 FMA with scalar-register weights instead of MFMA.
                            se_reduced = Cexp; parameters packed per expanded-channel PAIR, P = round_up(ceil(Cexp/2),8),
                            CINP = round_up(Cin,4), COP = round_up(cout,8), zero padded; scale/shift unused:
                            wgt  = expand    [P][CINP x 2 (input-channel major), times the BN scale | 1 1 | BN shift 2],
                            wgt2 = depthwise [P][9 taps (ky,kx) x 2, times the BN scale | 1 1 | BN shift 2],
                            b1   = project W[2P][COP] (expanded-channel major), b2 = project BN scale[COP] ++ shift[COP].
                            Built for (CINP/4,COP) in {(4,16),(4,24),(6,24),(6,32),(6,40),(6,48),(8,32),(8,40),(8,48)}.
                            wgt = NULL: a block WITHOUT expand conv (expand ratio 1, efficientnet.py:467 skipped; stride 1,
                            se_reduced = Cin): depthwise + project (+ residual) on the block input itself; built for
                            (4,16), (6,24), (8,32) */
    YR_OP_MBR = 13,      /* the MBCONV block (same layers) in float32 on the fp32 matrix pipe, ROW-WALKING and REGISTER-CHAINED
                            (mbr.hip): a wave walks a strip of 16 input columns row by row; expand GEMM (v_mfma_f32_16x16x4_f32, the
                            strip row's pixels as N, block input straight from global memory in operand layout) -> the MFMA result
                            registers ARE the depthwise conv's input (horizontal taps by DPP, vertical taps = the last three rows kept
                            in registers) -> the depthwise result registers ARE the projection MFMAs' B operand.  The expanded tensor
                            never leaves the register file; the waves of a workgroup split the expanded channels and add their partial
                            projections through LDS.  dtype = YR_F32; act = YR_ACT_RELU6; cin % 16 in {0, 8}; se_reduced = Cexp, a
                            multiple of 16; cout % 4 == 0; k = 3 | nw << 8 | segs << 16 (nw: waves per workgroup, segs: row segments
                            per strip; 0 = the library's choice); res (optional) = the block input.  T = Cexp / 16 expanded tiles,
                            TO = ceil(cout / 16), KE = cin / 4 expand steps; scale / shift / b1 unused:
                            wgt  = MFMA A fragments [T][KE + 4 TO][64 lanes]; register rho of lane l (m = l % 16, g = l / 16) of tile j:
                                   rho < KE: We[16 j + m][kperm(rho, g)] x expand BN scale, kperm(4 c + s, g) = 16 c + 4 g + s for the full
                                   16-channel chunks of cin and 16 n + 2 g + s (s < 2) for a trailing 8;
                                   rho = KE + 4 t + s: Wp[16 t + m][16 j + 4 g + s] x project BN scale (0 for 16 t + m >= cout);
                            wgt2 = [T][11][16]: the nine depthwise taps (ky, kx) x depthwise BN scale | depthwise BN shift | expand BN shift;
                            b2   = project BN shift [16 TO].
                            k bit 7 (0x80) = the SPLIT form (ABI 6): both 1x1 convolutions on the 16-bit matrix pipe with float32-grade operands -
                            every float32 value as two float16 planes, x = h + 2^-11 m, three MFMAs per product (mbr.hip "SPLIT form") -; then
                            wgt  = the float32 words holding [T][ceil(cin / 32)][2 planes][64 lanes][8 halves] (lane (m, g), step c:
                                   We[16 j + m][32 c + 8 g + i] x BN scale, zero beyond cin) followed by, per expanded-tile pair (tA, tB) of the nw
                                   waves in order (a wave pairs ITS tiles, an odd last one with nothing), [TO][2 planes][64][8]:
                                   Wp[16 t + m][16 tA + 4 g + i] (i < 4) | Wp[16 t + m][16 tB + 4 g + i - 4]; nw must be what the fragments were
                                   packed for (yoloret_amd.compiler.mbs_pack).  Precondition: |block input| < 65504 (undefined beyond: NaN or a ReLU6-clamped value).
                            k bits 6 AND 7 (0xc0) = the WEIGHT-STREAMING form (ABI 8; mbk.hip): the same block in ONE launch where the fragments of both
                            convolutions do not fit a CU's register file (MobileNetV2 x0.75 block_7..15, reference [3P] via code/yolo3/override.py:290-341).
                            The pixels are stationary - a wave owns `rows` (1 | 2) input rows of a 16-column strip and the projection accumulators of
                            its output rows for the whole kernel - and the weights stream through LDS one pair of expanded tiles at a time (LDS-direct
                            buffer loads, three chunk buffers); the neighbour rows of the vertical taps are exchanged between waves through LDS, one
                            barrier per pair.  k = 3 | 0xc0 | nw << 8 | rows << 16 (both fixed by the plan: nothing is tuned, the sums are grouped by
                            the shape alone); wgt2 unused;
                            wgt  = ceil(T / 2) chunks of (4 ceil(cin / 32) + 2 TO) KB + 2 KB: chunk q = pair (2 q, 2 q + 1) =
                                   [2 tiles][ceil(cin / 32)][2 planes][64 lanes][8 halves] expand fragments (as above) |
                                   [TO][2 planes][64][8] project fragments of the pair (Wp[16 t + m][16 (2 q) + 4 g + i] (i < 4) | [16 (2 q + 1) + 4 g + i - 4]) |
                                   [2 tiles][11][16] float32: taps x depthwise BN scale | depthwise BN shift | expand BN shift | zeros up to 2 KB
                                   (an odd T: the second tile of the last pair is all zeros) - yoloret_amd.compiler.mbk_pack;
                            b2   = project BN shift [16 TO].  Same precondition as the split form.  Built for (cin, Cexp, cout, stride, rows, nw) in
                            mbk.hip's MBK_CASE list.
                            Built for the MobileNetV2 x0.75 / x1.4 blocks (mbr.hip: MBR_CASE / MBS_CASE lists); other shapes: YR_ERR_ARG */
    YR_OP_MBE = 14,      /* the first two thirds of the MBCONV block in float32 - expand 1x1 + BN + ReLU6 -> depthwise 3x3 (stride 1 | 2) +
                            BN + ReLU6 - in YR_OP_MBR's register-chained form (mbr.hip: mbe_kernel), for blocks whose weights do not fit
                            one CU's register file (MobileNetV2 x0.75 block_11 on): every wave walks a strip segment for a few expanded
                            tiles and stores the depthwise map; the expand output never exists, the projection stays a POINTWISE op.
                            src[0] = block input (cin % 16 in {0, 8}); cout = Cexp (multiple of 16); k = 3 | segs << 16; act = RELU6;
                            wgt = expand A fragments [T][KE][64] (YR_OP_MBR's register order, rho < KE); wgt2 = [T][11][16] as YR_OP_MBR.
                            k bit 7 = the split form (see YR_OP_MBR): wgt = the float32 words holding [T][ceil(cin / 32)][2 planes][64][8 halves].
                            Built for cin in {48, 72, 88, 120, 136, 224} (the split form: all but 224) */
    YR_OP_HEAD = 15,     /* ABI 7: the first two thirds of a detection-head block (model.py:91-115: Conv2D 1x1 + BN + ReLU6 -> MBConvBlock's
                            depthwise 3x3 + BN + Swish [-> SE]) in one launch, float32 (headblock.hip): the 1x1 convolution as YR_OP_POINTWISE's
                            split form (same sources incl. up-sampling / pooling / concat gathers, a YR_X_UP2_ADD last source, an SE `gate` on a
                            single identity source; operands as two float16 planes, |x|, |w| < 65504) over a REGION of one image with its
                            one-pixel halo, the F-wide conv output kept in LDS, the depthwise conv (stride 1, TF SAME) computed from there,
                            stored, and - with the SE-tail fields below - summed per channel.  The F-wide conv output never reaches HBM.
                            h, w = the map (conv and depthwise output alike); cin = sum of src[].c; cout = F; act = the depthwise activation;
                            k = 3 | conv activation << 8; stride = 1; wgt = Wt[F][kp], scale / shift [F] (conv BN) as POINTWISE;
                            k bit 7 (0x80): wgt = the float32 words holding float16 planes in fragment order instead,
                            [ceil(F / 16)][NK][2 planes][64 lanes][8 halves] with the k space cut into chunks of 32 channels PER SOURCE (NK =
                            sum of ceil(src.c / 32); lane (m = l % 16, g = l / 16) of cout tile t, chunk j of source s: W[16 t + m][channel
                            32 j + 8 g + i of s], zero beyond the source / beyond F; h plane, then m = f16((w - h) 2^11)) - the LDS-direct
                            kernel (identity / up-sampled sources only; yoloret_amd.compiler.head_pack);
                            wgt2 = float32 [10][round_up(F, 4)]: the nine depthwise taps (ky, kx) x depthwise BN scale | depthwise BN shift.
                            k bit 6 (0x40): the WALKING form (headwalk.hip: a wave walks a strip of 16 columns with the weights of its cout tiles
                            in registers - YR_OP_MBE's scheme; identity sources only, at most 7 chunks of 32 channels, F % 16 == 0, conv
                            activation ReLU6 / none): wgt = the planes of bit 7 WITH the conv's BN scale folded in, scale = that BN scale [F],
                            shift unused, wgt2 = [F / 16][11][16]: depthwise taps x BN scale | depthwise BN shift | conv BN shift (YR_OP_MBR's
                            table); se_reduced = yr_head_walk_rows(h, w).
                            k bits 5 AND 6 (0x60; ABI 8): the WEIGHT-STREAMING form (headstream.hip; float32 plans): the pixels of a wave's one or two
                            rows of a 16-column strip stay in registers over the whole k space (gathered once: identity or 2 x 2 max-pooled
                            sources, the single source's SE gate), the conv's output channels stream past them in pairs of tiles through LDS.
                            Parameters exactly as bit 6 alone; one to three k-space sources (YR_X_IDENTITY | YR_X_MAXPOOL2) + an optional
                            YR_X_UP2_ADD last; at most 11 chunks of 32 channels on maps below 20 rows, 7 otherwise; F % 32 == 0; no SE tail;
                            se_reduced = yr_head_stream_rows(h, w) (one row of sums per strip, row segment and wave).
                            16-BIT PLANS (dtype = out_dtype = bf16 | f16; headwalk_h.hip): the walking form only (k bit 6) - identity sources
                            of the op's type (ld % 8 == 0), optionally a float32 YR_X_UP2_ADD last source, at most 8 chunks of 32 channels,
                            F % 128 == 0; wgt = the conv's 16-bit weights in fragment order [F / 16][NK][64 lanes][8] WITHOUT the BN scale
                            (yoloret_amd.compiler.head_pack16), scale = conv BN scale [F] float32, wgt2 as above; float32 from the
                            accumulator on, ONE rounding at the store (the F-wide conv output is never rounded to 16 bits), the sums are
                            those of the stored values; an SE gate (res) is folded into the stationary weights (w * g rounded once);
                            no SE tail (gate_out must be null).
                            res / res_ld (optional) = the float32 SE gate vector [B][res_ld] multiplied onto the single identity source on load
                            (`gate` is taken by the sums this op writes); k bits 16-23 (optional) = cout tiles of 16 per workgroup.
                            SE tail: gate = OUTPUT float32 [B][se_reduced][gate_ld] channel sums, one row per region (se_reduced = regions per
                            image: yr_head_regions()) */
    YR_OP_MBX = 12       /* the first two thirds of an MBConv block WITH squeeze-excite (efficientnet.py:406-536), 16-bit
                            activations: expand 1x1 + BN + act (bf16 / f16 MFMA) -> depthwise K = 3 | 5, stride 1 | 2 + BN +
                            act, the expanded input of the depthwise conv staying in LDS; the depthwise map is stored and
                            the squeeze (tf.reduce_mean over H, W, efficientnet.py:417) leaves as per-tile channel sums.
                            src[0] = block input (cin <= 128); cout = Cexp = width of the stored map; wgt, wgt2, k as MBH;
                            gate (optional) = OUTPUT float32 [B][se_reduced][gate_ld] sums of the STORED values, one row
                            per output tile, unused rows zeroed (se_reduced = rows >= tiles per image: the buffer is sized
                            for the smallest tile the host may pick, 4 x 8, or 4 x 4 on maps under 1000 pixels); SE_FC (k = H*W) adds the rows up; the projection stays a POINTWISE op */
} yr_op_kind;

/* One fused operation.  Weight-like fields are float offsets into the weight
 * blob in plan ops (`wgt_off` etc.), or device pointers in yr_op_* calls. */
typedef struct {
    int32_t kind;         /* yr_op_kind */
    int32_t act;          /* yr_act */
    int32_t h, w;         /* OUTPUT spatial dims per image */
    int32_t cin, cout;    /* logical channels (cin = sum of src[].c) */
    int32_t k, stride;    /* DEPTHWISE / STEM kernel size and stride */
    int32_t nsrc;
    int32_t se_reduced;   /* SE_FC: hidden width */
    int32_t dtype;        /* yr_dtype the op works in: element type of its sources, residual and (POINTWISE) `wgt`;
                             16-bit types: every op kind except MBCONV */
    int32_t out_dtype;    /* yr_dtype of `out`: equal to `dtype`, or YR_F32 from a 16-bit POINTWISE op (logit outputs,
                             the low-resolution partial sums of a hoisted conv); SE_MEAN / SE_FC always write float32 */
    yr_src src[YR_MAX_SRC];
    /* output */
    void* out;  int32_t out_buf;  int32_t out_ld;
    /* optional residual added after BN (same shape as output, element type `dtype`) */
    const void* res;  int32_t res_buf;  int32_t res_ld;
    /* optional SE gate [B, gate_ld] (float32) multiplied onto the (single) source on load */
    const float* gate;  int32_t gate_buf;  int32_t gate_ld;
    /* parameters (device pointers, or float offsets into the blob when in a plan); float32 unless stated:
     *   POINTWISE: wgt = Wt[cout][kp] of element type `dtype` (kp = sum of round_up(src.c, V), V = 4 for float32 and 8
     *              for the 16-bit types, zero padded; in the blob a 16-bit matrix occupies cout*kp/2 floats),
     *              scale/shift [cout] (folded BN and/or bias)
     *   DEPTHWISE: wgt = [k*k][round_up(c,V)] ; scale/shift [round_up(c,V)]  (V as above: 4 for float32 ops, 8 for 16-bit ops)
     *   STEM:      wgt = [27][round_up(cout,4)] ; scale/shift [round_up(cout,4)]
     *   SE_FC:     wgt = W1[ldc][R4] (ABI 7: the Keras kernel [1,1,C,R] as it is, rows padded to R4 = round_up(reduced, 4)), b1 [R4],
 *              wgt2 = W2[reduced][ldc], b2 [ldc], ldc = round_up(c,4); all 16-byte aligned
     *   WSUM:      wgt = alpha[4]
     *   (the fused block ops document their packed layouts at their yr_op_kind above) */
    const float* wgt;    int64_t wgt_off;
    const float* scale;  int64_t scale_off;
    const float* shift;  int64_t shift_off;
    const float* wgt2;   int64_t wgt2_off;
    const float* b1;     int64_t b1_off;
    const float* b2;     int64_t b2_off;
    /* ---- ABI 7: the "SE tail" - squeeze-excite finished by the op that produces the map (csrc/se_tail.h).  An op that writes
     * per-workgroup channel sums to `gate` (DEPTHWISE SE form, MBX, STEMBLOCK without projection, HEAD) and has gate_out set also
     * runs the SE block's FC pair (efficientnet.py:419-434): the workgroup that completes an image's `se_reduced` rows adds them up
     * in index order, divides by h * w and writes sigmoid(W2 . swish(W1 . mean + b1) + b2) to gate_out[b] - what an SE_FC op with
     * k = h * w reading `gate` would have written (same values to float32 rounding), without the launch. */
    float* gate_out;  int32_t gate_out_buf;  int32_t gate_out_ld;   /* float32 [B][gate_out_ld], gate_out_ld >= round_up(cout, 4) */
    int32_t se_hidden;    /* hidden width R of the FC pair */
    int32_t reserved0;    /* a two-output POINTWISE op (se_reduced bit 19): the second output's yr_act | pooled << 8; else 0 */
    const float* se_w;  int64_t se_w_off;   /* W1 [ldc][R4] (Keras kernel [1,1,C,R], rows padded to R4 = round_up(R, 4)) | W2 [R][ldc] | b1 [R4] | b2 [ldc], ldc = round_up(cout, 4) */
    uint32_t* sync;       /* [batch] arrival counters: zero before the launch, zero again after it.  Plan ops: assigned by yr_forward
                             from the tail of the workspace (cleared at the start of every pass); yr_op_run: the caller's */
} yr_op;

/* Buffer table entry of a plan: per-image size in BYTES and either an arena
 * offset (per-image bytes, a multiple of 16; the runtime multiplies by batch) or an
 * external slot (0 = input images, 1..3 = y1..y3; always float32). */
typedef struct {
    int64_t bytes_per_image;
    int64_t arena_off_per_image;   /* -1 for external buffers */
    int32_t external_slot;         /* -1 for arena buffers */
    int32_t dtype;                 /* yr_dtype of the buffer's elements */
} yr_buf;

typedef struct yr_handle yr_handle;

const char* yr_last_error(void);
int yr_abi_version(void);
/* sizeof(yr_src) / sizeof(yr_op) / sizeof(yr_buf) for which = 0 / 1 / 2 (else 0): lets a binding that restates the
 * structs (ctypes, cgo, JNA ...) verify its layout against the library's at load time. */
int yr_abi_sizeof(int which);

/* ---- whole-graph runtime: replaces tf.keras.Model.__call__ on the graph built by
 * yolov3_body (model.py:170-342; called at yolo.py:152, map.py:111). */
int yr_create(const yr_op* ops, int n_ops, const yr_buf* bufs, int n_bufs, yr_handle** out);
void yr_destroy(yr_handle* h);
/* ---- the same from ONE self-contained byte blob (yoloret_amd: Model.save_plan(); layout below): op list, buffer
 * table, parameter blob and the autotuned tile tables of a compiled model.  A host without the Python graph compiler
 * (C, C++, Go/cgo, JNI ...) instantiates "MobileNetV2x0.75 @416, bf16" from a file with this one call: the handle
 * comes back with its weights on the device and its tile tables installed.  Replaces yolov3_body + load_weights
 * (model.py:170-342, yolo.py:82-87) for deployments.  All fields little-endian:
 *   char magic[8] = "YRPLAN\0\0"; u32 abi (= YR_ABI_VERSION); u32 n_ops; u32 n_bufs; u32 sizeof(yr_op);
 *   u32 sizeof(yr_buf); u32 n_tables; u64 n_weight_floats; i32 in_h, in_w; i32 out_hwc[3][3] (y1, y2, y3: h, w, c);
 *   i32 reserved[3];   -- 96 bytes
 *   yr_op ops[n_ops] (pointer fields zero); yr_buf bufs[n_bufs]; float weights[n_weight_floats];
 *   n_tables x { i32 batch; i32 cfg[n_ops]; }   (yr_set_tuning tables) */
int yr_create_from_blob(const void* blob, size_t bytes, yr_handle** out);
/* Input / output geometry of a handle made by yr_create_from_blob (YR_ERR_STATE for other handles):
 * in_hw[2] = network input (h, w); out_hwc[9] = (h, w, c) of y1, y2, y3 (c = A*(C+5), dense). */
int yr_plan_io_dims(const yr_handle* h, int32_t* in_hw, int32_t* out_hwc);
/* Copies the flat fp32 parameter blob (host) to the device; owned by the handle.
 * Replaces tf.keras.Model.load_weights (yolo.py:87) once weights are in blob order. */
int yr_load_weights(yr_handle* h, const float* host_blob, size_t n_floats);
/* Bytes of caller-owned device workspace yr_forward needs for `batch` images (the arena of intermediate tensors followed by the
 * SE-tail arrival counters of the plan's ops, 4 * batch bytes each). */
size_t yr_workspace_bytes(const yr_handle* h, int batch);
/* images [B,H,W,3] -> y1,y2,y3 raw logits [B,G,G,A*(C+5)], G = H/32, H/16, H/8. */
int yr_forward(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
               void* workspace, size_t workspace_bytes, void* stream);
/* Measurement aid: the same replay with a hipEvent pair around every op, `iters` times.
 * ms_per_op [n_ops] gets each op's average duration; kernel_names [n_ops] (nullable) gets a static
 * string naming the kernel symbol the op dispatched to.  Synchronises the stream. */
int yr_forward_profile(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                       void* workspace, size_t workspace_bytes, void* stream, int iters,
                       float* ms_per_op, const char** kernel_names);
/* The same pass, instrumented: max_abs_per_op[n_ops] (host) receives the largest |value| among the float32 k-space sources each op
 * reads (+inf where a NaN was seen; 0 for ops without such sources).  A float32 plan's SPLIT-form ops (POINTWISE without se_reduced bit
 * 16, MBR / MBE with k bit 7, HEAD) carry their operands as two float16 planes and need |x| < 65504 - the reference's float32
 * convolutions have no such bound (model.py:20-30; MobileNetV2's linear bottlenecks and residual sums are unclamped, override.py:290-341).
 * yoloret_amd's Model runs this once per set of weights on its first batch and rebuilds the plan with the ops beyond 60000 on the
 * float32-MFMA forms (Model.check_ranges); a host without it does the same through this call.  Synchronises the stream. */
int yr_forward_ranges(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                      void* workspace, size_t workspace_bytes, void* stream, float* max_abs_per_op);
/* Per-op tile autotuning for `batch` images: times every pointwise tile shape on every pointwise op (and a list of
 * output tiles on every MBH / MBX op) and remembers the fastest for later yr_forward calls with the same batch (numerics do not depend on the
 * shape).  Runs the forward once first; synchronises the stream. */
int yr_autotune(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                void* workspace, size_t workspace_bytes, void* stream, int iters);
/* The autotuned table for `batch`: one int per plan op - POINTWISE: 0 = heuristic, else the 1-based tile shape;
 * MBH, MBX: 0 = heuristic, else the output tile as th << 8 | tw << 16; every other op kind: 0.  yr_get_tuning returns YR_ERR_STATE if that batch has not been tuned; yr_set_tuning installs a table
 * saved from an earlier run (tune once, deploy many: n must equal yr_plan_num_launches). */
int yr_get_tuning(const yr_handle* h, int batch, int32_t* cfg_per_op, int n);
int yr_set_tuning(yr_handle* h, int batch, const int32_t* cfg_per_op, int n);
/* Number of kernel launches one yr_forward enqueues. */
int yr_plan_num_launches(const yr_handle* h);

/* ---- single fused ops (device pointers inside `op`); parity-testable in isolation. */
int yr_op_run(const yr_op* op, int batch, void* stream);
/* How a YR_OP_HEAD launch cuts an h x w map into nsy x nsx regions (a function of the shape alone): the rows of the squeeze-excite
 * sums buffer such an op writes per image = nsy * nsx (its se_reduced). */
int yr_head_regions(int h, int w, int32_t* nsy, int32_t* nsx);
/* ... and of the WALKING form of YR_OP_HEAD (k bit 6): rows = strips of 14 columns x row segments (its se_reduced). */
int yr_head_walk_rows(int h, int w, int32_t* rows);
/* ... and of its WEIGHT-STREAMING form (k bits 5 and 6): rows = strips x row segments x waves per workgroup. */
/* chunks of 32 channels the pixel-stationary POINTWISE form (se_reduced bit 18) runs a k space of kp channels with - what the plan
 * pads the weight planes to (1 .. 10, 12, 16); 0: the form does not take a k space this deep. */
int yr_pwt_chunks(int kp);
int yr_head_stream_rows(int h, int w, int32_t* rows);

/* ---- preprocessing (the step before the path): decoded uint8 [ih,iw,3] image (device) ->
 * letterboxed float32 [H,W,3] network input.  Replaces tf.io.decode_image(dtype=float32)'s /255
 * (yolo.py:106) + letterbox_image (utils.py:67-83: bilinear half-pixel resize, zero pad). */
int yr_letterbox(const unsigned char* src_u8, int ih, int iw, float* dst, int H, int W, void* stream);
/* The same for a batch of EQUALLY sized decoded images [B,ih,iw,3] -> [B,H,W,3] in one launch: the ingestion path of a
 * resident batch (uint8 over PCIe is a quarter of the float32 bytes; SURVEY.md 8(d) "incl. H2D"). */
int yr_letterbox_batch(const unsigned char* src_u8, int batch, int ih, int iw, float* dst, int H, int W, void* stream);

/* ---- decode: replaces yolo_head + yolo_correct_boxes + yolo_boxes_and_scores
 * (model.py:344-428) for the three scales at once, per image.
 *   y[s]      [B,G_s,G_s,A*(C+5)] raw logits, s = 0,1,2 for strides 32,16,8
 *   anchors   host, 9x(w,h) in the order of model_data/yolo_anchors.txt; scale s uses
 *             anchor_mask [[6,7,8],[3,4,5],[0,1,2]][s] (model.py:444)
 *   image_hw  device int32 [B,2] original image (h,w) per image
 *   boxes     [B,N,4] (ymin,xmin,ymax,xmax) fp32, N = A*sum(G_s^2)
 *   scores    [B,C,N] fp32, CLASS-MAJOR (score = conf*prob, model.py:426) */
int yr_decode(const float* y1, const float* y2, const float* y3, int batch, int in_h, int in_w,
              int num_anchors, int num_classes, int num_scales, const float* anchors_host,
              const int32_t* image_hw, float* boxes, float* scores, void* stream);
/* The same with the zoom-in test-time-augmentation pass of yolo_boxes_and_scores (model.py:408-417, enabled by
 * YoloModel.call(zoom_in=True), yolo.py:154-159): z[s] are the logits of the network run on the centre crop;
 * their box_xy / box_wh are mapped back as xy*zoom_mul + zoom_add, wh*zoom_mul (the reference hard-codes
 * 224/416 and (416-224)/(2*416)) and concatenated with the plain pass on the ANCHOR axis, so a cell holds 2A
 * boxes: boxes [B,2N,4], scores [B,C,2N], index ((h*G+w)*2A + pass*A + a) inside a scale. */
int yr_decode_zoom(const float* y1, const float* y2, const float* y3, const float* z1, const float* z2,
                   const float* z3, float zoom_mul, float zoom_add, int batch, int in_h, int in_w,
                   int num_anchors, int num_classes, int num_scales, const float* anchors_host,
                   const int32_t* image_hw, float* boxes, float* scores, void* stream);

/* yolo_head alone (model.py:344-371): one scale, reference layouts
 * box_xy/box_wh [B,G,G,A,2], conf [B,G,G,A,1], probs [B,G,G,A,C]; `scores` (nullable)
 * receives conf*probs [B,G,G,A,C] (model.py:426). */
int yr_yolo_head(const float* feats, int batch, int gh, int gw, int num_anchors, int num_classes,
                 const float* anchors_host /*A x (w,h)*/, int in_h, int in_w,
                 float* box_xy, float* box_wh, float* conf, float* probs, float* scores, void* stream);
/* yolo_correct_boxes (model.py:374-399): n = number of (xy,wh) pairs per image. */
int yr_correct_boxes(const float* box_xy, const float* box_wh, int batch, int64_t n_per_image,
                     int in_h, int in_w, const int32_t* image_hw, float* boxes, void* stream);

/* ---- per-(image,class) hard NMS: replaces the loop of tf.image.non_max_suppression
 * calls at model.py:474-480 (NonMaxSuppressionV3 semantics).
 *   out_idx [B,C,max_boxes] int32 box indices in pick order (unused tail = -1)
 *   out_count [B,C] int32 */
int yr_nms(const float* boxes, const float* scores, int batch, int n, int num_classes, int max_boxes,
           float score_thr, float iou_thr, int32_t* out_idx, int32_t* out_count, void* stream);

/* ---- gather + cast: model.py:481-490.  Fixed-size padded records so that the
 * multi-GPU all-gather moves one dense tensor:
 *   det [B, C*max_boxes, 6] int32 words = {ymin,xmin,ymax,xmax (int32, truncated),
 *        score (fp32 bits), class (int32)}, rows ordered class-ascending then pick
 *        order, compacted to the front; det_count [B] int32. */
int yr_pack_detections(const float* boxes, const float* scores, const int32_t* nms_idx,
                       const int32_t* nms_count, int batch, int n, int num_classes, int max_boxes,
                       int32_t* det, int32_t* det_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLORET_HIP_H */

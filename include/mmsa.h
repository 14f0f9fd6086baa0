/* mmsa.h -- C ABI of libmmsa_hip.so: the MI355X (gfx950) kernels of the MM-SAM-Adapter image-encoder forward.
 *
 * Drop-in boundary.  The reference's only native interface on this path is the pybind11 module
 * `MultiScaleDeformableAttention` (segmentation/ops/src/vision.cpp:13-16) whose forward entry
 * `ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)`
 * (segmentation/ops/src/ms_deform_attn.h:20-39 -> cuda/ms_deform_attn_cuda.cu:20-80) is replaced by
 * `mmsa_ms_deform_attn_forward` below.  Every other arithmetic op of the path is a stock PyTorch/ATen call in
 * the reference (F.linear, F.conv2d, F.layer_norm, softmax, F.interpolate ...); the remaining entries are the
 * hand-written kernels that take their place (file:line of the replaced call site is given per entry).
 *
 * Conventions
 *  - plain C, no torch types; every pointer is a DEVICE pointer unless stated; `long` is 64-bit.
 *  - the caller owns all memory (inputs, outputs, workspaces); the library never allocates or frees.
 *  - every call enqueues work on `stream` (a hipStream_t) and returns immediately; no host sync inside, so the
 *    whole forward can be captured into a HIP graph.
 *  - the library is stateless: no environment variable is read and there is no writable global besides the thread-local error string
 *    (A/B knobs exist only in debug-knob builds of single sources, tools/build_variant.sh -DMMSA_DEBUG_KNOBS); what used to be set
 *    through `mmsa_debug_*_flavour` / `mmsa_gemm_next_extras` is an explicit argument of the call it concerns.
 *  - return 0 on success, negative on error; `mmsa_last_error()` returns a thread-local message.  Shapes,
 *    alignment and strides are validated on the host BEFORE anything is launched.
 *  - activations are fp32, token-major / NHWC: a [rows, channels] matrix with a row stride `ld*` in elements.
 *  - GEMM weights -- and the intermediate activations that feed GEMMs / attention -- are operand "planes": row r of a
 *    [rows, K] matrix (K padded to a multiple of 32) is 2K uint16 = one 128-byte line per 32-wide k-block (everything an MFMA
 *    k-step needs from that row).  Plane pointers are 128-byte aligned, row strides multiples of 64.  Two formats:
 *      MMSA_FMT_B3  bf16 hi/lo ("split3": x = hi + lo, three bf16 MFMA products, fp32 accumulate): the 32 hi values, then the 32 lo values;
 *      MMSA_FMT_H8  fp16 hi + e5m2 cross-term bytes (x = hi + lo; hi.hi on the fp16 MFMA, both cross terms of two k-blocks on
 *                   one block-scaled fp8 MFMA at twice the rate: same precision class, 2/3 of the matrix-pipe time): 32 fp16 hi
 *                   values, then four 16-byte chunks, chunk g = 8 bytes e5m2(lo * 2^11) + 8 bytes e5m2(hi) of k = 8g..8g+7 for an
 *                   ACTIVATION (A operand) row, the two halves swapped for a WEIGHT (W operand) row.  |x| is clamped to 57344.
 *      MMSA_FMT_F3  (round 4) the MMSA_FMT_B3 layout with fp16 halves: x = fp16(x) + fp16(x - fp16(x)), 22 significant bits instead of 16, the same
 *                   three MFMAs per product (on the fp16 MFMA); values are clamped to +-65504.  Read by mmsa_gemm_split3 (A as planes, or fp32 A
 *                   with fp32 outputs) and mmsa_convnext_mlp_fused; written by the GEMM, by mmsa_layernorm_rows, mmsa_split_planes (kind 4) and (round 6) mmsa_msda_fused / mmsa_dwconv_nhwc.
 *                   The TwinConvNeXt chain uses it.
 *      MMSA_FMT_H8C the h8 arithmetic on 3 bytes per element (round 4), laid out for the LDS-DMA operand stream of the GEMM: q(hi) is not
 *                   stored (the e5m2 image of an fp16 value is its top byte; the GEMM takes it in registers) and rows are stored in PAIRS --
 *                   pair j of a [rows, K] matrix (K padded to a multiple of 64, rows to even) occupies `ld` uint16 (>= 3 K):
 *                   [row 2j: K fp16 hi][row 2j+1: K fp16 hi][K / 64 lines of 128 bytes: chunk c = {row 2j: 64 lo bytes | row 2j+1: 64 lo bytes}],
 *                   a row's 64 lo bytes of a chunk = 4 groups g of 16 bytes = e5m2(lo * 2^11 * 1.09375) of k = 64c + 8g .. +7, then of k = 64c + 32 + 8g .. +7
 *                   (the factor 1.09375 makes up for q(hi) being the TRUNCATED top byte of hi in this format: csrc/common.h MMSA_H8C_LO_COMP).
 *                   Every `ld*` of h8c planes is the row-PAIR stride; activations and weights share the layout.  1.5 cache lines per row and
 *                   64 k-values where MMSA_FMT_H8 has 2: the GEMM's L2 -> LDS stream is what bounds its k loop (csrc/gemm_h8c.hip).
 *    `mmsa_split_planes` converts fp32; producer kernels emit the format of their `*_fmt` argument (h8c: mmsa_gemm_split3, mmsa_layernorm_rows,
 *    the three attention entries, mmsa_msda_fused, mmsa_split_planes).
 *  - Clamp watch (round 5).  The fp16-based formats clamp what they cannot hold (h8 / h8c: |x| > 57344, f3: |x| > 65504; bf16 hi/lo planes have fp32's
 *    range).  The entries that convert UNBOUNDED fp32 values to planes -- mmsa_gemm_split3, mmsa_layernorm_rows, mmsa_split_planes, mmsa_msda_fused,
 *    mmsa_dwconv_nhwc -- take `clamp_max`, an optional DEVICE float (NULL = no watch): a launch that had to clamp folds the largest |value| it met beyond
 *    the format's range (the GEMM's register epilogue: at least the format's limit -- it reports THAT it clamped, not by how much) into it with an atomic max (never lowered; the caller zeroes it).  0 after a forward = every operand plane holds its value.  The
 *    attention entries need none (their outputs are convex combinations of v, which the qkv GEMM's watch has seen).  The reference computes in fp32
 *    and has no such range; mmsa/backbone.py reads the word with the attention guard words and refuses a forward that clamped.
 */
#ifndef MMSA_H
#define MMSA_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* A binding sees an opaque pointer.  The library's own sources include this header too (csrc/common.h defines MMSA_BUILDING_LIBRARY behind
 * <hip/hip_runtime.h>), with the stream spelled as the HIP type they use, so that every definition is compiled against its declaration here:
 * a definition whose argument list differs from the prototype is a "conflicting types" error of the library build, not silent stack garbage
 * in a caller.  Both spellings are one pointer in the ABI. */
#ifdef MMSA_BUILDING_LIBRARY
typedef hipStream_t mmsa_stream_t;
#else
typedef void* mmsa_stream_t; /* hipStream_t */
#endif

/* mmsa_version() returns the MMSA_ABI_VERSION (mmsa_version.h) the library was built with; a binding must refuse a library whose number differs from
 * the header it was written against (mmsa/lib.py does) -- argument lists are not self-describing through a C ABI. */
#include "mmsa_version.h"
int mmsa_version(void);
const char* mmsa_last_error(void);

/* testing aid: fill the LDS of every CU with `pattern` (finds kernels that read LDS they did not write) */
int mmsa_debug_poison_lds(unsigned pattern, mmsa_stream_t stream);
/* HIP events on the launch stream (for bench.py; torch.cuda.Event only sees torch's current stream). */
int mmsa_event_create(void** ev);
int mmsa_event_record(void* ev, mmsa_stream_t stream);
int mmsa_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int mmsa_event_destroy(void* ev);

/* activation codes of the fused epilogues */
/* MMSA_ACT_GELU is the erf form (nn.GELU's default: IE:154-167, TC:107-111), evaluated to 3.3e-7 absolute in fp32 (csrc/common.h gelu1 / gelu2) */
enum { MMSA_ACT_NONE = 0, MMSA_ACT_GELU = 1, MMSA_ACT_RELU = 2, MMSA_ACT_RELU6 = 3, MMSA_ACT_HSWISH = 4, MMSA_ACT_SIGMOID = 5 };

/* scalar type codes of the dtype-dispatched entry points (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF) */
enum { MMSA_DT_F32 = 0, MMSA_DT_F16 = 1, MMSA_DT_F64 = 2 };

/* operand-plane formats (see Conventions) */
enum { MMSA_FMT_B3 = 0, MMSA_FMT_H8 = 1, MMSA_FMT_H8C = 2, MMSA_FMT_F3 = 3 };

/* --- reference native ops -----------------------------------------------------------------------------------
 * ms_deform_attn_forward (vision.cpp:14 -> ms_deform_attn.h:20-39 -> cuda/ms_deform_attn_cuda.cu:20-80).
 * value [N,S,M,D], spatial_shapes int64 [L,2] (H,W), level_start_index int64 [L], sampling_loc [N,Lq,M,L,P,2] (x,y in
 * [0,1]), attn_weight [N,Lq,M,L,P]; out [N,Lq,M*D] (overwritten).  `dtype` = the scalar type of value / sampling_loc /
 * attn_weight / out (ms_deform_attn_cuda.cu:64 dispatches float, double and half; f16 is computed in fp32 here).
 * Error behaviour mirrors ms_deform_attn_cuda.cu:52: batch % min(batch, im2col_step) must be 0. */
int mmsa_ms_deform_attn_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const void* sampling_loc, const void* attn_weight, void* out, int batch,
                                int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                int num_point, int im2col_step, int dtype, mmsa_stream_t stream);

/* ms_deform_attn_backward (vision.cpp:15 -> ms_deform_attn.h:42-61 -> cuda/ms_deform_attn_cuda.cu:83-151, kernels
 * ms_deform_im2col_cuda.cuh:301-920).  grad_output [N,Lq,M*D]; grad_value / grad_sampling_loc / grad_attn_weight have the
 * shapes of value / sampling_loc / attn_weight and are caller-allocated; the call zero-fills and then fully writes them on
 * `stream` (the reference returns freshly allocated zeros-initialised tensors, :121-123).  grad_value is a scatter of
 * floating-point atomic adds like the reference's, so its last bits depend on the arrival order. */
int mmsa_ms_deform_attn_backward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const void* sampling_loc, const void* attn_weight, const void* grad_output,
                                 void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int batch,
                                 int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                 int num_point, int im2col_step, int dtype, mmsa_stream_t stream);

/* Fused hot-path form of MSDeformAttn.forward's middle part (ops/modules/ms_deform_attn.py:105-127): takes the raw
 * output `raw` [N*Lq, ldraw] of the concatenated sampling_offsets|attention_weights projection
 * (columns [0, M*L*P*2) offsets, then M*L*P logits), the per-query reference points [Lq,2], does softmax over L*P,
 * loc = ref + off/(W_l,H_l), and the sampling gather.  out [N*Lq, ldo]. */
int mmsa_msda_fused(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                    const float* raw, long ldraw, const float* ref_points, float* out, long ldo,
                    uint16_t* out_planes, long ldop /* optional operand planes output */, int out_fmt /* MMSA_FMT_* */, int batch,
                    int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                    float* clamp_max /* optional clamp watch word of out_planes, see Conventions */, mmsa_stream_t stream);

/* mmsa_msda_fused on fp16 values (round 6; the model's opt-in `msda_value` attribute).  The gather is bound by the bytes it pulls through the CU's vector-memory
 * pipe; `value_planes` = the value projection's output (ops/modules/ms_deform_attn.py:103-104) as MMSA_FMT_H8 ACTIVATION planes [batch * spatial_size, row stride
 * ldvp >= 2 * num_heads * channels], written by that GEMM's own epilogue.  lo_bytes = 0: a corner reads the 8 fp16 hi values of a lane's 8 channels (half the
 * bytes of fp32 values; the value is rounded to 11 significant bits); lo_bytes != 0: + their e5m2 lo bytes (3/4 of the bytes, ~14 bits).  Everything else as
 * mmsa_msda_fused; channels % 8 == 0, num_heads * channels % 32 == 0. */
int mmsa_msda_fused_planes(const uint16_t* value_planes, long ldvp, int lo_bytes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* raw, long ldraw, const float* ref_points, float* out, long ldo,
                           uint16_t* out_planes, long ldop, int out_fmt, int batch,
                           int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                           float* clamp_max, mmsa_stream_t stream);

/* --- GEMM (replaces F.linear / 1x1 conv / patchify conv / ConvTranspose2d 2x2 s2) ---------------------------
 * C = beta*resid + colscale[n] * alpha * act(A[M,K] W[N,K]^T + bias[n]); batch > 1 = strided batched (strides in
 * elements; strideW / strideBias may be 0 to share; colscale, when given, uses strideBias too).  K % 32 == 0.  resid_mod > 0: residual row = row % resid_mod
 * (broadcast over the batch, e.g. pos_embed).  out_mode 1 = 2x2 pixel-shuffle store for ConvTranspose2d(k=2,s=2)
 * (BK:55,324): row (b,h,w), column (i,j,co) -> row (b,2h+i,2w+j), column co; resid uses the destination index.
 * Call sites replaced: IE:488,499,162-167; TC:107-111,297-304,328-335; AM:947-950,447-451,87-89,121-126,286-290;
 * ops/modules/ms_deform_attn.py:103,107-110,129; BK:324.
 * A is EITHER fp32 (`A`, split to hi/lo while staged) OR interleaved activation planes (`Ap`, written by the
 * producing kernel; lda/strideA then count uint16 elements, lda >= 2K).  W: interleaved planes, row stride 2K.
 * The result goes to fp32 `C`, to planes `Cp` (row stride ldcp >= 2*N rounded up to 64), or both.
 * fmt = format of the A and W planes (MMSA_FMT_H8 / MMSA_FMT_H8C: A must come as planes, K % 64 == 0, any M -- rows beyond M are clamped on the way in and masked on the way out; lda / ldcp of h8c planes =
 * row-pair strides >= 3 K / 3 pad64(N)); cp_fmt = format written to `Cp`:
 * bits 0..7 MMSA_FMT_*, bits 8.. = split / 32 -- columns >= split (a multiple of 32; 0 = none) are written as MMSA_FMT_H8 planes
 * whatever the base format (the qkv projection: q and k as f3 planes, v with an fp16 hi part for the attention kernels' v_fmt = 1).
 * max_grid > 0 caps the number of persistent workgroups (a caller running independent chains on concurrent streams gives each
 * its share of the CUs); 0 = all CUs.  Results do not depend on it. */
/* LayerNorm folded into a producer / consumer pair of GEMMs (base/image_encoder.py:396-421: x -> norm1 -> qkv, x -> norm2 -> lin1): three
 * optional arguments of the call (it must then be one the LDS-DMA kernel takes: activation planes, M >= 128):
 *   rowstats_out [M, N/64, 2] fp32: the call also writes, per output row and 64-column strip, the sum and the sum of squares of the fp32
 *     values it stores (plain fp32 output, no activation, N % 64 == 0) -- the producer of the residual stream;
 *   rownorm_mean_rstd [M, 2] + rownorm_colsum [N] (batch stride = strideBias): the call runs on the RAW stream's planes against W o w and its
 *     epilogue computes rstd_r * (acc - mean_r * colsum_n) + bias_n before the activation (planes-only output, N % 128 == 0) -- the consumer.
 * mmsa_rowstats_finalize turns the strip sums into (mean, rstd) per row (D = 64 * strips columns, biased variance, eps inside the root).
 * flavour: 0 = workgroup shape chosen by the problem (256-row ping-pong tiles; 128-row tiles, two workgroups per CU, for bf16 hi/lo
 * operands with K <= 256); 4 / 8 force the 128- / 256-row form (tests, A/B runs).  Results are bit-identical either way. */
int mmsa_rowstats_finalize(const float* rowstats, int rows, int strips, int D, float eps, float* mean_rstd, mmsa_stream_t stream);

/* Zero-fill `bytes` bytes at p on `stream` (hipMemsetAsync): the statistics blocks the neck's kernels accumulate into.  Host-side mirror: the
 * `.zero_()` calls of the reference's own accumulator initialisations are torch kernels; inside a captured step every node is the library's. */
int mmsa_zero_bytes(void* p, size_t bytes, mmsa_stream_t stream);
int mmsa_gemm_split3(const float* A, const uint16_t* Ap, long lda, long strideA,
                     const uint16_t* Wp, long strideW,
                     const float* bias, long strideBias, const float* colscale, const float* resid, long ldr,
                     long strideR, int resid_mod, float beta, float* C, long ldc, long strideC,
                     uint16_t* Cp, long ldcp, long strideCp, int M, int N, int K,
                     int batch, int act, float alpha, int out_mode, int ps_H, int ps_W, int ps_C, int fmt, int cp_fmt,
                     int max_grid, float* rowstats_out, const float* rownorm_mean_rstd, const float* rownorm_colsum, int flavour,
                     float* clamp_max /* optional clamp watch word of the planes output, see Conventions */, mmsa_stream_t stream);

/* ConvNeXt pointwise pair of the narrow stages as ONE kernel (TC:107-132): x[b] <- x[b] + gamma[b] * (GELU(A[b] W1[b]^T + b1[b]) W2[b]^T
 * + b2[b]); A = LayerNorm output as bf16 hi/lo planes [M, C] (row stride lda, batch stride strideA, uint16 units), W1 [4C, C] and
 * W2 [C, 4C] bf16 hi/lo planes per batch, b1 [batch, 4C], b2 / gamma [batch, C], x fp32 [M, C] (row stride ldx) updated in place.
 * C = 96 (the narrowest ConvNeXt stage); the 4C-wide hidden tensor stays in LDS.  max_grid as in mmsa_gemm_split3.  fmt = MMSA_FMT_B3 or
 * MMSA_FMT_F3: the format of A, W1, W2 (and of the hidden image the kernel keeps in LDS). */
int mmsa_convnext_mlp_fused(const uint16_t* Ap, long lda, long strideA, const uint16_t* W1p, long strideW1, const uint16_t* W2p,
                            long strideW2, const float* b1, const float* b2, const float* gamma, float* x, long ldx, long strideX,
                            int M, int C, int batch, int max_grid, int fmt,
                            float* clamp_max /* optional clamp watch word of the hidden tensor's planes (fmt = MMSA_FMT_F3), see Conventions */, mmsa_stream_t stream);

/* fp32 [rows, cols] (row stride ld) -> planes [rows, 2*cols_pad], zero padded (cols_pad % 32 == 0).
 * kind 0: bf16 hi/lo; 1: h8 activation rows; 2: h8 weight rows; 3: h8c planes [ceil(rows / 2), 3*cols_pad] (cols_pad % 64 == 0; activations and weights);
 * 4: f3 (fp16 hi/lo pairs in the bf16 hi/lo layout). */
int mmsa_split_planes(const float* src, long ld, int rows, int cols, int cols_pad, uint16_t* planes, int kind,
                      float* clamp_max /* optional clamp watch word, see Conventions */, mmsa_stream_t stream);

/* --- attention (IE:465-501 incl. window_partition/unpartition IE:504-551 and rel-pos IE:587-623) -------------
 * Attention logit guard: the planes entries take `max_abs_logit`, an optional DEVICE float (NULL = none).  The launch folds the largest
 * |logit| it scores -- scale * q.k + the rel-pos terms, natural units, over existing keys (pad tokens of a window included, they are
 * attended to) and live queries -- into it with an atomic max (never lowered; the caller zeroes it).  The reference computes attention
 * in fp32 (IE:488-499) and needs no such word; this library chooses the operand precision of a block's attention (v_fmt) from it:
 * the host reads it after the step and moves a block whose logits outgrow single fp16 operands to fp16 hi/lo PAIRS (f3 planes; mmsa/backbone.py).
 * qkv [B*H*W, ldq] = q|k|v, channel = head*head_dim + c; qkv_bias [3*D]; rp from mmsa_relpos_bias;
 * window_size 0 = global.  out [B*H*W, ldo]. head_dim in {32, 64}. */
int mmsa_attention(const float* qkv, long ldq, const float* qkv_bias, const float* rp, float* out, long ldo, int B,
                   int H, int W, int heads, int head_dim, int window_size, float scale, mmsa_stream_t stream);
/* same with qkv [.., 2*3D], qkv_bias [2*3D] and the output [.., 2*D] as interleaved planes (strides in uint16) */
int mmsa_attention_planes(const uint16_t* qkv_planes, long ldq, const uint16_t* bias_planes, const float* rp,
                          uint16_t* out_planes, long ldo, int B, int H, int W, int heads, int head_dim,
                          int window_size, float scale, int out_fmt /* MMSA_FMT_* of out_planes */,
                          int v_fmt /* 0: qkv_planes / bias_planes (and the rel-pos planes of the fused entries) are MMSA_FMT_F3 planes -- fp16 hi/lo pairs, three fp16 MFMAs per product; round 4: bf16 hi/lo planes before, same layout, NOT accepted any more; 1 (this entry): their v columns are h8 planes (fp16 hi,
                                       as the qkv GEMM writes them with cp_fmt = MMSA_FMT_F3 | (2D/32) << 8): P V runs on the fp16 MFMA with
                                       P rounded to fp16; 2 (the two fused rel-pos entries below): qkv_planes, bias_planes AND relpos_planes
                                       are h8 planes throughout (and the window kernel's selector holds fp16 ones): every contraction of
                                       the kernel is one fp16 MFMA on the hi parts */,
                          float* max_abs_logit /* optional, see "Attention logit guard" */, mmsa_stream_t stream);

/* rel-pos bias terms: rp [B, heads, H*W, KH+KW]; Rh [QS,KH,head_dim], Rw [QS,KW,head_dim] = gathered tables
 * get_rel_pos(...)  (IE:554-584), (KH,KW,QS) = (ws,ws,ws) for windowed blocks or (H,W,max) for global ones. */
int mmsa_relpos_bias(const float* qkv, long ldq, const float* Rh, const float* Rw, float* rp, int B, int H, int W,
                     int heads, int head_dim, int window_size, mmsa_stream_t stream);
int mmsa_relpos_bias_planes(const uint16_t* qkv_planes, long ldq, const float* Rh, const float* Rw,
                            float* rp, int B, int H, int W, int heads, int head_dim, int window_size, mmsa_stream_t stream);

/* --- normalisation / reductions ------------------------------------------------------------------------------
 * Row LayerNorm (biased variance): y = (x-mean)/sqrt(var+eps)*w + b; optional y2 = x + y.  map_mode 1 scatters
 * token (b,h,w) of an [B,map_H,map_W] grid to row (b,h/2,w/2), column block (h&1)*2+(w&1) (the im2col layout of
 * the ConvNeXt 2x2 s2 downsample conv, TC:328-335).  Replaces nn.LayerNorm / LN2d / WithBias_LayerNorm:
 * IE:367,377; AM:479-487,519-520,51-74; mmpretrain_custom/models/utils/norm.py:51-90.
 * Row groups (group_rows > 0; the two TwinConvNeXt streams stacked along the rows, TC:445-476): group g = row / group_rows
 * uses w + g*w_gstride, b + g*w_gstride and writes at column offset g*y_gcol; y_wrap != 0: output row = row % group_rows
 * (channel-concatenation of the two streams' stage outputs, TC:466-472).  group_rows = 0: one group. */
int mmsa_layernorm_rows(const float* x, long ldx, const float* w, const float* b, float eps, float* y, long ldy,
                        float* y2, long ldy2, uint16_t* y_planes, long ldp /* optional interleaved planes of y */,
                        int rows, int C, int map_mode, int map_H, int map_W, int group_rows, long w_gstride, long y_gcol,
                        int y_wrap, int plane_fmt /* MMSA_FMT_* of y_planes */, float* clamp_max /* optional clamp watch word, see Conventions */,
                        mmsa_stream_t stream);

/* out (double) [B,3,C]: sum_p x, sum_p x^2, sum_p wrow[p]*x over the HW rows of each image (wrow may be NULL). */
int mmsa_colstats(const float* x, long ldx, long strideB, const float* wrow, int B, int HW, int C, double* out,
                  int out_is_zero /* 1: the caller zeroed `out` on this stream; 0: the call zeroes it first */, mmsa_stream_t stream);

/* GFFM LayerNorm(H*W) statistics + FFRM gate (AM:241,265 and AM:158-162), see csrc/norm.hip. */
int mmsa_ffrm_finalize(const double* stats, int B, int HW, int C, float mean_w, float mean_b, const float* Wc,
                       const float* gn_w, const float* gn_b, float* mean_o, float* rstd_o, float* mult_o,
                       float* scratch /* [2,B,C] */, mmsa_stream_t stream);
int mmsa_lnhw_apply(const float* x, long ldx, const float* mean, const float* rstd, const float* mult, const float* w,
                    const float* bias, float* y, long ldy, int B, int HW, int C, mmsa_stream_t stream);

/* --- convolutions on NHWC maps (stride 1, same padding) and patch gather ------------------------------------
 * dwconv: weights tap-major [k*k, C]; replaces TC:69-70,102; AM:288; AM:459,464-469.
 * gconv: weights [G][k*k][cin_g][cout_g]; replaces AM:87-88,123-124.
 * im2col_nchw: out[(b,ph,pw)][(c,kh,kw)] from NCHW input channels [c0, c0+Cin) (IE:658-663, TC:297-304). */
int mmsa_dwconv_nhwc(const float* x, long ldx, long xstrideB, const float* w, const float* bias, float* y, long ldy,
                     long ystrideB, uint16_t* y_planes, long ldp, long pstrideB /* optional operand planes */, int planes_fmt /* MMSA_FMT_* */,
                     int B, int H, int W, int C, int k, int act,
                     int imgs_per_group /* > 0: image group g = b / imgs_per_group uses w + g*k*k*C, bias + g*C */,
                     float* rowstats /* optional, 7x7 only (C % 64 == 0, H, W % 8 == 0): per pixel and 64-channel chunk (sum, sum of squares) of the
                                        output, [B*H*W][C/64][2]: the strip sums of the LayerNorm fold (mmsa_rowstats_finalize) */,
                     float* clamp_max /* optional clamp watch word, see Conventions */, mmsa_stream_t stream);
int mmsa_gconv_nhwc(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int B, int H,
                    int W, int G, int cin_g, int cout_g, int k, int act, mmsa_stream_t stream);
/* gated pair stage of the neck Mlp (AM:127-132): y = gelu(dw3x3(x)[:, :C]) * dw3x3(x)[:, C:], x token-major [B*H*W, 2C], the
 * depthwise conv has 2 channels per group (C groups), weights TAP-major [9][C][ci=2][co=2]; fp32 and/or interleaved-planes output */
int mmsa_dwpair_gate(const float* x, long ldx, const float* w, float* y, long ldy, uint16_t* y_planes, long ldp, int B, int H,
                     int W, int C, mmsa_stream_t stream);
int mmsa_im2col_nchw(const float* x, int B, int Ctot, int c0, int Cin, int H, int W, int p, float* out, int Kpad,
                     mmsa_stream_t stream);

/* --- modality-fusion neck pieces (AM:75-109, 234-267, 110-132, 176-221) ------------------------------------- */
/* G[b] = X[b]^T Y[b] over the P rows of each image: fp32 MFMA per 256-row slice, the slices summed in double in slice order (deterministic;
 * G needs no zeroing).  nblk > 1: only the entries of the nblk diagonal head blocks (c / nblk channels each) are written.  `scratch`:
 * caller-owned, >= mmsa_gram_tn_scratch_bytes(B, P, c) bytes, 16-byte aligned (the per-slice partial sums). */
long mmsa_gram_tn_scratch_bytes(int B, int P, int c);
int mmsa_gram_tn(const float* X, long ldx, const float* Y, long ldy, long strideB, double* G /* [B,c,c] */, int B, int P, int c,
                 int nblk, void* scratch, long scratch_bytes, mmsa_stream_t stream);
int mmsa_chanattn_build(const double* G, const double* sq, long sq_strideB, const double* sk, long sk_strideB,
                        const float* temp, const float* Wp, uint16_t* planes /* [B,c,2*cpad] */, int B, int c, int cpad,
                        int heads, mmsa_stream_t stream);
int mmsa_gffm_build(const double* E, uint16_t* x_planes, uint16_t* y_planes /* [B,c,2*cpad] each */, int B, int c,
                    int cpad, mmsa_stream_t stream);
int mmsa_gelu_gate(const float* x, long ldx, float* y, long ldy, long rows, int C, mmsa_stream_t stream);
int mmsa_pool_hw(const float* z, long ldz, float* out, long ldo, int B, int H, int W, int C, mmsa_stream_t stream);
int mmsa_ca_apply(const float* z, long ldz, const float* att, long lda, float* out, long ldo,
                  uint16_t* out_planes, long ldp /* optional interleaved planes */, int B, int H, int W, int C, mmsa_stream_t stream);

/* --- tail (BK:316-337): out NCHW [B,C,Hc,Wc] = (cmap + bilinear(xtok)) * bn_scale + bn_shift; cmap image b starts at b*cstrideB;
 * xtok == NULL: no ViT feature is added (add_vit_feature = False, BK:326) --- */
int mmsa_tail_fuse(const float* cmap, long ldc, long cstrideB, const float* xtok, long ldx, const float* bn_scale,
                   const float* bn_shift, float* out, uint16_t* out_planes /* optional: the same map token-major
                   [B*Hc*Wc, 2*C] as interleaved planes, for the decode head's first 1x1 conv */, long ldp,
                   int B, int Hc, int Wc, int Hx, int Wx, int C, mmsa_stream_t stream);

/* --- global attention with the rel-pos terms computed in the kernel (Attention.forward IE:465-501 + add_decomposed_rel_pos IE:587-623 on a
 *     window_size = 0 block): planes in / out as mmsa_attention_planes; relpos_planes = interleaved planes of a [256, 64] matrix, rows
 *     0..2H-2 = rel_pos_h and rows 128..128+2W-2 = rel_pos_w (tables already resized to 2H-1 / 2W-1 rows, IE:568-575).  W = 64,
 *     H <= 64 and a multiple of 4, head_dim 64.  No mmsa_relpos_bias pass. --- */
int mmsa_global_attention_planes(const uint16_t* qkv_planes, long ldq, const uint16_t* bias_planes, const uint16_t* relpos_planes,
                                 uint16_t* out_planes, long ldo, int B, int H, int W, int heads, int head_dim, float scale,
                                 int out_fmt /* MMSA_FMT_* of out_planes */, int v_fmt /* 0 or 2: see mmsa_attention_planes */,
                                 float* max_abs_logit /* optional */, mmsa_stream_t stream);

/* --- windowed attention with the rel-pos bias fused (Block.forward IE:382-423 on a window_size > 0 block: window_partition ->
 *     Attention.forward IE:465-501 + add_decomposed_rel_pos IE:587-623 -> window_unpartition).  qkv / bias / out as in
 *     mmsa_attention_planes; relpos_planes = interleaved planes of a [64, 64] matrix whose rows 0..2ws-2 are the block's
 *     rel_pos_h table (already resized to 2ws-1 rows, IE:568-575) and rows 32..32+2ws-2 its rel_pos_w.  selector =
 *     [208, 32] bf16 (row-major, 128-byte aligned): selector[j][j / ws] = selector[j][14 + j % ws] = 1.0 for j < ws*ws,
 *     0 elsewhere (a constant of the window size).  head_dim 64, window_size <= 14.  No mmsa_relpos_bias pass. --- */
int mmsa_window_attention_planes(const uint16_t* qkv_planes, long ldq, const uint16_t* bias_planes,
                                 const uint16_t* relpos_planes, const uint16_t* selector, uint16_t* out_planes, long ldo, int B, int H, int W,
                                 int heads, int head_dim, int window_size, float scale, int out_fmt /* MMSA_FMT_* of out_planes */,
                                 int v_fmt /* 0 or 2: see mmsa_attention_planes */, float* max_abs_logit /* optional */, mmsa_stream_t stream);

/* --- Segformer decode head (segmentation/mmseg_custom/models/decode_heads/segformer_head.py:47-66; the 1x1 convs are
 *     mmsa_gemm_split3 calls).  nchw_to_planes: backbone map [B,C,HW] fp32 (image b at b*strideB) -> interleaved planes
 *     [B*HW, C].  head_fuse: planes/out32 [B*H*W, C] = act((z0 + sum_i bilinear_{align_corners=False}(z_i -> HxW)) * bn_scale
 *     + bn_shift); z_i token-major [B*H_i*W_i, ld] fp32, NULL levels skipped.  tokens_to_nchw: [B*HW, ld] -> [B,C,HW]. --- */
int mmsa_nchw_to_planes(const float* src, long strideB, uint16_t* planes, long ldp, int B, int C, long HW,
                        mmsa_stream_t stream);
int mmsa_head_fuse(const float* z0, const float* z1, int H1, int W1, const float* z2, int H2, int W2, const float* z3,
                   int H3, int W3, long ld, const float* bn_scale, const float* bn_shift, uint16_t* planes, long ldp,
                   float* out32, long ldo, int B, int H, int W, int C, int act, mmsa_stream_t stream);
int mmsa_tokens_to_nchw(const float* src, long ld, float* dst, int B, long HW, int C, mmsa_stream_t stream);

/* --- segmentor glue (segmentation/mmseg_custom/models/segmentors/encoder_decoder.py): bilinear_accum writes (accumulate = 0) or adds
 *     (accumulate != 0) resize(src -> hc x wc, bilinear, align_corners=False) into window (y0, x0) of the NCHW canvas dst and, when count
 *     is given, adds 1 to count[b, y, x] over the window (ED:90-94, 213-219); div_count: preds / count_mat (ED:225); argmax over C
 *     (ED:477) -> uint8 map, first maximum wins. --- */
int mmsa_bilinear_accum_nchw(const float* src, long src_strideB, int B, int C, int hs, int ws, float* dst, int Hd, int Wd, int y0, int x0,
                             int hc, int wc, float* count, int accumulate, mmsa_stream_t stream);
int mmsa_div_count_nchw(float* x, const float* count, int B, int C, long HW, mmsa_stream_t stream);
int mmsa_argmax_nchw(const float* x, unsigned char* out, int B, int C, long HW, mmsa_stream_t stream);
/* crop extraction of slide inference (ED:205-212) for a batch of windows in one launch: dst[k] = src[b_k, :, y0_k:+hc, x0_k:+wc];
 * `windows` is a HOST array [n,3] = (image, y0, x0), n <= 64 (copied into the launch arguments). */
int mmsa_crop_batch_nchw(const float* src, int B, int C, int H, int W, const int* windows, int n, float* dst, int hc, int wc,
                         mmsa_stream_t stream);
/* class map of a whole sliding-window frame in one pass (ED:213-225 + ED:449,477): out[b,y,x] = argmax_c (sum over the covering windows,
 * in window order, of bilinear(logits_k -> hc x wc)[c]) / count -- the same additions in the same order as bilinear_accum + div_count +
 * argmax without the [B,C,H,W] canvas; one full-size window per image = resize + argmax of whole-image inference.  logits [n,C,hs,ws];
 * `uncovered` (device int, zeroed by the caller) counts pixels that no window covers (ED:220). */
int mmsa_slide_argmax(const float* logits, int n, int C, int hs, int ws, const int* windows, unsigned char* out, int B, int H, int W,
                      int hc, int wc, int* uncovered, mmsa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMSA_H */

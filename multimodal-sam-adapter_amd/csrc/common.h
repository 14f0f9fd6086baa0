// Shared helpers for the mmsa HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define MMSA_OK 0
#define MMSA_ERR_ARG (-1)
#define MMSA_ERR_LAUNCH (-2)

// The C ABI's own header: every `extern "C"` definition of the library is compiled against its prototype there (a mismatch in the argument list is a
// "conflicting types" error of this build), and the enums (MMSA_DT_*, MMSA_FMT_*, MMSA_ACT_*) have one definition.
#define MMSA_BUILDING_LIBRARY 1
#include "../../include/mmsa.h"
void mmsa_set_error(const char* fmt, ...);

// A/B and timing knobs.  The RELEASE library reads no environment variable and holds no writable global besides the thread-local error
// string: MMSA_KNOB(name, default) is the default, as a constant.  A debug-knob build of one source (tools/build_variant.sh <so> <file>
// -DMMSA_DEBUG_KNOBS) reads the integer environment variable `name` instead -- the experiments recorded in LAB_NOTES.md were run that way.
#ifdef MMSA_DEBUG_KNOBS
#include <stdlib.h>
static inline int mmsa_knob_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#define MMSA_KNOB(name_, dflt_) mmsa_knob_env(name_, dflt_)
#else
#define MMSA_KNOB(name_, dflt_) (dflt_)
#endif

#define MMSA_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      mmsa_set_error(__VA_ARGS__);           \
      return MMSA_ERR_ARG;                   \
    }                                        \
  } while (0)

#define MMSA_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      mmsa_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return MMSA_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)

// XCD-aware workgroup order (round 4).  The dispatcher deals the workgroups of a launch round-robin to the 8 XCDs in dispatch order (x fastest, then
// y, z), and each XCD has its own L2: neighbouring workgroups -- the rows above and below in a 3 x 3 stencil, the query blocks that read the same
// K / V, the queries that gather from the same patch of the value map -- land on eight different L2s and each fetches its own copy (measured: 3 x the
// input bytes for the neck's pair conv, 8 x the value map for the deformable-attention gather, 10 x K / V for global attention).  mmsa_xcd_order maps
// the dispatch-order id to a work index such that every XCD walks ONE contiguous eighth of the work in order.  A pure permutation of work indices:
// results do not depend on it (build with -DMMSA_XCD_ORDER=0 for the A/B).
#ifndef MMSA_XCD_ORDER
#define MMSA_XCD_ORDER 1
#endif
__device__ __forceinline__ unsigned mmsa_xcd_order(unsigned lin, unsigned total) {
#if MMSA_XCD_ORDER
  const unsigned per = total >> 3;
  return lin < (per << 3) ? (lin & 7u) * per + (lin >> 3) : lin;
#else
  (void)total;
  return lin;
#endif
}
// dispatch-order id of this workgroup / number of workgroups of the launch
__device__ __forceinline__ unsigned mmsa_block_lin() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ unsigned mmsa_block_count() { return gridDim.x * gridDim.y * gridDim.z; }

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B operand (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 accumulator

// ---- bf16 hi/lo split of an fp32 value ("split3" operands): x ~= hi + lo with ~16 mantissa bits.
// Round-to-nearest-even on the raw bits; inputs are finite activations/weights (NaN not preserved).
__device__ __forceinline__ unsigned short f32_to_bf16_rn(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// Hardware conversion (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserving): 5 VALU instructions split TWO
// floats into packed hi/lo pairs, instead of ~10 integer ops per float with the bit-twiddling form above.
typedef __attribute__((ext_vector_type(2))) __bf16 mmsa_bf16x2;
typedef __attribute__((ext_vector_type(2))) float mmsa_f32x2;

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  const mmsa_f32x2 v = {a, b};
  const mmsa_bf16x2 h = __builtin_convertvector(v, mmsa_bf16x2);
  const mmsa_f32x2 r = v - __builtin_convertvector(h, mmsa_f32x2);
  const mmsa_bf16x2 l = __builtin_convertvector(r, mmsa_bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
  unsigned h, l;
  split2(x, 0.f, h, l);
  hi = (unsigned short)(h & 0xFFFFu);
  lo = (unsigned short)(l & 0xFFFFu);
}

// split 4 floats -> 4 hi (packed in 2 dwords) + 4 lo
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  split2(v.x, v.y, hi.x, lo.x);
  split2(v.z, v.w, hi.y, lo.y);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32-lane half a lane belongs to
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- interleaved hi/lo planes ("ilv"): the storage format of every bf16-split matrix (weights and activations).
// Row r of a [rows, K] matrix (K % 32 == 0) is 2K bf16 values: for every 32-wide k-block first the 32 hi values,
// then the 32 lo values.  One k-block of one row is therefore ONE 128-byte line holding everything an MFMA k-step
// needs from that row (full-line L2 requests for the LDS-DMA; two 64-byte half-line requests per row per k-step
// was the measured limiter of the two-array layout).  Element k: hi at ilv(k), lo at ilv(k) + 32.
__host__ __device__ __forceinline__ int ilv(int k) { return ((k >> 5) << 6) | (k & 31); }

// ---- "h8" planes: the second operand format of the GEMM (same 128-byte-per-k-block geometry as the bf16 hi/lo planes).
// x = hi + lo with hi = fp16(x) (11 significant bits) and lo = x - hi (|lo| <= 2^-11 |x|).  The product is
//   a.b ~= hi_a.hi_b  +  [ q(hi_a).q(lo_b) + q(lo_a).q(hi_b) ]
// with the first term on the fp16 MFMA (v_mfma_f32_16x16x32_f16, exact products, fp32 accumulate) and BOTH cross terms of TWO
// k-blocks on ONE block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4: 2x the bf16 rate), whose operands q(.) are e5m2
// roundings (3 significant bits): the cross terms are 2^-12 of the product, so rounding them to 3 bits costs 2^-16 relative --
// the precision of the bf16 hi/lo scheme at 2/3 of its matrix-pipe time (3 bf16 MFMAs per k-block -> 1 fp16 + 1/2 fp8 at double
// rate = 2 units).  lo is stored scaled by 2^11 (so that it sits in the normal range of e5m2 whenever x does in fp16's); the
// MFMA's block scale (e8m0 = 127 - 11 on one operand) undoes it.  Measured end to end on the CPU oracle with every ViT-block and
// interaction Linear in this format (tools/precision_study.py): 2-4e-5 relative on f1..f4 (bf16 hi/lo: 0.5-1e-5; gate 1e-3).
// Row r, k-block j (128 bytes): [32 x fp16 hi][4 chunks of 16 B: chunk g = 8 lo bytes + 8 q(hi) bytes of k = 8g .. 8g+7]
// for ACTIVATIONS (A operand); WEIGHTS store the chunk as 8 q(hi) bytes + 8 lo bytes, so that for both operands the lane's 32
// operand bytes of the fp8 MFMA are just the chunks of two consecutive k-blocks, and byte p of A always meets byte p of W with
// the roles (lo, q(hi)) crossed.  Values are clamped to +-57344 (the largest e5m2 / a finite fp16) before the split.
// (enum MMSA_FMT_B3 = 0, MMSA_FMT_H8 = 1, MMSA_FMT_H8C = 2, MMSA_FMT_F3 = 3: include/mmsa.h, included above)
// Output-plane format argument of the GEMM (`cp_fmt`): bits 0..7 = format of the columns below the split, bits 8.. = split / 32;
// columns >= split (a multiple of 32, 0 = no split) are written as MMSA_FMT_H8.  The qkv projection writes q and k as bf16 hi/lo
// planes and v with an fp16 hi part this way (the attention kernels run P V on the fp16 MFMA: LAB_NOTES.md 4.1).
#define MMSA_CP_BASE(f_) ((f_) & 0xff)
#define MMSA_CP_SPLIT(f_) (((f_) >> 8) * 32)
#define MMSA_CP_AT(f_, col_) ((MMSA_CP_SPLIT(f_) > 0 && (col_) >= MMSA_CP_SPLIT(f_)) ? MMSA_FMT_H8 : MMSA_CP_BASE(f_))
#define MMSA_H8_MAX 57344.0f
#define MMSA_H8_LO_SCALE 2048.0f          // 2^11
#define MMSA_H8_MFMA_SCALE 0x74747474     // e8m0 127 - 11 in every byte: the block scale that undoes MMSA_H8_LO_SCALE
// h8c planes: q(hi) is the TRUNCATED top byte of the fp16 value (taken in registers), i.e. on average 9 % short of hi -- every cross term hi x lo would come
// out 9 % small, which is the larger part of that format's error.  The lo bytes of h8c planes are therefore stored scaled by 2^11 x 1.09375: the mean of
// trunc(hi) x 1.09375 lo is hi x lo again (random 512 x 1024 x 512 products: 3.11e-5 -> 2.13e-5 relative, rounded q(hi): 2.08e-5; profiles/r04_f3_study.txt).
#define MMSA_H8C_LO_COMP 1.09375f
typedef __attribute__((ext_vector_type(2))) _Float16 mmsa_h2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;   // operand of v_mfma_f32_16x16x32_f16
// two floats -> fp16 hi pair + fp16 lo pair (lo = a - hi, itself rounded to fp16: 22 significant bits together)
__device__ __forceinline__ void split2_f16(float a, float b, unsigned& hi, unsigned& lo) {
  const mmsa_f32x2 v = {a, b};
  const mmsa_h2 h = __builtin_convertvector(v, mmsa_h2);
  const mmsa_f32x2 r = v - __builtin_convertvector(h, mmsa_f32x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, mmsa_h2));
}
// two floats -> 2 packed fp16, round to nearest even (softmax probabilities of the attention kernels' fp16 P V)
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const mmsa_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mmsa_h2));
}

// ---- "f3" planes (round 4): the bf16 hi/lo layout (32 hi values, then 32 lo values per k-block) with fp16 halves -- x = hi + lo, hi = fp16(x),
// lo = fp16(x - hi): 22 significant bits where bf16 hi/lo has 16, the same three MFMAs per product (v_mfma_f32_16x16x32_f16: hh + hl + lh) at the
// same rate.  Used where operand rounding is what the outputs see: the TwinConvNeXt chain, whose error GFFM multiplies by ~15 (LAB_NOTES.md section 2;
// tools/f3_study.py).  fp16's range is the price: values are clamped to +-65504 (LayerNorm outputs, GELU hidden activations and weights are far
// inside it); below 6.1e-5 hi and lo become subnormal -- the MFMA does not flush them -- and the pair degrades gracefully to an absolute 6e-8.
#define MMSA_F3_MAX 65504.0f
__device__ __forceinline__ void f3_split2(float a, float b, unsigned& hi, unsigned& lo) {
  a = __builtin_amdgcn_fmed3f(a, -MMSA_F3_MAX, MMSA_F3_MAX);
  b = __builtin_amdgcn_fmed3f(b, -MMSA_F3_MAX, MMSA_F3_MAX);
  split2_f16(a, b, hi, lo);
}
__device__ __forceinline__ void f3_split4(const float4 v, uint2& hi, uint2& lo) {
  f3_split2(v.x, v.y, hi.x, lo.x);
  f3_split2(v.z, v.w, hi.y, lo.y);
}
// 4 values -> hi / lo halves of a 16-bit pair format (bf16 hi/lo or f3)
__device__ __forceinline__ void split4_fmt(const float4 v, uint2& hi, uint2& lo, int fmt) {
  if (fmt == MMSA_FMT_F3) f3_split4(v, hi, lo);
  else split4(v, hi, lo);
}

// two floats -> hi (2 packed fp16), lo8 / qh8 (2 e5m2 bytes each, written into the low or high half of `lo8` / `qh8`)
template <bool UPPER, bool COMP = false>
__device__ __forceinline__ void h8_split2(float a, float b, unsigned& hi, unsigned& lo8, unsigned& qh8) {
  a = __builtin_amdgcn_fmed3f(a, -MMSA_H8_MAX, MMSA_H8_MAX);
  b = __builtin_amdgcn_fmed3f(b, -MMSA_H8_MAX, MMSA_H8_MAX);
  const mmsa_f32x2 v = {a, b};
  const mmsa_h2 h = __builtin_convertvector(v, mmsa_h2);            // round to nearest even
  const mmsa_f32x2 hf = __builtin_convertvector(h, mmsa_f32x2);
  constexpr float ls_ = COMP ? MMSA_H8_LO_SCALE * MMSA_H8C_LO_COMP : MMSA_H8_LO_SCALE;
  const mmsa_f32x2 l = (v - hf) * ls_;   // (exact without COMP); packed: v_pk_add_f32 + v_pk_mul_f32 for the pair -- the GEMM epilogues are issue-bound (profiles/r05_gelu_forms.txt)
  hi = __builtin_bit_cast(unsigned, h);
  lo8 = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(l.x, l.y, (int)lo8, UPPER);
  qh8 = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(hf.x, hf.y, (int)qh8, UPPER);
}
// four floats -> 4 fp16 (8 bytes), 4 lo bytes, 4 q(hi) bytes
template <bool COMP = false>
__device__ __forceinline__ void h8_split4(const float4 v, uint2& hi, unsigned& lo8, unsigned& qh8) {
  lo8 = 0u; qh8 = 0u;
  h8_split2<false, COMP>(v.x, v.y, hi.x, lo8, qh8);
  h8_split2<true, COMP>(v.z, v.w, hi.y, lo8, qh8);
}
// byte offset, inside its row, of the 4-byte group holding the lo bytes of columns c .. c+3 (c % 4 == 0) of an ACTIVATION row;
// the q(hi) bytes sit 8 bytes further
__host__ __device__ __forceinline__ int h8_lo_off(int c) { return ((c >> 5) << 7) + 64 + (((c & 31) >> 3) << 4) + (c & 7); }

// Store 4 consecutive columns c .. c+3 (c % 4 == 0) of a planes row in either format (row = start of the row, uint16 units).
__device__ __forceinline__ void store_planes4(unsigned short* row, int c, const float4 v, int fmt) {
  if (fmt == MMSA_FMT_H8) {
    uint2 hi; unsigned lo8, qh8;
    h8_split4(v, hi, lo8, qh8);
    *reinterpret_cast<uint2*>(row + ilv(c)) = hi;
    unsigned char* rb = reinterpret_cast<unsigned char*>(row) + h8_lo_off(c);
    *reinterpret_cast<unsigned*>(rb) = lo8;
    *reinterpret_cast<unsigned*>(rb + 8) = qh8;
  } else {
    uint2 hh, ll;
    split4_fmt(v, hh, ll, fmt);
    *reinterpret_cast<uint2*>(row + ilv(c)) = hh;
    *reinterpret_cast<uint2*>(row + ilv(c) + 32) = ll;
  }
}

// The same for a PAIR of lanes (lane ^ XOR) that hold columns c8 .. c8+3 (the `odd == false` lane) and c8+4 .. c8+7 (`odd == true`),
// c8 % 8 == 0: the pair's 8 columns are one 16-byte hi chunk and one 16-byte second chunk of the line, so after exchanging halves the
// even lane stores the hi chunk and the odd lane the other one -- ONE 16-byte store per lane, and a wave instruction writes whole
// 128-byte lines (8- and 4-byte stores write every line as several partial-line requests).  Both lanes of a pair must be active and
// agree on `do_store`.
template <int XOR>
__device__ __forceinline__ void store_planes8_pair(unsigned short* row, int c8, const float4 v, int fmt, bool odd, bool do_store) {
  uint2 mine_hi, mine_x;   // hi chunk half; second-chunk half (bf16: lo values; h8: .x = lo bytes, .y = q(hi) bytes)
  if (fmt == MMSA_FMT_H8) h8_split4(v, mine_hi, mine_x.x, mine_x.y);
  else split4_fmt(v, mine_hi, mine_x, fmt);
  const uint2 snd = odd ? mine_hi : mine_x;
  uint2 rcv;
  rcv.x = __shfl_xor(snd.x, XOR, 64);
  rcv.y = __shfl_xor(snd.y, XOR, 64);
  uint4 pk;
  if (!odd) pk = make_uint4(mine_hi.x, mine_hi.y, rcv.x, rcv.y);
  else if (fmt == MMSA_FMT_H8) pk = make_uint4(rcv.x, mine_x.x, rcv.y, mine_x.y);
  else pk = make_uint4(rcv.x, rcv.y, mine_x.x, mine_x.y);
  const int off = odd ? (fmt == MMSA_FMT_H8 ? h8_lo_off(c8) >> 1 : ilv(c8) + 32) : ilv(c8);
  if (do_store) *reinterpret_cast<uint4*>(row + off) = pk;
}

// one element (ragged edges; rare)
__device__ __forceinline__ void store_planes1(unsigned short* row, int c, float x, int fmt) {
  if (fmt == MMSA_FMT_H8) {
    unsigned hi, lo8 = 0u, qh8 = 0u;
    h8_split2<false>(x, 0.f, hi, lo8, qh8);
    row[ilv(c)] = (unsigned short)(hi & 0xFFFFu);
    unsigned char* rb = reinterpret_cast<unsigned char*>(row) + h8_lo_off(c & ~3) + (c & 3);
    rb[0] = (unsigned char)(lo8 & 0xFFu);
    rb[8] = (unsigned char)(qh8 & 0xFFu);
  } else {
    unsigned short hh, ll;
    if (fmt == MMSA_FMT_F3) {
      unsigned h2, l2;
      f3_split2(x, 0.f, h2, l2);
      hh = (unsigned short)(h2 & 0xFFFFu);
      ll = (unsigned short)(l2 & 0xFFFFu);
    } else {
      split_bf16(x, hh, ll);
    }
    row[ilv(c)] = hh;
    row[ilv(c) + 32] = ll;
  }
}

// ---- "h8c" planes (round 4): the h8 arithmetic on 3 bytes per element instead of 4, laid out for the LDS-DMA operand stream of
// gemm_h8c.hip (an LDS-DMA instruction costs per 128-byte LINE it touches -- profiles/r03_dma_lanes_microbench.txt -- so the win is in
// lines, not in masked bytes).  q(hi) is not stored: e5m2 has fp16's exponent width, so the e5m2 image of an fp16 value is its top
// byte (truncation instead of round-to-nearest: the cross terms carry 2^-12 of a product, their operands' 3 significant bits cost
// 2^-15 either way), taken in registers with v_perm_b32.  A [rows, K] matrix (K % 64 == 0; rows padded to even) is stored by ROW PAIRS;
// pair j (rows 2j, 2j+1) occupies `ld` uint16 (>= 3 K):
//     [row 2j: K fp16 hi][row 2j+1: K fp16 hi][K / 64 lines of 128 B: chunk c = {row 2j: 64 lo bytes | row 2j+1: 64 lo bytes}]
// and a row's 64 lo bytes of chunk c (k = 64 c .. 64 c + 63) are 4 groups g of 16 B = [e5m2(lo * 2^11) of k = 64c + 8g .. + 7 | of k = 64c + 32 + 8g .. + 7]:
// the 16 bytes a lane of the fp8 MFMA needs from the two k-tiles of a chunk, contiguous.  Every 64-k chunk of a row is one whole line
// of hi values, every chunk of a row pair one whole line of lo bytes: 1.5 lines per row and chunk where the h8 line format has 2.
// Activations and weights use the SAME layout (the (lo, q(hi)) role swap of the h8 line format happens in registers).
struct H8cRow {
  unsigned short* hi;   // this row's K fp16 hi values
  unsigned char* lo;    // this row's 64 lo bytes of chunk 0; chunk c: + 128 c
};
__device__ __forceinline__ H8cRow h8c_row(unsigned short* base, long ld_pair, long row, int kpad) {
  unsigned short* pb = base + (row >> 1) * ld_pair;
  H8cRow r;
  r.hi = pb + (row & 1) * kpad;
  r.lo = reinterpret_cast<unsigned char*>(pb + 2 * kpad) + (row & 1) * 64;
  return r;
}
// byte offset, from H8cRow::lo, of the lo byte of column k
__host__ __device__ __forceinline__ int h8c_lo_off(int k) { return ((k >> 6) << 7) + (((k & 31) >> 3) << 4) + (((k >> 5) & 1) << 3) + (k & 7); }
// 4 consecutive columns c .. c+3 (c % 4 == 0) of one row
__device__ __forceinline__ void h8c_store4(const H8cRow r, int c, const float4 v) {
  uint2 hi; unsigned lo8, qh8;
  h8_split4<true>(v, hi, lo8, qh8);
  *reinterpret_cast<uint2*>(r.hi + c) = hi;
  *reinterpret_cast<unsigned*>(r.lo + h8c_lo_off(c)) = lo8;
}
// a PAIR of lanes (lane ^ XOR) holding columns c8 .. c8+3 (`odd == false`) and c8+4 .. c8+7 (`odd == true`), c8 % 8 == 0: the even lane
// stores the 16-byte hi chunk, the odd lane the 8 lo bytes.  Both lanes of a pair must be active and agree on `do_store`.
template <int XOR>
__device__ __forceinline__ void h8c_store8_pair(const H8cRow r, int c8, const float4 v, bool odd, bool do_store) {
  uint2 hi; unsigned lo8, qh8;
  h8_split4<true>(v, hi, lo8, qh8);
  const uint2 snd = odd ? hi : make_uint2(lo8, 0u);
  uint2 rcv;
  rcv.x = __shfl_xor(snd.x, XOR, 64);
  rcv.y = __shfl_xor(snd.y, XOR, 64);
  if (!do_store) return;
  if (!odd) *reinterpret_cast<uint4*>(r.hi + c8) = make_uint4(hi.x, hi.y, rcv.x, rcv.y);
  else *reinterpret_cast<uint2*>(r.lo + h8c_lo_off(c8)) = make_uint2(rcv.x, lo8);
}
// one element (ragged edges; rare)
__device__ __forceinline__ void h8c_store1(const H8cRow r, int c, float x) {
  unsigned hi, lo8 = 0u, qh8 = 0u;
  h8_split2<false, true>(x, 0.f, hi, lo8, qh8);
  r.hi[c] = (unsigned short)(hi & 0xFFFFu);
  r.lo[h8c_lo_off(c)] = (unsigned char)(lo8 & 0xFFu);
}
// Plane-output helpers over all three formats: `base` / `ld` as the entry points receive them (h8c: ld = row-PAIR stride), kpad = columns
// padded to 64 (h8c only).
template <int XOR>
__device__ __forceinline__ void store_planes8_pair_any(unsigned short* base, long ld, long row, int kpad, int c8, const float4 v, int fmt, bool odd, bool do_store) {
  if (fmt == MMSA_FMT_H8C) h8c_store8_pair<XOR>(h8c_row(base, ld, row, kpad), c8, v, odd, do_store);
  else store_planes8_pair<XOR>(base + row * ld, c8, v, fmt, odd, do_store);
}
__device__ __forceinline__ void store_planes4_any(unsigned short* base, long ld, long row, int kpad, int c, const float4 v, int fmt) {
  if (fmt == MMSA_FMT_H8C) h8c_store4(h8c_row(base, ld, row, kpad), c, v);
  else store_planes4(base + row * ld, c, v, fmt);
}
#define MMSA_PAD64(x_) (((x_) + 63) & ~63)

// ---- clamp watch (round 5, VERDICT r04 "the h8 / h8c clamp is silent").  The fp16-based operand formats clamp on the way in (h8 / h8c: +-57344, f3:
// +-65504; bf16 hi/lo planes have fp32's range and never clamp).  The kernels that CONVERT unbounded fp32 values to those formats -- the GEMM epilogues,
// LayerNorm, split_planes, the deformable-attention gather, the depthwise convs -- take an optional device float `clamp_max` (NULL = none): a lane keeps
// the largest |value| it converted, and when that exceeds the format's limit it is folded into the word with an atomic max (positive floats order like
// their bit patterns; never lowered: the caller zeroes it).  0 after a forward = nothing was clamped.  The attention kernels need no watch: their
// outputs are convex combinations of v, which the qkv GEMM's watch has already seen.
__device__ __forceinline__ float mmsa_clamp_limit(int fmt) {
  return (fmt == MMSA_FMT_H8 || fmt == MMSA_FMT_H8C) ? MMSA_H8_MAX : fmt == MMSA_FMT_F3 ? MMSA_F3_MAX : 3.0e38f;
}
#ifndef MMSA_CW_LEG
#define MMSA_CW_LEG 1
#endif
#ifndef MMSA_CW_REGS
#define MMSA_CW_REGS 1
#endif
#ifndef MMSA_CLAMP_WATCH
#define MMSA_CLAMP_WATCH 1   // 0 (A/B timing builds): the watch compiled out
#endif
__device__ __forceinline__ void clamp_see(float& m, const float4 v) {
  if (MMSA_CLAMP_WATCH) m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void clamp_see1(float& m, float x) { if (MMSA_CLAMP_WATCH) m = fmaxf(m, fabsf(x)); }
__device__ __forceinline__ void clamp_report(float* word, float m, float limit) {
  if (MMSA_CLAMP_WATCH && word != nullptr && m > limit) atomicMax(reinterpret_cast<int*>(word), __float_as_int(m));
}

// activation codes shared by the GEMM epilogue and the conv kernels
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_RELU6 = 3, ACT_HSWISH = 4, ACT_SIGMOID = 5 };

// GELU(x) = x Phi(x) (erf form, nn.GELU's default) as  h + |h| (1 - e),  h = x / 2,  e = erfc(|x| / sqrt 2) = exp2(P(min(|x|, 6)))  with P a degree-7 polynomial without
// constant term, fitted (weighted least squares, Lawson iterations: tools/gelu_fit.py) to log2 erfc so that |x| / 2 * |exp2(P) - erfc| is minimal over [0, 6]:
// 3.7e-8 with exact arithmetic, 3.3e-7 absolute in fp32 (the previous Abramowitz & Stegun 7.1.26 form: 4.6e-7 in fp32) -- two orders below the operand formats' rounding.
// Beyond |x| = 6, erfc < 2e-9 and 1 - e rounds to 1: the result is x or 0.  ONE transcendental per value instead of two (v_rcp_f32 + v_exp_f32), the same
// number of polynomial steps: a transcendental costs 2.5 issue slots of a plain instruction on this chip and the GELU of the MLP GEMMs is issue time of the epilogue, 26 of lin1's
// 163 us (profiles/r05_gelu_forms.txt).  gelu1 (single values: element-wise epilogues, the gated MLPs of the neck) and gelu2 (packed pairs: v_pk_fma_f32, two values
// per instruction) perform the same operations in the same order on every value: bit-identical results.
#ifndef MMSA_GELU_DEG
#define MMSA_GELU_DEG 7   // 6, 8 (A/B builds).  Fit error with exact arithmetic 8.2e-8 / 3.7e-8 / 9.0e-9, in fp32 3.6e-7 / 3.3e-7 / 3.1e-7: from degree 7 on the fp32 evaluation is the floor
#endif
#if MMSA_GELU_DEG == 6
#define MMSA_GELU_COEFFS {-1.151147127e+00f, -4.589156806e-01f, -5.323817953e-02f, 7.977456786e-03f, -7.398718735e-04f, 2.992386726e-05f}
#elif MMSA_GELU_DEG == 7
#define MMSA_GELU_COEFFS {-1.151126981e+00f, -4.590439200e-01f, -5.294858292e-02f, 7.670805324e-03f, -5.757985055e-04f, -1.279479329e-05f, 4.278379038e-06f}
#else
#define MMSA_GELU_COEFFS {-1.151111007e+00f, -4.591621459e-01f, -5.262760073e-02f, 7.245406508e-03f, -2.720646735e-04f, -1.314778056e-04f, 2.805791701e-05f, -1.902148711e-06f}
#endif
__device__ __forceinline__ float gelu1(float x) {
  constexpr float c[MMSA_GELU_DEG] = MMSA_GELU_COEFFS;
  const float a = fminf(fabsf(x), 6.0f);
  float p = fmaf(c[MMSA_GELU_DEG - 1], a, c[MMSA_GELU_DEG - 2]);
#pragma unroll
  for (int k = MMSA_GELU_DEG - 3; k >= 0; --k) p = fmaf(p, a, c[k]);
  p *= a;
  const float r = 1.0f - __builtin_amdgcn_exp2f(p);   // erf(|x| / sqrt 2)
  const float h = x * 0.5f;
  return fmaf(fabsf(h), r, h);
}
typedef __attribute__((ext_vector_type(2))) float mmsa_f2;
__device__ __forceinline__ mmsa_f2 gelu2(mmsa_f2 x) {
  mmsa_f2 a;
  a.x = fminf(fabsf(x.x), 6.0f);
  a.y = fminf(fabsf(x.y), 6.0f);
  constexpr float c[MMSA_GELU_DEG] = MMSA_GELU_COEFFS;
  mmsa_f2 p = a * c[MMSA_GELU_DEG - 1] + c[MMSA_GELU_DEG - 2];
#pragma unroll
  for (int k = MMSA_GELU_DEG - 3; k >= 0; --k) p = p * a + c[k];
  p = p * a;
  mmsa_f2 e;
  e.x = __builtin_amdgcn_exp2f(p.x);
  e.y = __builtin_amdgcn_exp2f(p.y);
  const mmsa_f2 r = 1.0f - e;
  const mmsa_f2 h = x * 0.5f;
  mmsa_f2 o;
  o.x = fmaf(fabsf(h.x), r.x, h.x);
  o.y = fmaf(fabsf(h.y), r.y, h.y);
  return o;
}
__device__ __forceinline__ float4 gelu4(float4 v) {
  const mmsa_f2 a = gelu2((mmsa_f2){v.x, v.y}), b = gelu2((mmsa_f2){v.z, v.w});
  return make_float4(a.x, a.y, b.x, b.y);
}

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case ACT_GELU: return gelu1(x);  // erf-form GELU (nn.GELU default)
    case ACT_RELU: return fmaxf(x, 0.0f);
    case ACT_RELU6: return fminf(fmaxf(x, 0.0f), 6.0f);
    case ACT_HSWISH: return x * (fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) / 6.0f);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    default: return x;
  }
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One-time PER-DEVICE launch setup.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy of a kernel, and the CU
// count is a property of the device: a process that packs a model on a second device (mmsa/backbone.py supports it) must not run there on the first device's
// settings -- a > 64 KiB LDS launch would fail, or the persistent grid would be sized for the wrong chip.  `st` = a call site's zero-initialised table;
// `setup()` runs the first time the current device is seen there; returns the current device's CU count.  (Idempotent, so a race between two host
// threads is benign; devices beyond the table run setup() on every call.)
#define MMSA_MAX_DEVICES 16
struct MmsaPerDevice { int cus[MMSA_MAX_DEVICES]; };
template <class F>
static inline int mmsa_per_device(MmsaPerDevice& st, F&& setup) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  if (dev < MMSA_MAX_DEVICES && st.cus[dev] > 0) return st.cus[dev];
  setup();
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev < MMSA_MAX_DEVICES) st.cus[dev] = n;
  return n;
}

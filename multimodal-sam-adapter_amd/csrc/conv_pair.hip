// Pair convolutions of the neck's gated MLP (AM:123-132): 3x3 conv with 2 input / 2 output channels per group, plain and
// fused with chunk + gelu gate + planes output.
//
// THIS FILE IS COMPILED WITH -fno-slp-vectorize (build.py).  With the SLP vectoriser hipcc (ROCm 7.2) packs the neighbouring
// scalar FMAs of these kernels into v_pk_fma_f32 / v_pk_mul_f32 with op_sel operands, and dwpair_gate_kernel compiled that
// way returned wrong values for the SECOND channel of 16 consecutive lanes (the upper half of a packed result) in 3-13 of 80
// forward passes -- but only while another stream kept the GPU busy (a second encoder instance, or the pipelined mode of
// backbone.py), never alone and never without the packing (0 of 240).  The inputs of the failing launches were bit-identical
// to those of the clean ones (tools/pipe_debug.py compares every intermediate); the same kernels built scalar are clean.
// tests/test_backbone_gpu.py::test_results_do_not_depend_on_concurrent_work keeps watch.
#include "common.h"

// 3x3 conv with 2 input / 2 output channels per group (gated-MLP dwconv, AM:123-124), float4 = two groups.
// weights [G][9][ci=2][co=2] -> one float4 per (group, tap).
// One thread per (pixel, 4 channels); blockIdx.y = image row (b*H + h), blockIdx.x walks the W * C/4 vectors of the row:
// 32-bit index arithmetic only (the first version's grid-stride loop spent most of its time in four 64-bit divisions per
// element: 473 us for a map that streams in ~100 us).
__global__ __launch_bounds__(256) void dwpair_nhwc_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long ldy, int B, int H, int W, int C) {
  const int c4n = C >> 2;
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());   // XCD-contiguous rows (common.h)
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)(wg_ / gridDim.x);
  const int idx = bx_ * 256 + threadIdx.x;
  if (idx >= W * c4n) return;
  const int ww = idx / c4n;
  const int c = (idx - ww * c4n) * 4;
  const int hh = by_ % H, b = by_ / H;
  const float* xb = x + (long)b * H * W * ldx + c;
  const float* wa = w + (long)(c >> 1) * 36;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = hh + kh - 1;
    if (ih < 0 || ih >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ww + kw - 1;
      if (iw < 0 || iw >= W) continue;
      const float4 v = *reinterpret_cast<const float4*>(xb + ((long)ih * W + iw) * ldx);
      const float4 fa = *reinterpret_cast<const float4*>(wa + (kh * 3 + kw) * 4);
      const float4 fb = *reinterpret_cast<const float4*>(wa + 36 + (kh * 3 + kw) * 4);
      acc.x += v.x * fa.x + v.y * fa.z;
      acc.y += v.x * fa.y + v.y * fa.w;
      acc.z += v.z * fb.x + v.w * fb.z;
      acc.w += v.z * fb.y + v.w * fb.w;
    }
  }
  *reinterpret_cast<float4*>(y + (((long)b * H + hh) * W + ww) * ldy + c) = acc;
}

// Gated depthwise-pair stage of the neck's Mlp (AM:127-132: dwconv 3x3 with 2 channels per group on [.., 2C], chunk,
// gelu(x1) * x2), fused: one thread computes 4 channels of x1 and the matching 4 of x2, gates them and writes fp32 and/or
// interleaved planes.  Weights TAP-major [9][G = C][ci = 2][co = 2]: consecutive lanes read consecutive 32 bytes (the
// group-major layout made every weight load touch 64 different lines and was ~4x the kernel's data traffic).
__global__ __launch_bounds__(256) void dwpair_gate_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long ldy, unsigned short* __restrict__ yp, long ldp,
                                                          int H, int W, int C) {
  const int c4n = C >> 2;
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());   // XCD-contiguous rows (common.h)
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)(wg_ / gridDim.x);
  const int idx = bx_ * 256 + threadIdx.x;
  if (idx >= W * c4n) return;
  const int ww = idx / c4n;
  const int c = (idx - ww * c4n) * 4;
  const int hh = by_ % H, b = by_ / H;
  const float* xb = x + (long)b * H * W * ldx + c;
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = hh + kh - 1;
    if (ih < 0 || ih >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ww + kw - 1;
      if (iw < 0 || iw >= W) continue;
      const float* xp = xb + ((long)ih * W + iw) * ldx;
      const float4 v1 = *reinterpret_cast<const float4*>(xp);
      const float4 v2 = *reinterpret_cast<const float4*>(xp + C);
      const float* wt = w + ((long)(kh * 3 + kw) * C) * 4;          // [tap][group][4], C groups
      const float4 fa = *reinterpret_cast<const float4*>(wt + (c >> 1) * 4);
      const float4 fb = *reinterpret_cast<const float4*>(wt + (c >> 1) * 4 + 4);
      const float4 ga = *reinterpret_cast<const float4*>(wt + ((C + c) >> 1) * 4);
      const float4 gb = *reinterpret_cast<const float4*>(wt + ((C + c) >> 1) * 4 + 4);
      a1.x += v1.x * fa.x + v1.y * fa.z;
      a1.y += v1.x * fa.y + v1.y * fa.w;
      a1.z += v1.z * fb.x + v1.w * fb.z;
      a1.w += v1.z * fb.y + v1.w * fb.w;
      a2.x += v2.x * ga.x + v2.y * ga.z;
      a2.y += v2.x * ga.y + v2.y * ga.w;
      a2.z += v2.z * gb.x + v2.w * gb.z;
      a2.w += v2.z * gb.y + v2.w * gb.w;
    }
  }
  float4 o;
  o.x = apply_act(a1.x, ACT_GELU) * a2.x;
  o.y = apply_act(a1.y, ACT_GELU) * a2.y;
  o.z = apply_act(a1.z, ACT_GELU) * a2.z;
  o.w = apply_act(a1.w, ACT_GELU) * a2.w;
  const long row = ((long)b * H + hh) * W + ww;
  if (y) *reinterpret_cast<float4*>(y + row * ldy + c) = o;
  if (yp) {
    uint2 h2, l2;
    split4(o, h2, l2);
    unsigned short* q_ = yp + row * ldp + ilv(c);
    *reinterpret_cast<uint2*>(q_) = h2;
    *reinterpret_cast<uint2*>(q_ + 32) = l2;
  }
}

// The same stage with a 1 x 4 pixel strip per thread (W % 4 == 0): the kernel above issues 54 16-byte loads per output vector -- 36 of them tap
// weights that every pixel of the row reads again -- and runs at the rate of the vector-memory pipe, not of HBM (164 MB read for 47 MB
// written, 2.1 TB/s: profiles/r03_hbm_kernels.json).  Here a thread keeps one tap's four weight vectors in registers for its four pixels
// and loads a kernel row's six input pixels once for the three taps of that row: 72 loads for four outputs (18 per output), every load of
// a kernel row unconditional (clamped position, zero mask: the same sums of the same terms) so that they are in flight together.
__global__ __launch_bounds__(256) void dwpair_gate4_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                           float* __restrict__ y, long ldy, unsigned short* __restrict__ yp, long ldp,
                                                           int H, int W, int C) {
  const int c4n = C >> 2;
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());   // XCD-contiguous rows (common.h)
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)(wg_ / gridDim.x);
  const int idx = bx_ * 256 + threadIdx.x;
  if (idx >= (W >> 2) * c4n) return;
  const int wq = idx / c4n;
  const int c = (idx - wq * c4n) * 4;
  const int w0 = wq * 4;
  const int hh = by_ % H, b = by_ / H;
  const float* xb = x + (long)b * H * W * ldx + c;
  float4 a1[4], a2[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) a1[p] = a2[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1   // (unrolled, all 72 loads are hoisted: 256 registers, one wave per SIMD)
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = hh + kh - 1;
    const bool rok = ih >= 0 && ih < H;
    const int ihc = min(max(ih, 0), H - 1);
    float4 v1[6], v2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int iw = w0 + i - 1;
      const unsigned m = (rok && iw >= 0 && iw < W) ? 0xffffffffu : 0u;
      const float* xp = xb + ((long)ihc * W + min(max(iw, 0), W - 1)) * ldx;
      float4 t1 = *reinterpret_cast<const float4*>(xp), t2 = *reinterpret_cast<const float4*>(xp + C);
      v1[i] = make_float4(__uint_as_float(__float_as_uint(t1.x) & m), __uint_as_float(__float_as_uint(t1.y) & m), __uint_as_float(__float_as_uint(t1.z) & m), __uint_as_float(__float_as_uint(t1.w) & m));
      v2[i] = make_float4(__uint_as_float(__float_as_uint(t2.x) & m), __uint_as_float(__float_as_uint(t2.y) & m), __uint_as_float(__float_as_uint(t2.z) & m), __uint_as_float(__float_as_uint(t2.w) & m));
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float* wt = w + ((long)(kh * 3 + kw) * C) * 4;          // [tap][group][4], C groups
      const float4 fa = *reinterpret_cast<const float4*>(wt + (c >> 1) * 4);
      const float4 fb = *reinterpret_cast<const float4*>(wt + (c >> 1) * 4 + 4);
      const float4 ga = *reinterpret_cast<const float4*>(wt + ((C + c) >> 1) * 4);
      const float4 gb = *reinterpret_cast<const float4*>(wt + ((C + c) >> 1) * 4 + 4);
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // the per-pixel expressions of dwpair_gate_kernel, term for term and in its tap order (a masked tap adds +0)
        const float4 u1 = v1[p + kw], u2 = v2[p + kw];
        a1[p].x += u1.x * fa.x + u1.y * fa.z;
        a1[p].y += u1.x * fa.y + u1.y * fa.w;
        a1[p].z += u1.z * fb.x + u1.w * fb.z;
        a1[p].w += u1.z * fb.y + u1.w * fb.w;
        a2[p].x += u2.x * ga.x + u2.y * ga.z;
        a2[p].y += u2.x * ga.y + u2.y * ga.w;
        a2[p].z += u2.z * gb.x + u2.w * gb.z;
        a2[p].w += u2.z * gb.y + u2.w * gb.w;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float4 o;
    o.x = apply_act(a1[p].x, ACT_GELU) * a2[p].x;
    o.y = apply_act(a1[p].y, ACT_GELU) * a2[p].y;
    o.z = apply_act(a1[p].z, ACT_GELU) * a2[p].z;
    o.w = apply_act(a1[p].w, ACT_GELU) * a2[p].w;
    const long row = ((long)b * H + hh) * W + w0 + p;
    if (y) *reinterpret_cast<float4*>(y + row * ldy + c) = o;
    if (yp) {
      uint2 h2, l2;
      split4(o, h2, l2);
      unsigned short* q_ = yp + row * ldp + ilv(c);
      *reinterpret_cast<uint2*>(q_) = h2;
      *reinterpret_cast<uint2*>(q_ + 32) = l2;
    }
  }
}

int mmsa_dwpair_nhwc_launch(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int C, hipStream_t stream) {
  hipLaunchKernelGGL(dwpair_nhwc_kernel, dim3(cdiv((long)W * (C / 4), 256), B * H), dim3(256), 0, stream, x, ldx, w, y, ldy, B, H, W, C);
  MMSA_CHECK_LAUNCH("gconv_nhwc(pair)");
  return MMSA_OK;
}

extern "C" int mmsa_dwpair_gate(const float* x, long ldx, const float* w, float* y, long ldy, unsigned short* yp, long ldp,
                                int B, int H, int W, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(x && w && (y || yp) && B > 0 && H > 0 && W > 0 && C > 0, "dwpair_gate: bad args");
  MMSA_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && ldx >= 2L * C && (!y || ((ldy & 3) == 0 && ldy >= C)), "dwpair_gate: C/ld must be multiples of 4");
  MMSA_CHECK_ARG(!yp || ((((uintptr_t)yp) & 127) == 0 && (ldp & 63) == 0 && ldp >= 2L * ((C + 31) / 32 * 32)), "dwpair_gate: bad output planes");
  MMSA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w)) & 15) == 0, "dwpair_gate: pointers must be 16-byte aligned");
  MMSA_CHECK_ARG((long)B * H <= 65535, "dwpair_gate: B*H too large for the launch grid");
  if ((W & 3) == 0 && MMSA_KNOB("MMSA_DWPAIR_STRIP", 1) != 0)
    hipLaunchKernelGGL(dwpair_gate4_kernel, dim3(cdiv((long)(W >> 2) * (C >> 2), 256), B * H), dim3(256), 0, stream, x, ldx, w, y, ldy, yp, ldp, H, W, C);
  else
    hipLaunchKernelGGL(dwpair_gate_kernel, dim3(cdiv((long)W * (C >> 2), 256), B * H), dim3(256), 0, stream, x, ldx, w, y, ldy, yp, ldp, H, W, C);
  MMSA_CHECK_LAUNCH("dwpair_gate");
  return MMSA_OK;
}


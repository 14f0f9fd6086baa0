// Multi-scale deformable attention forward (sampling gather) for gfx950.
//
// Replaces the reference's only native op: pybind `ms_deform_attn_forward`
// (segmentation/ops/src/vision.cpp:14 -> ms_deform_attn.h:20-39 -> cuda/ms_deform_attn_cuda.cu:20-80 ->
// cuda/ms_deform_im2col_cuda.cuh:237-299 kernel, :33-84 bilinear).  Semantics kept exactly:
//   pixel = loc*(W,H) - 0.5 ; sample contributes only if -1 < h < H and -1 < w < W ; 4-tap bilinear with
//   zero padding per tap ; out[b,q,m,:] = sum_l sum_p weight * bilinear(value_l).
//
// MI355X mapping (not the reference's one-thread-per-scalar): the value map is read as 16-byte channel
// vectors.  With D channels per head, D/4 lanes cover one (query, head); a 64-lane wavefront covers
// 256/D (query, head) pairs, and for every bilinear corner the lanes of a pair read one contiguous
// D*4-byte segment (128 B at D=32) -> fully coalesced gathers that stay L2/MALL resident
// (value maps are <= 44 MB fp32).  Sampling locations / weights are read once per pair (broadcast loads).
#include "common.h"

template <bool FUSED>
__global__ __launch_bounds__(256) void msda_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc,   // !FUSED: [N,Lq,M,L,P,2]
    const float* __restrict__ aw,    // !FUSED: [N,Lq,M,L,P] (already normalised)
    const float* __restrict__ raw, long ldraw,  // FUSED: [N*Lq, M*L*P*2 (offsets) + M*L*P (logits)]
    const float* __restrict__ ref,   // FUSED: [Lq,2] reference points (x,y) in [0,1]
    float* __restrict__ out, long ldo,
    unsigned short* __restrict__ op, long ldop,
    int N, int S, int M, int D, int L, int Lq, int P) {
  const int d4 = D >> 2;                        // lanes per (q, m) pair
  // 32-bit index arithmetic (N*Lq*M*d4 < 2^32, checked by the launcher): the four 64-bit div/mod of the first version
  // were a quarter of the kernel's instructions
  const unsigned pair = (blockIdx.x * blockDim.x + threadIdx.x) / (unsigned)d4;
  const int c = ((threadIdx.x) % d4) * 4;       // blockDim.x is a multiple of d4
  const unsigned npairs = (unsigned)N * (unsigned)Lq * (unsigned)M;
  if (pair >= npairs) return;
  const unsigned bq = pair / (unsigned)M;
  const int m = (int)(pair - bq * (unsigned)M);
  const int b = (int)(bq / (unsigned)Lq);
  const int q = (int)(bq - (unsigned)b * (unsigned)Lq);
  const int LP = L * P;

  const float* offp = nullptr;
  const float* logp = nullptr;
  float mx = 0.f, inv = 1.f, rx = 0.f, ry = 0.f;
  if (FUSED) {
    offp = raw + bq * ldraw + (long)m * LP * 2;
    logp = raw + bq * ldraw + (long)M * LP * 2 + (long)m * LP;
    mx = -INFINITY;
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, logp[i]);
    float s = 0.f;
    for (int i = 0; i < LP; ++i) s += expf(logp[i] - mx);
    inv = 1.0f / s;
    rx = ref[2 * q];
    ry = ref[2 * q + 1];
  }
  const long lw = pair * LP;  // index into aw; loc index = 2*lw
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long vstride = (long)M * D;  // floats per spatial position
  const float* vb = value + (long)b * S * vstride + (long)m * D + c;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float* vl = vb + (long)lsi[l] * vstride;
    for (int p = 0; p < P; ++p) {
      float lx, ly, wgt;
      if (FUSED) {
        // OPS/modules/ms_deform_attn.py:105-119: loc = ref + off / (W_l, H_l); weights = softmax over L*P
        lx = rx + offp[(l * P + p) * 2] / (float)W;
        ly = ry + offp[(l * P + p) * 2 + 1] / (float)H;
        wgt = expf(logp[l * P + p] - mx) * inv;
      } else {
        lx = loc[(lw + l * P + p) * 2];
        ly = loc[(lw + l * P + p) * 2 + 1];
        wgt = aw[lw + l * P + p];
      }
      const float h_im = ly * H - 0.5f;
      const float w_im = lx * W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lwf = w_im - w_low;
        const float hh = 1.f - lh, hw = 1.f - lwf;
        const int h_high = h_low + 1, w_high = w_low + 1;
        float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1, v3 = v1, v4 = v1;
        if (h_low >= 0 && w_low >= 0) v1 = *reinterpret_cast<const float4*>(vl + ((long)h_low * W + w_low) * vstride);
        if (h_low >= 0 && w_high <= W - 1) v2 = *reinterpret_cast<const float4*>(vl + ((long)h_low * W + w_high) * vstride);
        if (h_high <= H - 1 && w_low >= 0) v3 = *reinterpret_cast<const float4*>(vl + ((long)h_high * W + w_low) * vstride);
        if (h_high <= H - 1 && w_high <= W - 1) v4 = *reinterpret_cast<const float4*>(vl + ((long)h_high * W + w_high) * vstride);
        const float w1 = hh * hw, w2 = hh * lwf, w3 = lh * hw, w4 = lh * lwf;
        acc.x += (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * wgt;
        acc.y += (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * wgt;
        acc.z += (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * wgt;
        acc.w += (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * wgt;
      }
    }
  }
  if (out) *reinterpret_cast<float4*>(out + bq * ldo + (long)m * D + c) = acc;
  if (op) {  // ilv planes
    uint2 hh, ll;
    split4(acc, hh, ll);
    unsigned short* q_ = op + bq * ldop + ilv(m * D + c);
    *reinterpret_cast<uint2*>(q_) = hh;
    *reinterpret_cast<uint2*>(q_ + 32) = ll;
  }
}


// Generic scalar path (any D): one lane per output element.  Used for head widths the vector path does
// not cover (e.g. the reference's own known-answer fixture OPS/test.py:16-33 has D = 2).
__global__ __launch_bounds__(256) void msda_scalar_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc, const float* __restrict__ aw, float* __restrict__ out,
    int N, int S, int M, int D, int L, int Lq, int P) {
  const long total = (long)N * Lq * M * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    const long pair = idx / D;
    const int m = (int)(pair % M);
    const int b = (int)(pair / M / Lq);
    const long lw = pair * L * P;
    const long vstride = (long)M * D;
    const float* vb = value + (long)b * S * vstride + (long)m * D + c;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float* vl = vb + (long)lsi[l] * vstride;
      for (int p = 0; p < P; ++p) {
        const float lx = loc[(lw + l * P + p) * 2], ly = loc[(lw + l * P + p) * 2 + 1];
        const float wgt = aw[lw + l * P + p];
        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - h_low, lwf = w_im - w_low, hh = 1.f - lh, hw = 1.f - lwf;
          const int h_high = h_low + 1, w_high = w_low + 1;
          float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
          if (h_low >= 0 && w_low >= 0) v1 = vl[((long)h_low * W + w_low) * vstride];
          if (h_low >= 0 && w_high <= W - 1) v2 = vl[((long)h_low * W + w_high) * vstride];
          if (h_high <= H - 1 && w_low >= 0) v3 = vl[((long)h_high * W + w_low) * vstride];
          if (h_high <= H - 1 && w_high <= W - 1) v4 = vl[((long)h_high * W + w_high) * vstride];
          acc += (hh * hw * v1 + hh * lwf * v2 + lh * hw * v3 + lh * lwf * v4) * wgt;
        }
      }
    }
    out[idx] = acc;
  }
}

static int msda_check(int N, int S, int M, int D, int L, int Lq, int P, const char* name) {
  MMSA_CHECK_ARG(N > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, "%s: bad shape", name);
  return MMSA_OK;
}

// Drop-in for ms_deform_attn_forward (vision.cpp:14).  All pointers are DEVICE pointers, tensors
// contiguous with the reference's layouts; `out` is [batch, num_query, num_heads*channels] and is fully
// overwritten (the reference zero-fills then writes, ms_deform_attn_cuda.cu:54).  `im2col_step` is
// accepted and validated like the reference (batch %% min(batch, im2col_step) == 0, :52) but the launch
// covers the whole batch at once.
extern "C" int mmsa_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes,
                                           const int64_t* level_start_index, const float* sampling_loc,
                                           const float* attn_weight, float* out, int batch, int spatial_size,
                                           int num_heads, int channels, int num_levels, int num_query,
                                           int num_point, int im2col_step, hipStream_t stream) {
  MMSA_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out, "ms_deform_attn_forward: null pointer");
  int rc = msda_check(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, "ms_deform_attn_forward");
  if (rc) return rc;
  const int step = batch < im2col_step ? batch : im2col_step;
  MMSA_CHECK_ARG(step > 0 && batch % step == 0, "batch(%d) must divide im2col_step(%d)", batch, step);
  const bool vec = (channels & 3) == 0 && channels <= 1024 && 256 % (channels >> 2) == 0 &&
                   ((((uintptr_t)value) | ((uintptr_t)out)) & 15) == 0;
  if (!vec) {
    const long total = (long)batch * num_query * num_heads * channels;
    int blocks = cdiv(total, 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(msda_scalar_kernel, dim3(blocks), dim3(256), 0, stream, value, spatial_shapes, level_start_index,
                       sampling_loc, attn_weight, out, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point);
    MMSA_CHECK_LAUNCH("ms_deform_attn_forward(scalar)");
    return MMSA_OK;
  }
  const long threads = (long)batch * num_query * num_heads * (channels >> 2);
  MMSA_CHECK_ARG(threads < (1L << 31), "ms_deform_attn_forward: problem too large for the 32-bit index arithmetic");
  hipLaunchKernelGGL(msda_kernel<false>, dim3(cdiv(threads, 256)), dim3(256), 0, stream, value, spatial_shapes,
                     level_start_index, sampling_loc, attn_weight, nullptr, 0L, nullptr, out,
                     (long)num_heads * channels, nullptr, 0L, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point);
  MMSA_CHECK_LAUNCH("ms_deform_attn_forward");
  return MMSA_OK;
}

// Fused hot-path variant: consumes the raw output of the (concatenated) sampling_offsets|attention_weights
// projection and does the softmax over L*P and the location arithmetic of
// OPS/modules/ms_deform_attn.py:105-119 in-kernel (reference points are per query, broadcast over levels,
// as produced by AM:397-431).
extern "C" int mmsa_msda_fused(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* raw, long ldraw, const float* ref_points, float* out, long ldo,
                               unsigned short* out_p, long ldop,
                               int batch, int spatial_size, int num_heads, int channels, int num_levels,
                               int num_query, int num_point, hipStream_t stream) {
  MMSA_CHECK_ARG(value && spatial_shapes && level_start_index && raw && ref_points && (out || out_p), "msda_fused: null pointer");
  MMSA_CHECK_ARG(!out_p || (ldop >= 2L * num_heads * channels && (ldop & 63) == 0 && (((uintptr_t)out_p) & 127) == 0),
                 "msda_fused: bad output planes");
  int rc = msda_check(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, "msda_fused");
  if (rc) return rc;
  MMSA_CHECK_ARG(ldraw >= (long)num_heads * num_levels * num_point * 3, "msda_fused: ldraw too small");
  MMSA_CHECK_ARG(!out || (ldo >= (long)num_heads * channels && (ldo & 3) == 0), "msda_fused: bad ldo");
  MMSA_CHECK_ARG((channels & 3) == 0 && channels <= 1024 && 256 % (channels >> 2) == 0, "msda_fused: channels per head D=%d must be a multiple of 4 with 256 %% (D/4) == 0", channels);
  MMSA_CHECK_ARG(((((uintptr_t)value) | ((uintptr_t)out)) & 15) == 0, "msda_fused: value/out must be 16-byte aligned");
  const long threads = (long)batch * num_query * num_heads * (channels >> 2);
  MMSA_CHECK_ARG(threads < (1L << 31), "msda_fused: problem too large for the 32-bit index arithmetic");
  hipLaunchKernelGGL(msda_kernel<true>, dim3(cdiv(threads, 256)), dim3(256), 0, stream, value, spatial_shapes,
                     level_start_index, nullptr, nullptr, raw, ldraw, ref_points, out, ldo, out_p, ldop, batch, spatial_size,
                     num_heads, channels, num_levels, num_query, num_point);
  MMSA_CHECK_LAUNCH("msda_fused");
  return MMSA_OK;
}

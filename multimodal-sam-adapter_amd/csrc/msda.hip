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
#include <hip/hip_fp16.h>

template <bool FUSED>
__global__ __launch_bounds__(256) void msda_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ loc,   // !FUSED: [N,Lq,M,L,P,2]
    const float* __restrict__ aw,    // !FUSED: [N,Lq,M,L,P] (already normalised)
    const float* __restrict__ raw, long ldraw,  // FUSED: [N*Lq, M*L*P*2 (offsets) + M*L*P (logits)]
    const float* __restrict__ ref,   // FUSED: [Lq,2] reference points (x,y) in [0,1]
    float* __restrict__ out, long ldo,
    unsigned short* __restrict__ op, long ldop, int op_fmt,
    int N, int S, int M, int D, int L, int Lq, int P, float* __restrict__ clamp_max) {
  const int d4 = D >> 2;                        // lanes per (q, m) pair
  const int op_kpad = MMSA_PAD64(M * D);        // h8c output planes: fp16 values per row
  // 32-bit index arithmetic (N*Lq*M*d4 < 2^32, checked by the launcher): the four 64-bit div/mod of the first version
  // were a quarter of the kernel's instructions
  // XCD-contiguous query order (common.h): a workgroup = a few queries x all heads; queries that are neighbours in the map sample the same patch of
  // the value map, so every XCD walks one contiguous eighth of the queries (in dispatch order each XCD's L2 pulled the whole value map)
  const unsigned pair = (mmsa_xcd_order(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x) / (unsigned)d4;
  const int c = ((threadIdx.x) % d4) * 4;       // blockDim.x is a multiple of d4 (msda_block)
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
  if (op) {   // operand planes for output_proj, either format
    float cw_ = 0.f;   // clamp watch (common.h): a sample is a convex combination of value rows, which arrive as unbounded fp32
    clamp_see(cw_, acc);
    clamp_report(clamp_max, cw_, mmsa_clamp_limit(op_fmt));
    // D % 8 == 0: threads 2j / 2j+1 of a (query, head) group hold 8 consecutive channels: whole-line stores through the lane-pair exchange
    if ((D & 7) == 0) store_planes8_pair_any<1>(op, ldop, bq, op_kpad, m * D + (c & ~7), acc, op_fmt, (c >> 2) & 1, true);
    else store_planes4_any(op, ldop, bq, op_kpad, m * D + c, acc, op_fmt);
  }
}


// ---- the gather on fp16 values (round 6; opt-in through the model's `msda_value` attribute).  The gather is bound by the bytes it pulls through the CU's
// vector-memory pipe (64 B/clk: 128 B per corner, head and sample with fp32 values; LAB_NOTES 4.3 "(7)").  Here `value` arrives as H8 ACTIVATION PLANES
// (common.h: per row and 32-channel block one 128-byte line = 32 fp16 hi values + four 16-byte chunks {8 e5m2(lo * 2^11) bytes | 8 q(hi) bytes}), written
// by the value projection's own epilogue (cp_fmt = MMSA_FMT_H8, no fp32 output).  A lane owns 8 channels of a (query, head) pair and reads per corner
//   LO = false: the 8 fp16 hi values (16 B): half the bytes, the value rounded to 11 significant bits (the oracle study: tools/msda_value_f16_study.py);
//   LO = true : + the 8 lo bytes (8 B): 3/4 of the bytes, value = hi + lo / 2^11 to ~14 bits.
// Same sampling arithmetic, same fused prologue (softmax over L*P, loc = ref + off / (W, H)) as msda_kernel<true>.
__device__ __forceinline__ void msda_decode_hi8(const uint4 u, float* f) {
  const __half2* h = reinterpret_cast<const __half2*>(&u);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float2 t = __half22float2(h[i]); f[2 * i] = t.x; f[2 * i + 1] = t.y; }
}
__device__ __forceinline__ void msda_decode_lo8(const uint2 b, float* f) {   // e5m2 = the top byte of an fp16 value
  const unsigned w[4] = {__builtin_amdgcn_perm(0u, b.x, 0x010c000cu), __builtin_amdgcn_perm(0u, b.x, 0x030c020cu),
                         __builtin_amdgcn_perm(0u, b.y, 0x010c000cu), __builtin_amdgcn_perm(0u, b.y, 0x030c020cu)};
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float2 t = __half22float2(*reinterpret_cast<const __half2*>(&w[i])); f[2 * i] = t.x; f[2 * i + 1] = t.y; }
}
template <bool LO>
__global__ __launch_bounds__(256) void msda_planes_kernel(
    const unsigned short* __restrict__ vp, long ldv, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ raw, long ldraw, const float* __restrict__ ref, float* __restrict__ out, long ldo,
    unsigned short* __restrict__ op, long ldop, int op_fmt, int N, int S, int M, int D, int L, int Lq, int P, float* __restrict__ clamp_max) {
  const int d8 = D >> 3;                        // lanes per (q, m) pair
  const int op_kpad = MMSA_PAD64(M * D);
  const unsigned pair = (mmsa_xcd_order(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x) / (unsigned)d8;
  const int c = ((threadIdx.x) % d8) * 8;       // blockDim.x is a multiple of d8
  const unsigned npairs = (unsigned)N * (unsigned)Lq * (unsigned)M;
  if (pair >= npairs) return;
  const unsigned bq = pair / (unsigned)M;
  const int m = (int)(pair - bq * (unsigned)M);
  const int b = (int)(bq / (unsigned)Lq);
  const int q = (int)(bq - (unsigned)b * (unsigned)Lq);
  const int LP = L * P;
  const float* offp = raw + bq * ldraw + (long)m * LP * 2;
  const float* logp = raw + bq * ldraw + (long)M * LP * 2 + (long)m * LP;
  float mx = -INFINITY;
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, logp[i]);
  float ssum = 0.f;
  for (int i = 0; i < LP; ++i) ssum += expf(logp[i] - mx);
  const float inv = 1.0f / ssum;
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const int ch = m * D + c;                     // first of this lane's 8 channels: line ch / 32 of the row, halves (ch & 31) .. + 7 of its hi part
  const unsigned char* vb = reinterpret_cast<const unsigned char*>(vp) + (long)b * S * ldv * 2 + (long)(ch >> 5) * 128;
  const int hi_off = (ch & 31) * 2, lo_off = 64 + ((ch & 31) >> 3) * 16;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const unsigned char* vl = vb + (long)lsi[l] * ldv * 2;
    for (int p = 0; p < P; ++p) {
      const float lx = rx + offp[(l * P + p) * 2] / (float)W;
      const float ly = ry + offp[(l * P + p) * 2 + 1] / (float)H;
      const float wgt = expf(logp[l * P + p] - mx) * inv;
      const float h_im = ly * H - 0.5f;
      const float w_im = lx * W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lwf = w_im - w_low;
        const float hh = 1.f - lh, hw = 1.f - lwf;
        const int h_high = h_low + 1, w_high = w_low + 1;
        const bool ok[4] = {h_low >= 0 && w_low >= 0, h_low >= 0 && w_high <= W - 1, h_high <= H - 1 && w_low >= 0, h_high <= H - 1 && w_high <= W - 1};
        const long pos[4] = {(long)h_low * W + w_low, (long)h_low * W + w_high, (long)h_high * W + w_low, (long)h_high * W + w_high};
        const float cw[4] = {hh * hw * wgt, hh * lwf * wgt, lh * hw * wgt, lh * lwf * wgt};
        uint4 uh[4];
        uint2 ul[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {           // all corner loads of the sample in flight together
          uh[k] = make_uint4(0u, 0u, 0u, 0u);
          ul[k] = make_uint2(0u, 0u);
          if (ok[k]) {
            const unsigned char* rp = vl + pos[k] * ldv * 2;
            uh[k] = *reinterpret_cast<const uint4*>(rp + hi_off);
            if (LO) ul[k] = *reinterpret_cast<const uint2*>(rp + lo_off);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float f[8];
          msda_decode_hi8(uh[k], f);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = fmaf(cw[k], f[i], acc[i]);
          if (LO) {
            float g8[8];
            msda_decode_lo8(ul[k], g8);
            const float cl = cw[k] * (1.0f / 2048.0f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(cl, g8[i], acc[i]);
          }
        }
      }
    }
  }
  const float4 a0 = make_float4(acc[0], acc[1], acc[2], acc[3]), a1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
  if (out) {
    *reinterpret_cast<float4*>(out + bq * ldo + ch) = a0;
    *reinterpret_cast<float4*>(out + bq * ldo + ch + 4) = a1;
  }
  if (op) {
    float cw_ = 0.f;
    clamp_see(cw_, a0);
    clamp_see(cw_, a1);
    clamp_report(clamp_max, cw_, mmsa_clamp_limit(op_fmt));
    store_planes4_any(op, ldop, bq, op_kpad, ch, a0, op_fmt);
    store_planes4_any(op, ldop, bq, op_kpad, ch + 4, a1, op_fmt);
  }
}


// Generic path (any D, any dtype): one lane per output element, arithmetic in AT (float for f32 / f16 storage, double for f64).
// Used for head widths the vector path does not cover (e.g. the reference's own known-answer fixture OPS/test.py:16-33 has
// D = 2) and for the f16 / f64 instantiations of the reference's dispatch (ms_deform_attn_cuda.cu:64).
template <typename T> struct msda_acc { typedef float type; };
template <> struct msda_acc<double> { typedef double type; };

template <typename T>
__global__ __launch_bounds__(256) void msda_scalar_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const T* __restrict__ loc, const T* __restrict__ aw, T* __restrict__ out,
    int N, int S, int M, int D, int L, int Lq, int P) {
  typedef typename msda_acc<T>::type AT;
  const long total = (long)N * Lq * M * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    const long pair = idx / D;
    const int m = (int)(pair % M);
    const int b = (int)(pair / M / Lq);
    const long lw = pair * L * P;
    const long vstride = (long)M * D;
    const T* vb = value + (long)b * S * vstride + (long)m * D + c;
    AT acc = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* vl = vb + (long)lsi[l] * vstride;
      for (int p = 0; p < P; ++p) {
        const AT lx = (AT)loc[(lw + l * P + p) * 2], ly = (AT)loc[(lw + l * P + p) * 2 + 1];
        const AT wgt = (AT)aw[lw + l * P + p];
        const AT h_im = ly * H - (AT)0.5, w_im = lx * W - (AT)0.5;
        if (h_im > -1 && w_im > -1 && h_im < (AT)H && w_im < (AT)W) {
          const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
          const AT lh = h_im - h_low, lwf = w_im - w_low, hh = 1 - lh, hw = 1 - lwf;
          const int h_high = h_low + 1, w_high = w_low + 1;
          AT v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          if (h_low >= 0 && w_low >= 0) v1 = (AT)vl[((long)h_low * W + w_low) * vstride];
          if (h_low >= 0 && w_high <= W - 1) v2 = (AT)vl[((long)h_low * W + w_high) * vstride];
          if (h_high <= H - 1 && w_low >= 0) v3 = (AT)vl[((long)h_high * W + w_low) * vstride];
          if (h_high <= H - 1 && w_high <= W - 1) v4 = (AT)vl[((long)h_high * W + w_high) * vstride];
          acc += (hh * hw * v1 + hh * lwf * v2 + lh * hw * v3 + lh * lwf * v4) * wgt;
        }
      }
    }
    out[idx] = (T)acc;
  }
}

// ---- backward (ms_deform_attn_backward, vision.cpp:15 -> ms_deform_attn_cuda.cu:83-151 -> ms_deform_im2col_cuda.cuh:301-920 kernels
// with the per-sample arithmetic of :86-160): for every (b, q, m, c) and sample (l, p), with g = grad_output[b,q,m,c],
//   grad_value[corner]      += w_corner * g * attn_weight                      (scatter: atomics, like the reference)
//   grad_attn_weight[b,q,m,l,p] = sum_c g * bilinear(value)
//   grad_sampling_loc[..., x] = sum_c W_l * (d bilinear / d w) * g * attn_weight,  [..., y] likewise with H_l
// One lane per (b, q, m, c): the lanes of a (q, m) pair hold the channel partials of the two per-sample gradients; when D is a
// power of two <= 64 they are summed with xor-shuffles inside the lane group and written once (no atomics, deterministic, where
// the reference block-reduces through shared memory); any other D adds the partials with float atomics into zeroed outputs.
__device__ __forceinline__ void msda_atomic_add(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void msda_atomic_add(double* p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void msda_atomic_add(__half* p, float v) {
  // 16-bit float atomics are packed pairs on gfx950 (global_atomic_pk_add_f16): add (v, 0) or (0, v) to the aligned pair
  const uintptr_t a = (uintptr_t)p;
  __half2* p2 = reinterpret_cast<__half2*>(a & ~(uintptr_t)3);
  const __half hv = __float2half(v), z = __float2half(0.f);
  unsafeAtomicAdd(p2, (a & 2) ? __halves2half2(z, hv) : __halves2half2(hv, z));
}

template <typename T, bool SHFL>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const T* __restrict__ loc, const T* __restrict__ aw, const T* __restrict__ gout,
    T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gaw,
    int N, int S, int M, int D, int L, int Lq, int P) {
  typedef typename msda_acc<T>::type AT;
  const long total = (long)N * Lq * M * D;
  const long idx0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = idx0 < total;            // inactive lanes still take part in the shuffles with zero partials
  const long idx = active ? idx0 : total - 1;
  const int c = (int)(idx % D);
  const long pair = idx / D;
  const int m = (int)(pair % M);
  const int b = (int)(pair / M / Lq);
  const long lw = pair * L * P;
  const long vstride = (long)M * D;
  const long voff = (long)b * S * vstride + (long)m * D + c;
  const AT g = active ? (AT)gout[idx] : (AT)0;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long lbase = voff + (long)lsi[l] * vstride;
    for (int p = 0; p < P; ++p) {
      const long si = lw + l * P + p;
      const AT lx = (AT)loc[si * 2], ly = (AT)loc[si * 2 + 1];
      const AT wgt = (AT)aw[si];
      const AT h_im = ly * H - (AT)0.5, w_im = lx * W - (AT)0.5;
      AT g_w = 0, g_x = 0, g_y = 0;
      if (h_im > -1 && w_im > -1 && h_im < (AT)H && w_im < (AT)W) {
        const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
        const AT lh = h_im - h_low, lwf = w_im - w_low, hh = 1 - lh, hw = 1 - lwf;
        const int h_high = h_low + 1, w_high = w_low + 1;
        const AT tg = g * wgt;
        AT val = 0, dh = 0, dw = 0;
        if (h_low >= 0 && w_low >= 0) {
          const long o = lbase + ((long)h_low * W + w_low) * vstride;
          const AT v = (AT)value[o];
          val += hh * hw * v; dh -= hw * v; dw -= hh * v;
          if (active) msda_atomic_add(gvalue + o, hh * hw * tg);
        }
        if (h_low >= 0 && w_high <= W - 1) {
          const long o = lbase + ((long)h_low * W + w_high) * vstride;
          const AT v = (AT)value[o];
          val += hh * lwf * v; dh -= lwf * v; dw += hh * v;
          if (active) msda_atomic_add(gvalue + o, hh * lwf * tg);
        }
        if (h_high <= H - 1 && w_low >= 0) {
          const long o = lbase + ((long)h_high * W + w_low) * vstride;
          const AT v = (AT)value[o];
          val += lh * hw * v; dh += hw * v; dw -= lh * v;
          if (active) msda_atomic_add(gvalue + o, lh * hw * tg);
        }
        if (h_high <= H - 1 && w_high <= W - 1) {
          const long o = lbase + ((long)h_high * W + w_high) * vstride;
          const AT v = (AT)value[o];
          val += lh * lwf * v; dh += lwf * v; dw += lh * v;
          if (active) msda_atomic_add(gvalue + o, lh * lwf * tg);
        }
        g_w = g * val;
        g_x = (AT)W * dw * tg;
        g_y = (AT)H * dh * tg;
      }
      if constexpr (SHFL) {
        for (int o = D >> 1; o > 0; o >>= 1) {
          g_w += __shfl_xor(g_w, o, 64);
          g_x += __shfl_xor(g_x, o, 64);
          g_y += __shfl_xor(g_y, o, 64);
        }
        if (active && c == 0) {
          gaw[si] = (T)g_w;
          gloc[si * 2] = (T)g_x;
          gloc[si * 2 + 1] = (T)g_y;
        }
      } else if (active) {
        msda_atomic_add(gaw + si, g_w);
        msda_atomic_add(gloc + si * 2, g_x);
        msda_atomic_add(gloc + si * 2 + 1, g_y);
      }
    }
  }
}

static int msda_check(int N, int S, int M, int D, int L, int Lq, int P, const char* name) {
  MMSA_CHECK_ARG(N > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, "%s: bad shape", name);
  return MMSA_OK;
}
// workgroup size of the vector kernel: the largest multiple of D/4 lanes that fits 256 (D = 40 -> 250 threads)
static int msda_block(int channels) { const int d4 = channels >> 2; return (256 / d4) * d4; }
static size_t msda_elt(int dtype) { return dtype == MMSA_DT_F16 ? 2 : dtype == MMSA_DT_F64 ? 8 : 4; }

// Drop-in for ms_deform_attn_forward (vision.cpp:14).  All pointers are DEVICE pointers, tensors
// contiguous with the reference's layouts; `out` is [batch, num_query, num_heads*channels] and is fully
// overwritten (the reference zero-fills then writes, ms_deform_attn_cuda.cu:54).  `im2col_step` is
// accepted and validated like the reference (batch %% min(batch, im2col_step) == 0, :52) but the launch
// covers the whole batch at once.  dtype: the scalar type of value / sampling_loc / attn_weight / out, the reference's
// AT_DISPATCH_FLOATING_TYPES_AND_HALF (ms_deform_attn_cuda.cu:64); f16 computes in fp32 and rounds the result once.
extern "C" int mmsa_ms_deform_attn_forward(const void* value, const int64_t* spatial_shapes,
                                           const int64_t* level_start_index, const void* sampling_loc,
                                           const void* attn_weight, void* out, int batch, int spatial_size,
                                           int num_heads, int channels, int num_levels, int num_query,
                                           int num_point, int im2col_step, int dtype, hipStream_t stream) {
  MMSA_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out, "ms_deform_attn_forward: null pointer");
  MMSA_CHECK_ARG(dtype == MMSA_DT_F32 || dtype == MMSA_DT_F16 || dtype == MMSA_DT_F64, "ms_deform_attn_forward: dtype %d not supported (0 f32, 1 f16, 2 f64)", dtype);
  int rc = msda_check(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, "ms_deform_attn_forward");
  if (rc) return rc;
  const int step = batch < im2col_step ? batch : im2col_step;
  MMSA_CHECK_ARG(step > 0 && batch % step == 0, "batch(%d) must divide im2col_step(%d)", batch, step);
  const bool vec = dtype == MMSA_DT_F32 && (channels & 3) == 0 && channels <= 1024 &&
                   ((((uintptr_t)value) | ((uintptr_t)out)) & 15) == 0;
  if (!vec) {
    const long total = (long)batch * num_query * num_heads * channels;
    int blocks = cdiv(total, 256);
    if (blocks > 65535) blocks = 65535;
#define MSDA_SCALAR(T_)                                                                                                   \
    hipLaunchKernelGGL(msda_scalar_kernel<T_>, dim3(blocks), dim3(256), 0, stream, (const T_*)value, spatial_shapes,      \
                       level_start_index, (const T_*)sampling_loc, (const T_*)attn_weight, (T_*)out, batch, spatial_size,  \
                       num_heads, channels, num_levels, num_query, num_point)
    if (dtype == MMSA_DT_F16) MSDA_SCALAR(__half);
    else if (dtype == MMSA_DT_F64) MSDA_SCALAR(double);
    else MSDA_SCALAR(float);
#undef MSDA_SCALAR
    MMSA_CHECK_LAUNCH("ms_deform_attn_forward(scalar)");
    return MMSA_OK;
  }
  const long threads = (long)batch * num_query * num_heads * (channels >> 2);
  MMSA_CHECK_ARG(threads < (1L << 31), "ms_deform_attn_forward: problem too large for the 32-bit index arithmetic");
  const int bs = msda_block(channels);
  hipLaunchKernelGGL(msda_kernel<false>, dim3(cdiv(threads, bs)), dim3(bs), 0, stream, (const float*)value, spatial_shapes,
                     level_start_index, (const float*)sampling_loc, (const float*)attn_weight, nullptr, 0L, nullptr, (float*)out,
                     (long)num_heads * channels, nullptr, 0L, MMSA_FMT_B3, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, nullptr);
  MMSA_CHECK_LAUNCH("ms_deform_attn_forward");
  return MMSA_OK;
}

// Drop-in for ms_deform_attn_backward (vision.cpp:15 -> ms_deform_attn_cuda.cu:83-151).  grad_output [N,Lq,M*D]; the three
// gradients have the shapes of value / sampling_loc / attn_weight and are fully (over)written: the library zero-fills
// grad_value (and, on the atomics path, the other two) on `stream` first, where the reference allocates zeros (:121-123).
extern "C" int mmsa_ms_deform_attn_backward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                            const void* sampling_loc, const void* attn_weight, const void* grad_output,
                                            void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int batch,
                                            int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                            int num_point, int im2col_step, int dtype, hipStream_t stream) {
  MMSA_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_output && grad_value &&
                 grad_sampling_loc && grad_attn_weight, "ms_deform_attn_backward: null pointer");
  MMSA_CHECK_ARG(dtype == MMSA_DT_F32 || dtype == MMSA_DT_F16 || dtype == MMSA_DT_F64, "ms_deform_attn_backward: dtype %d not supported (0 f32, 1 f16, 2 f64)", dtype);
  int rc = msda_check(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, "ms_deform_attn_backward");
  if (rc) return rc;
  const int step = batch < im2col_step ? batch : im2col_step;
  MMSA_CHECK_ARG(step > 0 && batch % step == 0, "batch(%d) must divide im2col_step(%d)", batch, step);
  MMSA_CHECK_ARG(dtype != MMSA_DT_F16 || (((uintptr_t)grad_value) & 3) == 0, "ms_deform_attn_backward: f16 grad_value must be 4-byte aligned");
  const size_t es = msda_elt(dtype);
  const long nsamp = (long)batch * num_query * num_heads * num_levels * num_point;
  const bool shfl = channels <= 64 && (channels & (channels - 1)) == 0;
  if (hipMemsetAsync(grad_value, 0, (size_t)batch * spatial_size * num_heads * channels * es, stream) != hipSuccess ||
      (!shfl && (hipMemsetAsync(grad_sampling_loc, 0, (size_t)nsamp * 2 * es, stream) != hipSuccess ||
                 hipMemsetAsync(grad_attn_weight, 0, (size_t)nsamp * es, stream) != hipSuccess))) {
    mmsa_set_error("ms_deform_attn_backward: hipMemsetAsync failed");
    return MMSA_ERR_LAUNCH;
  }
  const long total = (long)batch * num_query * num_heads * channels;
  MMSA_CHECK_ARG(cdiv(total, 256) < (1L << 31) - 1, "ms_deform_attn_backward: problem too large");
#define MSDA_BWD(T_, S_)                                                                                                  \
  hipLaunchKernelGGL((msda_bwd_kernel<T_, S_>), dim3(cdiv(total, 256)), dim3(256), 0, stream, (const T_*)value, spatial_shapes, \
                     level_start_index, (const T_*)sampling_loc, (const T_*)attn_weight, (const T_*)grad_output, (T_*)grad_value, \
                     (T_*)grad_sampling_loc, (T_*)grad_attn_weight, batch, spatial_size, num_heads, channels, num_levels,  \
                     num_query, num_point)
  if (dtype == MMSA_DT_F16) { if (shfl) MSDA_BWD(__half, true); else MSDA_BWD(__half, false); }
  else if (dtype == MMSA_DT_F64) { if (shfl) MSDA_BWD(double, true); else MSDA_BWD(double, false); }
  else { if (shfl) MSDA_BWD(float, true); else MSDA_BWD(float, false); }
#undef MSDA_BWD
  MMSA_CHECK_LAUNCH("ms_deform_attn_backward");
  return MMSA_OK;
}

// Fused hot-path variant: consumes the raw output of the (concatenated) sampling_offsets|attention_weights
// projection and does the softmax over L*P and the location arithmetic of
// OPS/modules/ms_deform_attn.py:105-119 in-kernel (reference points are per query, broadcast over levels,
// as produced by AM:397-431).
extern "C" int mmsa_msda_fused(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* raw, long ldraw, const float* ref_points, float* out, long ldo,
                               unsigned short* out_p, long ldop, int out_fmt,
                               int batch, int spatial_size, int num_heads, int channels, int num_levels,
                               int num_query, int num_point, float* clamp_max, hipStream_t stream) {
  MMSA_CHECK_ARG(out_fmt >= MMSA_FMT_B3 && out_fmt <= MMSA_FMT_F3, "msda_fused: bad output plane format %d", out_fmt);   // (f3 since round 6: an interaction that followed its blocks onto fp16 pairs)
  MMSA_CHECK_ARG(value && spatial_shapes && level_start_index && raw && ref_points && (out || out_p), "msda_fused: null pointer");
  MMSA_CHECK_ARG(!out_p || (ldop >= (out_fmt == MMSA_FMT_H8C ? 3L * MMSA_PAD64(num_heads * channels) : 2L * num_heads * channels) && (ldop & 63) == 0 && (((uintptr_t)out_p) & 127) == 0),
                 "msda_fused: bad output planes");
  int rc = msda_check(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, "msda_fused");
  if (rc) return rc;
  MMSA_CHECK_ARG(ldraw >= (long)num_heads * num_levels * num_point * 3, "msda_fused: ldraw too small");
  MMSA_CHECK_ARG(!out || (ldo >= (long)num_heads * channels && (ldo & 3) == 0), "msda_fused: bad ldo");
  MMSA_CHECK_ARG((channels & 3) == 0 && channels <= 1024, "msda_fused: channels per head D=%d must be a multiple of 4 (<= 1024)", channels);
  MMSA_CHECK_ARG(((((uintptr_t)value) | ((uintptr_t)out)) & 15) == 0, "msda_fused: value/out must be 16-byte aligned");
  const long threads = (long)batch * num_query * num_heads * (channels >> 2);
  MMSA_CHECK_ARG(threads < (1L << 31), "msda_fused: problem too large for the 32-bit index arithmetic");
  const int bs = msda_block(channels);
  hipLaunchKernelGGL(msda_kernel<true>, dim3(cdiv(threads, bs)), dim3(bs), 0, stream, value, spatial_shapes,
                     level_start_index, nullptr, nullptr, raw, ldraw, ref_points, out, ldo, out_p, ldop, out_fmt, batch, spatial_size,
                     num_heads, channels, num_levels, num_query, num_point, clamp_max);
  MMSA_CHECK_LAUNCH("msda_fused");
  return MMSA_OK;
}

// The same on fp16 values (msda_planes_kernel above): `value_planes` = the value projection's output as MMSA_FMT_H8 activation planes [batch * spatial_size,
// >= 2 * num_heads * channels] (row stride ldvp in uint16, 128-byte aligned); lo_bytes != 0: the planes' e5m2 lo bytes are added (~14 bits instead of 11).
extern "C" int mmsa_msda_fused_planes(const unsigned short* value_planes, long ldvp, int lo_bytes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                      const float* raw, long ldraw, const float* ref_points, float* out, long ldo,
                                      unsigned short* out_p, long ldop, int out_fmt,
                                      int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                      int num_query, int num_point, float* clamp_max, hipStream_t stream) {
  MMSA_CHECK_ARG(out_fmt >= MMSA_FMT_B3 && out_fmt <= MMSA_FMT_F3, "msda_fused_planes: bad output plane format %d", out_fmt);
  MMSA_CHECK_ARG(value_planes && spatial_shapes && level_start_index && raw && ref_points && (out || out_p), "msda_fused_planes: null pointer");
  MMSA_CHECK_ARG(ldvp >= 2L * num_heads * channels && (ldvp & 63) == 0 && (((uintptr_t)value_planes) & 127) == 0 && ((num_heads * channels) & 31) == 0,
                 "msda_fused_planes: value planes need a 128-byte aligned base, a row stride that is a multiple of 64 and >= 2 * heads * channels, heads * channels %% 32 == 0");
  MMSA_CHECK_ARG(!out_p || (ldop >= (out_fmt == MMSA_FMT_H8C ? 3L * MMSA_PAD64(num_heads * channels) : 2L * num_heads * channels) && (ldop & 63) == 0 && (((uintptr_t)out_p) & 127) == 0),
                 "msda_fused_planes: bad output planes");
  int rc = msda_check(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, "msda_fused_planes");
  if (rc) return rc;
  MMSA_CHECK_ARG(ldraw >= (long)num_heads * num_levels * num_point * 3, "msda_fused_planes: ldraw too small");
  MMSA_CHECK_ARG(!out || (ldo >= (long)num_heads * channels && (ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0), "msda_fused_planes: bad out / ldo");
  MMSA_CHECK_ARG((channels & 7) == 0 && channels <= 2048, "msda_fused_planes: channels per head D=%d must be a multiple of 8", channels);
  const long threads = (long)batch * num_query * num_heads * (channels >> 3);
  MMSA_CHECK_ARG(threads < (1L << 31), "msda_fused_planes: problem too large for the 32-bit index arithmetic");
  const int d8 = channels >> 3, bs = (256 / d8) * d8;
  if (lo_bytes) hipLaunchKernelGGL(msda_planes_kernel<true>, dim3(cdiv(threads, bs)), dim3(bs), 0, stream, value_planes, ldvp, spatial_shapes, level_start_index, raw, ldraw,
                                   ref_points, out, ldo, out_p, ldop, out_fmt, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, clamp_max);
  else hipLaunchKernelGGL(msda_planes_kernel<false>, dim3(cdiv(threads, bs)), dim3(bs), 0, stream, value_planes, ldvp, spatial_shapes, level_start_index, raw, ldraw,
                          ref_points, out, ldo, out_p, ldop, out_fmt, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, clamp_max);
  MMSA_CHECK_LAUNCH("msda_fused_planes");
  return MMSA_OK;
}

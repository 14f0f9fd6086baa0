/* ORACLE (test infrastructure, NOT product code): plain-C CPU restatement of the reference's native op
 * ms_deform_attn_forward -- the loop of segmentation/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 and the
 * 4-tap zero-padded bilinear read of :33-84 -- one output scalar at a time, fp32 or fp64.
 * Pinned by tests/test_oracle_c.py against the golden vectors generated from the reference
 * (tests/golden/msda.npz, incl. the reference's own fixture ops/test.py:16-33).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. */
#include <math.h>
#include <stdint.h>

#define DEFINE_MSDA(NAME, T)                                                                                          \
  static T NAME##_bilinear(const T* data, int H, int W, int nheads, int channels, T h, T w, int m, int c) {          \
    const int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                                          \
    const int h_high = h_low + 1, w_high = w_low + 1;                                                                 \
    const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                                 \
    const long w_stride = (long)nheads * channels, h_stride = (long)W * w_stride;                                     \
    const long base = (long)m * channels + c;                                                                         \
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                                 \
    if (h_low >= 0 && w_low >= 0) v1 = data[h_low * h_stride + w_low * w_stride + base];                              \
    if (h_low >= 0 && w_high <= W - 1) v2 = data[h_low * h_stride + w_high * w_stride + base];                        \
    if (h_high <= H - 1 && w_low >= 0) v3 = data[h_high * h_stride + w_low * w_stride + base];                        \
    if (h_high <= H - 1 && w_high <= W - 1) v4 = data[h_high * h_stride + w_high * w_stride + base];                  \
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;                                                 \
  }                                                                                                                   \
  void NAME(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw, T* out, int N,     \
            int S, int M, int D, int L, int Lq, int P) {                                                              \
    const long qid_stride = (long)M * D;                                                                              \
    for (long idx = 0; idx < (long)N * Lq * M * D; ++idx) {                                                           \
      long t = idx;                                                                                                   \
      const int c = (int)(t % D); t /= D;                                                                             \
      const long sampling_index = t;                                                                                  \
      const int m = (int)(t % M); t /= M;                                                                             \
      t /= Lq;                                                                                                        \
      const int b = (int)t;                                                                                           \
      long wptr = sampling_index * L * P, lptr = wptr * 2;                                                            \
      T col = 0;                                                                                                      \
      for (int l = 0; l < L; ++l) {                                                                                   \
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                                 \
        const T* v = value + ((long)b * S + lsi[l]) * qid_stride;                                                     \
        for (int p = 0; p < P; ++p) {                                                                                 \
          const T loc_w = loc[lptr], loc_h = loc[lptr + 1], weight = aw[wptr];                                        \
          const T h_im = loc_h * H - (T)0.5, w_im = loc_w * W - (T)0.5;                                               \
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                                                         \
            col += NAME##_bilinear(v, H, W, M, D, h_im, w_im, m, c) * weight;                                         \
          wptr += 1; lptr += 2;                                                                                       \
        }                                                                                                             \
      }                                                                                                               \
      out[idx] = col;                                                                                                 \
    }                                                                                                                 \
  }

DEFINE_MSDA(msda_ref_f32, float)
DEFINE_MSDA(msda_ref_f64, double)

/* Backward: restatement of ms_deform_attn_backward's arithmetic -- the per-sample gradient of the 4-tap bilinear read,
 * segmentation/ops/src/cuda/ms_deform_im2col_cuda.cuh:86-160, inside the (b, q, m, c) x (l, p) loops of the col2im kernels
 * (:301-920; host side ms_deform_attn_cuda.cu:83-151): grad_value scatters w_corner * g * weight, grad_attn_weight and
 * grad_sampling_loc sum over the channels of a head.  Plain sequential loops, fp32 or fp64; all three outputs are zeroed first.
 * Pinned by tests/test_oracle_c.py against gradients that autograd takes through the reference's own
 * ms_deform_attn_core_pytorch (tests/golden/msda_bwd.npz). */
#define DEFINE_MSDA_BWD(NAME, T)                                                                                      \
  void NAME(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw, const T* gout,     \
            T* gvalue, T* gloc, T* gaw, int N, int S, int M, int D, int L, int Lq, int P) {                           \
    const long vstride = (long)M * D;                                                                                 \
    for (long i = 0; i < (long)N * S * vstride; ++i) gvalue[i] = 0;                                                   \
    for (long i = 0; i < (long)N * Lq * M * L * P; ++i) { gaw[i] = 0; gloc[2 * i] = 0; gloc[2 * i + 1] = 0; }         \
    for (long idx = 0; idx < (long)N * Lq * M * D; ++idx) {                                                           \
      const int c = (int)(idx % D);                                                                                   \
      const long pair = idx / D;                                                                                      \
      const int m = (int)(pair % M);                                                                                  \
      const int b = (int)(pair / M / Lq);                                                                             \
      const T g = gout[idx];                                                                                          \
      for (int l = 0; l < L; ++l) {                                                                                   \
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                                 \
        const long base = ((long)b * S + lsi[l]) * vstride + (long)m * D + c;                                         \
        for (int p = 0; p < P; ++p) {                                                                                 \
          const long si = pair * L * P + (long)l * P + p;                                                             \
          const T weight = aw[si];                                                                                    \
          const T h = loc[2 * si + 1] * H - (T)0.5, w = loc[2 * si] * W - (T)0.5;                                     \
          if (!(h > -1 && w > -1 && h < H && w < W)) continue;                                                        \
          const int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                                    \
          const int h_high = h_low + 1, w_high = w_low + 1;                                                           \
          const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                           \
          const T tg = g * weight;                                                                                    \
          T val = 0, dh = 0, dw = 0;                                                                                  \
          if (h_low >= 0 && w_low >= 0) {                                                                             \
            const long o = base + ((long)h_low * W + w_low) * vstride;                                                \
            val += hh * hw * value[o]; dh -= hw * value[o]; dw -= hh * value[o]; gvalue[o] += hh * hw * tg;           \
          }                                                                                                           \
          if (h_low >= 0 && w_high <= W - 1) {                                                                        \
            const long o = base + ((long)h_low * W + w_high) * vstride;                                               \
            val += hh * lw * value[o]; dh -= lw * value[o]; dw += hh * value[o]; gvalue[o] += hh * lw * tg;           \
          }                                                                                                           \
          if (h_high <= H - 1 && w_low >= 0) {                                                                        \
            const long o = base + ((long)h_high * W + w_low) * vstride;                                               \
            val += lh * hw * value[o]; dh += hw * value[o]; dw -= lh * value[o]; gvalue[o] += lh * hw * tg;           \
          }                                                                                                           \
          if (h_high <= H - 1 && w_high <= W - 1) {                                                                   \
            const long o = base + ((long)h_high * W + w_high) * vstride;                                              \
            val += lh * lw * value[o]; dh += lw * value[o]; dw += lh * value[o]; gvalue[o] += lh * lw * tg;           \
          }                                                                                                           \
          gaw[si] += g * val;                                                                                         \
          gloc[2 * si] += (T)W * dw * tg;                                                                             \
          gloc[2 * si + 1] += (T)H * dh * tg;                                                                         \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  }

DEFINE_MSDA_BWD(msda_ref_bwd_f32, float)
DEFINE_MSDA_BWD(msda_ref_bwd_f64, double)

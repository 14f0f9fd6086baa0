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

// Stand-alone ablation of the ConvNeXt 7x7 depthwise kernels (csrc/conv.hip: dwconv7_tiled_kernel / dwconv7_slide_kernel) at the stage shapes of
// ViT-L 1024^2: which of {halo loads, arithmetic, stores} holds the time.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I multimodal-sam-adapter_amd/csrc
// ABL bits: 1 = no global loads of activations, 2 = no arithmetic (accumulate one tap), 4 = no stores (one lane stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float mmsa_f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int ABL>
__global__ __launch_bounds__(256) void tiled(const float* __restrict__ x, long ldx, long xstrideB, const float* __restrict__ w, const float* __restrict__ bias,
                                             float* __restrict__ y, long ldy, long ystrideB, int H, int W, int C, int tilesX) {
  constexpr int TW = 14, CB = 64;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int b = blockIdx.z, c0 = blockIdx.y * CB;
  const int tx0 = (blockIdx.x % tilesX) * 8, ty0 = (blockIdx.x / tilesX) * 8;
  const float* xb = x + (long)b * xstrideB;
  {
    float4 v[13];
#pragma unroll
    for (int it = 0; it < 13; ++it) {
      const int i = threadIdx.x + it * 256;
      const int cv = i & 15, pos = i >> 4;
      const int ly = pos / TW, lx = pos - ly * TW;
      const int iy = ty0 + ly - 3, ix = tx0 + lx - 3;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!(ABL & 1) && i < TW * TW * 16 && iy >= 0 && iy < H && ix >= 0 && ix < W) v[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + c0 + cv * 4);
    }
#pragma unroll
    for (int it = 0; it < 13; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < TW * TW * 16) *reinterpret_cast<float4*>(tile + (i >> 4) * CB + (i & 15) * 4) = v[it];
    }
  }
  __syncthreads();
  const int cv = threadIdx.x & 15, strip = threadIdx.x >> 4;
  const int oy = strip >> 1, ox0 = (strip & 1) * 4;
  const int c = c0 + cv * 4;
  mmsa_f2 acc01[4], acc23[4];
  const float4 bv = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
  for (int p = 0; p < 4; ++p) { acc01[p] = (mmsa_f2){bv.x, bv.y}; acc23[p] = (mmsa_f2){bv.z, bv.w}; }
  float4 fa[7], fb[7];
#define DW7_LOADW(f_, kh_) _Pragma("unroll") for (int kw = 0; kw < 7; ++kw) f_[kw] = *reinterpret_cast<const float4*>(w + (long)((kh_) * 7 + kw) * C + c);
#define DW7_ROW(f_, kh_)                                                                                                          \
  {                                                                                                                               \
    float4 in[10];                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 10; ++i) in[i] = *reinterpret_cast<const float4*>(tile + ((oy + (kh_)) * TW + ox0 + i) * CB + cv * 4); \
    _Pragma("unroll") for (int kw = 0; kw < ((ABL & 2) ? 1 : 7); ++kw) {                                                          \
      const mmsa_f2 f01 = {f_[kw].x, f_[kw].y}, f23 = {f_[kw].z, f_[kw].w};                                                       \
      _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                                             \
        const mmsa_f2 i01 = {in[p + kw].x, in[p + kw].y}, i23 = {in[p + kw].z, in[p + kw].w};                                     \
        acc01[p] = __builtin_elementwise_fma(i01, f01, acc01[p]);                                                                 \
        acc23[p] = __builtin_elementwise_fma(i23, f23, acc23[p]);                                                                 \
      }                                                                                                                           \
    }                                                                                                                             \
  }
  DW7_LOADW(fa, 0)
#pragma unroll 1
  for (int kh = 0; kh < ((ABL & 2) ? 2 : 6); kh += 2) {
    __builtin_amdgcn_sched_barrier(0);
    DW7_LOADW(fb, kh + 1)
    __builtin_amdgcn_sched_barrier(0);
    DW7_ROW(fa, kh)
    __builtin_amdgcn_sched_barrier(0);
    DW7_LOADW(fa, kh + 2)
    __builtin_amdgcn_sched_barrier(0);
    DW7_ROW(fb, kh + 1)
  }
  DW7_ROW(fa, 6)
  const int gy = ty0 + oy;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int gx = tx0 + ox0 + p;
    const float4 a = make_float4(acc01[p].x, acc01[p].y, acc23[p].x, acc23[p].y);
    if (!(ABL & 4) || (a.x == 12345.f)) *reinterpret_cast<float4*>(y + (long)b * ystrideB + ((long)gy * W + gx) * ldy + c) = a;
  }
}

template <int ABL>
__global__ __launch_bounds__(256) void slide(const float* __restrict__ x, long ldx, long xstrideB, const float* __restrict__ w, const float* __restrict__ bias,
                                             float* __restrict__ y, long ldy, long ystrideB, int H, int W, int C, int tilesX, int seg_tiles) {
  constexpr int TW = 14, CB = 64, RING = 14;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;
  float* wl = smem + RING * TW * CB;
  const int b = blockIdx.z, c0 = blockIdx.y * CB;
  const int tx0 = (blockIdx.x % tilesX) * 8, ty_begin = (blockIdx.x / tilesX) * seg_tiles * 8;
  const int T = min(seg_tiles, (H - ty_begin + 7) >> 3);
  const float* xb = x + (long)b * xstrideB;
  {
    float4 v[13], wv[4];
#pragma unroll
    for (int it = 0; it < 13; ++it) {
      const int i = threadIdx.x + it * 256;
      const int cv = i & 15, pos = i >> 4;
      const int ly = pos / TW, lx = pos - ly * TW;
      const int iy = ty_begin + ly - 3, ix = tx0 + lx - 3;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!(ABL & 1) && i < TW * TW * 16 && iy >= 0 && iy < H && ix >= 0 && ix < W) v[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + c0 + cv * 4);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = threadIdx.x + it * 256;
      wv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 49 * 16) wv[it] = *reinterpret_cast<const float4*>(w + (long)(i >> 4) * C + c0 + (i & 15) * 4);
    }
#pragma unroll
    for (int it = 0; it < 13; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < TW * TW * 16) *reinterpret_cast<float4*>(tile + (i >> 4) * CB + (i & 15) * 4) = v[it];
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < 49 * 16) *reinterpret_cast<float4*>(wl + i * 4) = wv[it];
    }
  }
  __syncthreads();
  const int cv = threadIdx.x & 15, strip = threadIdx.x >> 4;
  const int oy = strip >> 1, ox0 = (strip & 1) * 4;
  const int c = c0 + cv * 4;
  const float4 bv = *reinterpret_cast<const float4*>(bias + c);
  for (int t = 0; t < T; ++t) {
    const bool more = t + 1 < T;
    float4 pv[7];
    if (more) {
#pragma unroll
      for (int it = 0; it < 7; ++it) {
        const int i = threadIdx.x + it * 256;
        const int pcv = i & 15, pos = i >> 4;
        const int ly = pos / TW, lx = pos - ly * TW;
        const int iy = ty_begin + 8 * t + 11 + ly, ix = tx0 + lx - 3;
        pv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(ABL & 1) && iy < H && ix >= 0 && ix < W) pv[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + c0 + pcv * 4);
      }
    }
    mmsa_f2 acc01[4], acc23[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { acc01[p] = (mmsa_f2){bv.x, bv.y}; acc23[p] = (mmsa_f2){bv.z, bv.w}; }
    int srow = (8 * t + oy) % RING;
#pragma unroll 1
    for (int kh = 0; kh < ((ABL & 2) ? 1 : 7); ++kh) {
      float4 in[10], f[7];
#pragma unroll
      for (int i = 0; i < 10; ++i) in[i] = *reinterpret_cast<const float4*>(tile + (srow * TW + ox0 + i) * CB + cv * 4);
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) f[kw] = *reinterpret_cast<const float4*>(wl + (kh * 7 + kw) * CB + cv * 4);
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const mmsa_f2 f01 = {f[kw].x, f[kw].y}, f23 = {f[kw].z, f[kw].w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const mmsa_f2 i01 = {in[p + kw].x, in[p + kw].y}, i23 = {in[p + kw].z, in[p + kw].w};
          acc01[p] = __builtin_elementwise_fma(i01, f01, acc01[p]);
          acc23[p] = __builtin_elementwise_fma(i23, f23, acc23[p]);
        }
      }
      srow = srow + 1 == RING ? 0 : srow + 1;
    }
    const int gy = ty_begin + 8 * t + oy;
    if (gy < H) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int gx = tx0 + ox0 + p;
        const float4 a = make_float4(acc01[p].x, acc01[p].y, acc23[p].x, acc23[p].y);
        if (!(ABL & 4) || (a.x == 12345.f)) *reinterpret_cast<float4*>(y + (long)b * ystrideB + ((long)gy * W + gx) * ldy + c) = a;
      }
    }
    if (!more) break;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int i = threadIdx.x + it * 256;
      const int pos = i >> 4;
      const int ly = pos / TW, lx = pos - ly * TW;
      const int slot = (8 * t + ly) % RING;
      *reinterpret_cast<float4*>(tile + (slot * TW + lx) * CB + (i & 15) * 4) = pv[it];
    }
    __syncthreads();
  }
}

template <typename F>
static float time_us(F launch, int reps = 50) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms * 1e3f / reps);
  }
  return best;
}

template <int ABL>
static void run(const char* tag, int C, int H, int B, int seg, float* x, float* w, float* bias, float* y) {
  const int tx = H / 8, ty = H / 8;
  const long sB = (long)H * H * C;
  float t0 = time_us([&] { hipLaunchKernelGGL(tiled<ABL>, dim3(tx * ty, C / 64, B), dim3(256), 14 * 14 * 64 * 4, 0, x, (long)C, sB, w, bias, y, (long)C, sB, H, H, C, tx); });
  printf("%-10s C=%4d %3dx%-3d  tiled %7.1f us |", tag, C, H, H, t0);
  for (int s = seg; s >= 1 && s >= seg / 4; s /= 2) {
    float t1 = time_us([&] { hipLaunchKernelGGL(slide<ABL>, dim3(tx * ((ty + s - 1) / s), C / 64, B), dim3(256), (14 * 14 * 64 + 49 * 64) * 4, 0, x, (long)C, sB, w, bias, y, (long)C, sB, H, H, C, tx, s); });
    printf("  slide seg %2d: %7.1f us", s, t1);
  }
  printf("\n");
}

int main() {
  const int B = 4;
  const size_t n = (size_t)B * 256 * 256 * 128;
  float *x, *y, *w, *bias;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&w, 49 * 768 * 4)); CK(hipMalloc(&bias, 768 * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), 49 * 768 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data(), 768 * 4, hipMemcpyHostToDevice));
  const int shapes[4][3] = {{64, 256, 32}, {192, 128, 16}, {384, 64, 8}, {768, 32, 4}};   // (C, H, tile rows); stage 0 with C = 64 (one full chunk of its 96)
  for (auto& s : shapes) {
    run<0>("full", s[0], s[1], B, s[2], x, w, bias, y);
    run<1>("no loads", s[0], s[1], B, s[2], x, w, bias, y);
    run<2>("no fma", s[0], s[1], B, s[2], x, w, bias, y);
    run<4>("no stores", s[0], s[1], B, s[2], x, w, bias, y);
    run<6>("loads only", s[0], s[1], B, s[2], x, w, bias, y);
    run<5>("fma only", s[0], s[1], B, s[2], x, w, bias, y);
    run<3>("stores only", s[0], s[1], B, s[2], x, w, bias, y);
  }
  return 0;
}

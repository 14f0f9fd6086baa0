// Loop (+ epilogue) microbenchmark of the TWO-WORKGROUPS-PER-CU form of the h8c GEMM (round 5; VERDICT r04 item 1: "a <= 80 KiB-LDS flavour of the h8c kernel
// (128 x 128 tile, 4 waves, 2-unit ring) so that two workgroups share a CU and one's store burst runs under the other's k loop ... microbenchmark first").
//
// Same operand layout and arithmetic as tools/exp/h8c_loop.hip (HI plane fp16 row-major, LO plane one 128-byte line per row pair and 64-k chunk, q(hi) = the
// fp16 top byte).  Workgroup = 4 waves (2 x 2, wave tile 64 x 64), tile 128 x 128, LDS = two HI units of 32 KiB (128 A rows + 128 W rows x 128 B) + ONE LO unit
// of 16 KiB = 80 KiB: two workgroups per CU, one wave of each on every SIMD, NO ping-pong inside a workgroup -- the overlap of operand stream, fragment reads and
// matrix work comes from the other workgroup's wave on the same SIMD.  Per pair of k-tiles: hi fragments (16 ds_read_b128) -> barrier A (every wave's reads
// done, HI(j+1) and LO(j) landed) -> HI(j+2) requested into the unit just read (8 DMA instructions per wave) -> 32 fp16 MFMAs -> lo fragments + 32 v_perm_b32 ->
// barrier B -> LO(j+1) requested (4 DMA instructions per wave) -> 16 block-scaled fp8 MFMAs.  96 DMA instructions per CU and (256 x 128 x 64)-equivalent of work
// against the 8-wave kernel's 72: the question is whether two independent workgroups overlap the stream with the matrix pipe well enough to pay for that.
// EPI: 0 = none (loop only), 1 = bias + GELU + fp16 hi / e5m2 lo stores from the accumulator layout (lin1-like), 2 = fp32 read-modify-write (proj / out-proj / fc2-like).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DEPI=1] tools/exp/h8c_2wg.hip -o gpurun_out/h8c_2wg ;  run: h8c_2wg check | time
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;
typedef __attribute__((ext_vector_type(2))) _Float16 h2v;
typedef __attribute__((ext_vector_type(2))) float f2v;

#define H_UNIT 32768
#define L_UNIT 16384
#define LDS_TOTAL (2 * H_UNIT + L_UNIT)
#define H8_SCALE 0x74747474
#ifndef EPI
#define EPI 0
#endif

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct Args {
  const unsigned char* Ah; long ldh;
  const unsigned char* Al; long ldl;
  const unsigned char* Wh; long ldwh;
  const unsigned char* Wl; long ldwl;
  float* C; long ldc;                   // validation (EPI 0) / residual stream (EPI 2)
  unsigned short* Ph; unsigned char* Pl;   // EPI 1: fp16 hi [M, N] and lo bytes [M, N]
  const float* bias;
  int M, N, K, nbm, nbn, ntiles;
};

__device__ __forceinline__ float gelu1(float x) {   // the library's form (csrc/common.h): one transcendental
  const float c[7] = {-1.151126981e+00f, -4.590439200e-01f, -5.294858292e-02f, 7.670805324e-03f, -5.757985055e-04f, -1.279479329e-05f, 4.278379038e-06f};
  const float u = fminf(fabsf(x), 6.0f);
  float p = fmaf(c[6], u, c[5]);
#pragma unroll
  for (int k = 4; k >= 0; --k) p = fmaf(p, u, c[k]);
  p *= u;
  const float r = 1.0f - __builtin_amdgcn_exp2f(p);
  const float h = 0.5f * x;
  return fmaf(fabsf(h), r, h);
}

#ifndef PRIO
#define PRIO 0   // 1: the k loop runs at s_setprio 3, the epilogue at 0 -- the co-resident workgroup's MFMA phases win the issue port, the epilogue fills what they leave
#endif
__global__ __launch_bounds__(256, 2) void h8c_2wg_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int np = a.K >> 6;
  const int G = gridDim.x;
  int rb = blockIdx.x;
  { const int xcd = rb & 7, q = G >> 3, r = G & 7; rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rb >> 3); }
  const int my_tiles = (a.ntiles - rb + G - 1) / G;
  if (my_tiles <= 0) return;
  const int total = my_tiles * np;

  const int drow = lane >> 3;
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 16;
  const int lq = ((lane & 7) ^ ((-(drow >> 1)) & 3)) * 16;
  const int lds_ha = wave * 32 * 128, lds_hw = 16384 + wave * 32 * 128;
  const int lds_la = wave * 2048, lds_lw = 8192 + wave * 2048;

  const unsigned char *hA, *hW, *lA, *lW;
  unsigned oa[4], ow[4], la_[2], lw_[2];
  int hp_tile = rb, hp_p = 0, hp_j = 0, lp_tile = rb, lp_p = 0, lp_j = 0;
#define TILE_MN(t_, m0_, n0_) { const int mi_ = (t_) / a.nbn; m0_ = mi_ * 128; n0_ = ((t_) - mi_ * a.nbn) * 128; }
#define SET_H(t_)                                                                                \
  { int m0_, n0_; TILE_MN(t_, m0_, n0_)                                                          \
    hA = a.Ah + (long)m0_ * a.ldh; hW = a.Wh + (long)n0_ * a.ldwh;                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                           \
      const int ab_ = m0_ + wave * 32 + 8 * i_ + drow, wb_ = n0_ + wave * 32 + 8 * i_ + drow;    \
      oa[i_] = (unsigned)((min(ab_, a.M - 1) - m0_) * (int)a.ldh + ((i_ & 1) ? (dpiece ^ 64) : dpiece));   \
      ow[i_] = (unsigned)((min(wb_, a.N - 1) - n0_) * (int)a.ldwh + ((i_ & 1) ? (dpiece ^ 64) : dpiece)); } }
#define SET_L(t_)                                                                                \
  { int m0_, n0_; TILE_MN(t_, m0_, n0_)                                                          \
    lA = a.Al + (long)(m0_ >> 1) * a.ldl; lW = a.Wl + (long)(n0_ >> 1) * a.ldwl;                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                           \
      const int aj_ = (m0_ >> 1) + wave * 16 + 8 * i_ + drow, wj_ = (n0_ >> 1) + wave * 16 + 8 * i_ + drow;   \
      la_[i_] = (unsigned)((min(aj_, (a.M - 1) >> 1) - (m0_ >> 1)) * (int)a.ldl + lq);            \
      lw_[i_] = (unsigned)((min(wj_, (a.N - 1) >> 1) - (n0_ >> 1)) * (int)a.ldwl + lq); } }
  SET_H(hp_tile) SET_L(lp_tile)
#define H_ISSUE()                                                                                \
  { unsigned char* d_ = smem + H_UNIT * (hp_j & 1); const long ko_ = (long)hp_p * 128;            \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { GLDS16(hA + ko_ + oa[i_], d_ + lds_ha + 1024 * i_); GLDS16(hW + ko_ + ow[i_], d_ + lds_hw + 1024 * i_); } }
#define H_ADVANCE() { ++hp_j; if (++hp_p == np) { hp_p = 0; hp_tile += G; if (hp_j < total) SET_H(hp_tile) } }
#define L_ISSUE()                                                                                \
  { unsigned char* d_ = smem + 2 * H_UNIT; const long ko_ = (long)lp_p * 128;                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) { GLDS16(lA + ko_ + la_[i_], d_ + lds_la + 1024 * i_); GLDS16(lW + ko_ + lw_[i_], d_ + lds_lw + 1024 * i_); } }
#define L_ADVANCE() { ++lp_j; if (++lp_p == np) { lp_p = 0; lp_tile += G; if (lp_j < total) SET_L(lp_tile) } }

  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag0 = l15 * 128 + fslot * 16, frag1 = l15 * 128 + (fslot ^ 4) * 16;
  const int fha = (wm * 64) * 128, fhw = 16384 + (wn * 64) * 128;
  const int lo_off = 128 * (l15 >> 1) + 16 * ((((l15 & 1) << 2) | g) ^ ((-(l15 >> 2)) & 3));
  const int fla = (wm * 4) * 1024 + lo_off, flw = 8192 + (wn * 4) * 1024 + lo_off;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u4v ah0[4], ah1[4], wh0[4], wh1[4];
  v8i opA[4], opW[4];

  H_ISSUE() H_ADVANCE()
  if (total > 1) { H_ISSUE() H_ADVANCE() }
  L_ISSUE() L_ADVANCE()
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int j = 0, tile = rb;
  const unsigned psel = 0x07050301u;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define BAR() { SB(); __builtin_amdgcn_s_barrier(); SB(); }
#define PERM(d_, hi_, lo_) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d_) : "v"(hi_), "v"(lo_), "s"(psel))
  for (int tdone = 0; tdone < my_tiles; ++tdone) {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
    for (int p = 0; p < np; ++p) {
      const unsigned char* hb = smem + H_UNIT * (j & 1);
      const unsigned char* lb = smem + 2 * H_UNIT;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah0[i] = *reinterpret_cast<const u4v*>(hb + fha + i * 2048 + frag0);
        ah1[i] = *reinterpret_cast<const u4v*>(hb + fha + i * 2048 + frag1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh0[i] = *reinterpret_cast<const u4v*>(hb + fhw + i * 2048 + frag0);
        wh1[i] = *reinterpret_cast<const u4v*>(hb + fhw + i * 2048 + frag1);
      }
      SB();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // HI(j+1) (requested a pair ago) and LO(j) (half a pair ago)
      BAR()                                                // A: unit j & 1 is free; HI(j+1) and LO(j) are visible
      if (hp_j < total) { H_ISSUE() H_ADVANCE() }
      SB();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, wh0[ni]), __builtin_bit_cast(h8v, ah0[mi]), acc[ni][mi], 0, 0, 0);
      SB();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, wh1[ni]), __builtin_bit_cast(h8v, ah1[mi]), acc[ni][mi], 0, 0, 0);
      SB();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u4v la = *reinterpret_cast<const u4v*>(lb + fla + i * 1024);
        const u4v lw = *reinterpret_cast<const u4v*>(lb + flw + i * 1024);
        int a0, a1, a2, a3, w0, w1, w2, w3;
        PERM(a0, ah0[i][1], ah0[i][0]); PERM(a1, ah0[i][3], ah0[i][2]);
        PERM(a2, ah1[i][1], ah1[i][0]); PERM(a3, ah1[i][3], ah1[i][2]);
        PERM(w0, wh0[i][1], wh0[i][0]); PERM(w1, wh0[i][3], wh0[i][2]);
        PERM(w2, wh1[i][1], wh1[i][0]); PERM(w3, wh1[i][3], wh1[i][2]);
        opA[i] = (v8i){a0, a1, a2, a3, (int)la[0], (int)la[1], (int)la[2], (int)la[3]};
        opW[i] = (v8i){(int)lw[0], (int)lw[1], (int)lw[2], (int)lw[3], w0, w1, w2, w3};
      }
      SB();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BAR()                                                // B: the LO unit is free
      if (lp_j < total) { L_ISSUE() L_ADVANCE() }
      SB();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni], opA[mi], acc[ni][mi], 1, 1, 0, H8_SCALE, 0, 0x7f7f7f7f);
      SB();
      ++j;
    }
    // ---- epilogue, straight from the accumulator layout (lane: row l15 of a 16-row block, columns 4g .. 4g+3 of a 16-column block)
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    {
      int m0, n0;
      TILE_MN(tile, m0, n0)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int l15_ = lane_o & 15, g_ = lane_o >> 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int m = m0 + wm * 64 + mi * 16 + l15_, n = n0 + wn * 64 + ni * 16 + 4 * g_;
          const bool ok = m < a.M && n + 3 < a.N;
          f32x4 v = acc[ni][mi];
          acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if EPI == 0
          if (a.C && ok) *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n) = v;
          else asm volatile("" :: "v"(v));
#elif EPI == 1
          if (ok) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = gelu1(v[r] + b[r]);
            const h2v h01 = __builtin_convertvector((f2v){o[0], o[1]}, h2v), h23 = __builtin_convertvector((f2v){o[2], o[3]}, h2v);
            const f2v r01 = (f2v){o[0], o[1]} - __builtin_convertvector(h01, f2v), r23 = (f2v){o[2], o[3]} - __builtin_convertvector(h23, f2v);
            unsigned lo8 = 0u;
            lo8 = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(r01.x * 2240.f, r01.y * 2240.f, (int)lo8, false);
            lo8 = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(r23.x * 2240.f, r23.y * 2240.f, (int)lo8, true);
            *reinterpret_cast<uint2*>(a.Ph + (long)m * a.N + n) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            *reinterpret_cast<unsigned*>(a.Pl + (long)m * a.N + n) = lo8;
          }
#else
          if (ok) {
            f32x4* c = reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n);
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
            const f32x4 r = *c;
            *c = v + b + r;
          }
#endif
        }
    }
    tile += G;
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static unsigned char f2e5m2(float x) {   // round to nearest even via fp16
  unsigned short h = f2h(x);
  unsigned r = (unsigned)h + 0x7Fu + ((h >> 8) & 1u);
  return (unsigned char)(r >> 8);
}
static float e5m2f(unsigned char b) { return h2f((unsigned short)(b << 8)); }

struct Packed { std::vector<unsigned char> hi, lo; long ldh, ldl; };
static Packed pack(const std::vector<float>& x, int rows, int K) {
  Packed p;
  const int rp = (rows + 1) / 2, nc = K / 64;
  p.ldh = (long)K * 2; p.ldl = (long)nc * 128;
  p.hi.assign((size_t)rows * K * 2, 0); p.lo.assign((size_t)rp * nc * 128, 0);
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) {
      const float v = x[(size_t)r * K + k];
      const unsigned short h = f2h(v);
      memcpy(&p.hi[((size_t)r * K + k) * 2], &h, 2);
      const float lo = (v - h2f(h)) * 2048.f;
      const int c = k >> 6, kk = k & 63, kt = kk >> 5, gq = (kk & 31) >> 3, e = kk & 7;
      p.lo[(size_t)(r >> 1) * p.ldl + c * 128 + (r & 1) * 64 + gq * 16 + kt * 8 + e] = f2e5m2(lo);
    }
  return p;
}

int main(int argc, char** argv) {
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  if (check && EPI != 0) { printf("check needs the EPI=0 build (the other epilogues write buffers that the check mode does not allocate)\n"); return 2; }
  hipFuncSetAttribute((const void*)h8c_2wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  struct Shape { int M, N, K; const char* name; };
  std::vector<Shape> shapes;
  if (check) shapes = {{512, 256, 256, "small"}, {700, 384, 192, "ragged"}, {256, 128, 64, "one pair"}, {1024, 512, 1024, "deep"}};
  else shapes = {{8192, 4096, 1024, "lin1"}, {8192, 1024, 4096, "lin2"}, {8192, 3072, 1024, "qkv"}, {8192, 1024, 1024, "proj"},
                 {43008, 1024, 512, "ext out"}, {43008, 1024, 256, "ffn fc2"}, {4096, 4096, 1024, "lin1 1img"}, {4096, 1024, 4096, "lin2 1img"}};
  int bad = 0;
  for (auto& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xFFFF) / 65536.f; };
    auto nrm = [&]() { float u1 = rnd() + 1e-7f, u2 = rnd(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    for (auto& v : A) v = nrm();
    for (auto& v : W) v = nrm() / sqrtf((float)K);
    Packed pa = pack(A, M, K), pw = pack(W, N, K);
    unsigned char *dAh, *dAl, *dWh, *dWl; float* dC = nullptr;
    hipMalloc(&dAh, pa.hi.size()); hipMalloc(&dAl, pa.lo.size()); hipMalloc(&dWh, pw.hi.size()); hipMalloc(&dWl, pw.lo.size());
    hipMemcpy(dAh, pa.hi.data(), pa.hi.size(), hipMemcpyHostToDevice); hipMemcpy(dAl, pa.lo.data(), pa.lo.size(), hipMemcpyHostToDevice);
    hipMemcpy(dWh, pw.hi.data(), pw.hi.size(), hipMemcpyHostToDevice); hipMemcpy(dWl, pw.lo.data(), pw.lo.size(), hipMemcpyHostToDevice);
    Args a;
    a.Ah = dAh; a.ldh = pa.ldh; a.Al = dAl; a.ldl = pa.ldl; a.Wh = dWh; a.ldwh = pw.ldh; a.Wl = dWl; a.ldwl = pw.ldl;
    a.Ph = nullptr; a.Pl = nullptr; a.bias = nullptr;
    a.M = M; a.N = N; a.K = K; a.nbm = (M + 127) / 128; a.nbn = (N + 127) / 128; a.ntiles = a.nbm * a.nbn; a.C = nullptr; a.ldc = N;
    if (check) {
      hipMalloc(&dC, (size_t)M * N * 4); hipMemset(dC, 0xff, (size_t)M * N * 4);
      a.C = dC;
      for (int grid : {512, 3}) {
        const int gsz = a.ntiles < grid ? a.ntiles : grid;
        hipLaunchKernelGGL(h8c_2wg_kernel, dim3(gsz), dim3(256), LDS_TOTAL, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<float> C((size_t)M * N);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        // references: exact product, and the kernel's arithmetic (hi.hi + trunc(hi).q(lo) + q(lo).trunc(hi))
        double e_exact = 0, e_model = 0, nrm2 = 0;
        for (int m = 0; m < M; m += (M > 600 ? 7 : 1))
          for (int n = 0; n < N; n += (N > 300 ? 5 : 1)) {
            double ex = 0, md = 0;
            for (int k = 0; k < K; ++k) {
              const float av = A[(size_t)m * K + k], wv = W[(size_t)n * K + k];
              const unsigned short ha = f2h(av), hw = f2h(wv);
              const float la = e5m2f(f2e5m2((av - h2f(ha)) * 2048.f)) / 2048.f, lw = e5m2f(f2e5m2((wv - h2f(hw)) * 2048.f)) / 2048.f;
              const float qa = h2f(ha & 0xFF00), qw = h2f(hw & 0xFF00);
              ex += (double)av * wv;
              md += (double)h2f(ha) * h2f(hw) + (double)qa * lw + (double)la * qw;
            }
            const double c = C[(size_t)m * N + n];
            e_exact += (c - ex) * (c - ex); e_model += (c - md) * (c - md); nrm2 += ex * ex;
          }
        const double re = sqrt(e_exact / nrm2), rm = sqrt(e_model / nrm2);
        const bool ok = rm < 2e-6 && re < 1e-4;
        printf("check %-9s M=%d N=%d K=%d grid=%d: rel-L2 vs exact %.3e, vs the kernel's arithmetic %.3e %s\n", s.name, M, N, K, gsz, re, rm, ok ? "OK" : "FAIL");
        bad += !ok;
      }
    } else {
      float* dBias; hipMalloc(&dBias, (size_t)N * 4); hipMemset(dBias, 0, (size_t)N * 4); a.bias = dBias;
#if EPI == 1
      unsigned short* dPh; unsigned char* dPl; hipMalloc(&dPh, (size_t)M * N * 2); hipMalloc(&dPl, (size_t)M * N); a.Ph = dPh; a.Pl = dPl;
#elif EPI == 2
      float* dCr; hipMalloc(&dCr, (size_t)M * N * 4); hipMemset(dCr, 0, (size_t)M * N * 4); a.C = dCr;
#endif
      for (int cap : {512, 256}) {
        const int gsz0 = a.ntiles < cap ? a.ntiles : cap;
        const int rounds = (a.ntiles + gsz0 - 1) / gsz0, gsz = (a.ntiles + rounds - 1) / rounds;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(h8c_2wg_kernel, dim3(gsz), dim3(256), LDS_TOTAL, 0, a);
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(h8c_2wg_kernel, dim3(gsz), dim3(256), LDS_TOTAL, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / reps;
#ifdef FSTAMPS
        if (cap == 256 && &s == &shapes[0]) {
          unsigned long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fst), sizeof(h));
          const char* nm[16] = {"start", "X reads issued", "X dma issued", "X lgkm0", "X vmwait(g1)", "barrier a", "X mfma issued", "X vmwait(g0)", "barrier b", "Y reads+perm", "Y dma issued", "Y lgkm0", "barrier c", "Y mfma issued", "Y vmwait(g0)", "barrier d"};
          for (int gq = 0; gq < 2; ++gq) { printf("fine stamps group %d:", gq); for (int i = 1; i < 16; ++i) printf(" [%s +%llu]", nm[i], h[gq * 16 + i] - h[gq * 16 + i - 1]); printf("\n"); }
        }
#endif
#ifdef STAMPS
        if (cap == 256 && &s == &shapes[0]) {
          unsigned long long h[68]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
          for (int gq = 0; gq < 2; ++gq) { printf("stamps group %d (cycles between barriers):", gq); for (int i = 1; i < 32; ++i) printf(" %llu", h[gq * 32 + i] - h[gq * 32 + i - 1]); printf("\n"); }
        }
#endif
        printf("h8c 2wg EPI=%d %-12s M=%5d N=%5d K=%5d grid=%3d (%d tiles): %8.1f us  %6.1f TFLOP/s algorithmic\n", EPI, s.name, M, N, K, gsz, a.ntiles, us,
               2.0 * M * N * K / us / 1e6);
      }
    }
    hipFree(dAh); hipFree(dAl); hipFree(dWh); hipFree(dWl); if (dC) hipFree(dC);
  }
  if (check) printf(bad ? "CHECK FAILED\n" : "CHECK OK\n");
  return bad ? 1 : 0;
}

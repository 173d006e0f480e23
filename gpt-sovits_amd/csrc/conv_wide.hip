// Persistent variant of the tile convolution for the generator's 128-channel stage (fp16, Cin = Cout = 128, stride 1,
// 512 000 time steps at the benchmark shape: 16 of the generator's 90 ResBlock convs, 5.5 of its 14.3 ms).
//
// conv_lds_kernel (conv_lds.hip) runs ONE 256-step tile per workgroup, one workgroup per CU (83 KB window + 2 x 34 KB weight
// slabs of LDS): load the window (an exposed HBM round trip), 3-11 taps, write the tile -- strictly one after the other, so the
// HBM time of a tile (78 KB in, 64 KB residual, 64 KB out: ~10 us at a CU's share of the bandwidth) is never overlapped with
// its MFMAs.  tools/conv_probe.py: 321 / 382 / 446 us at 3 / 7 / 11 taps, i.e. ~275 us that do not depend on the tap count,
// against 100 us of HBM time and 47 us of MFMA time.  Here the workgroup is persistent over tiles (same construction as
// conv_narrow_f16_kernel): the NEXT tile's window is requested into registers before the current tile's tap loop and written to
// LDS after its epilogue, and the residual rows of a pass are requested one pass ahead.  Arithmetic, MFMA order and epilogue formula are those of conv_lds_kernel<half, 2, 4, 2, 2, 128>:
// results are bit-identical (tests/test_ops_gpu.py::test_wide_persistent_conv_matches_tile_kernel).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace gsv {

namespace {

typedef _Float16 T;
typedef h8 F;
typedef _Float16 T4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma32w(f16v& acc, const h8& a, const h8& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ h8 lrelu8w(h8 v, float s) { h8 t = v * (_Float16)s; return __builtin_elementwise_max(v, t); }
__device__ __forceinline__ h8 relu8w(h8 v) { return __builtin_elementwise_max(v, (h8){0, 0, 0, 0, 0, 0, 0, 0}); }

// NW = waves per workgroup: 4 (each wave 64 channels x 128 steps) or 8 (64 x 64: two waves per SIMD, so that one wave's LDS
// operand reads and waits overlap the other's MFMAs -- with one wave per SIMD every stall of the tap loop is exposed)
template <bool RES, bool ACCU, int NW>
__global__ __launch_bounds__(NW * 64) void conv_wide_f16_kernel(ConvArgs a, int rows_win, int ntiles) {
  constexpr int G = 8, KC = 16, CC = 128, CT = 128, TM = 2, TN = NW == 8 ? 2 : 4, WN = NW == 8 ? 4 : 2, NT = NW * 64;
  constexpr int LDX = CC + G, VPR = CC / G;             // 136, 16
  constexpr int WLOADS = CT * VPR / NT;                 // 8 vectors per thread per tap slab
  constexpr int XB = (306 * VPR + NT - 1) / NT;         // 20 vectors per thread per window
  constexpr int LDO = CT + 4, PR = TN * 32, IPR = CT / 4, NI = PR * IPR / NT;   // 128 (64) rows per epilogue pass, 16 (4) items per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = (T*)smem;                                     // [rows_win <= 306][LDX]; the fp32 epilogue tile [PR][LDO] aliases it
  T* ws = xs + (size_t)306 * LDX;                       // [2][CT][LDX]
  float* os = (float*)smem;
  static_assert((size_t)PR * LDO * 4 <= (size_t)306 * LDX * 2, "epilogue tile must fit in the window");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ w = (const T*)a.w;
  const int total = rows_win * VPR;
  const int ecg = tid % IPR, ec = 4 * ecg;
  f4 ebias = (f4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) for (int j = 0; j < 4; ++j) ebias[j] = a.bias[ec + j];

  auto load_w = [&](int tap, F* regs) {
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int v = tid + i * NT;
      const int row = v / VPR, col = v - row * VPR;
      regs[i] = *(const F*)(w + (long long)row * a.ldw + (long long)tap * CC + col * G);
    }
  };
  auto store_w = [&](int buf, const F* regs) {
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int v = tid + i * NT;
      const int row = v / VPR, col = v - row * VPR;
      *(F*)(ws + ((size_t)buf * CT + row) * LDX + col * G) = regs[i];
    }
  };
  // window of tile `tile` -> registers: clamped (always valid) addresses, zero rows outside the sequence selected after
  auto load_window = [&](int tile, F* regs) {
    const int win_start = tile * 256 - a.pad;
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int v = min(tid + i * NT, total - 1);
      const int row = v / VPR, col = v - row * VPR;
      const int ti = win_start + row;
      const F val = *(const F*)(x + (long long)min(max(ti, 0), a.T_in - 1) * a.ldx + col * G);
      regs[i] = (ti >= 0 && ti < a.T_in) ? val : (F){0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto store_window = [&](const F* regs) {
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int v = tid + i * NT;
      if (v < total) {
        const int row = v / VPR, col = v - row * VPR;
        F val = regs[i];
        if (a.pre_act == ACT_LRELU) val = lrelu8w(val, a.pre_slope);
        else if (a.pre_act == ACT_RELU) val = relu8w(val);
        *(F*)(xs + (size_t)row * LDX + col * G) = val;
      }
    }
  };
  int tile = blockIdx.x;
  {
    F first[XB], w0[WLOADS];
    load_window(min(tile, ntiles - 1), first);
    load_w(0, w0);
    store_window(first);
    store_w(0, w0);
  }
  lds_barrier();
  for (; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * 256;
    const bool pf = a.prof && blockIdx.x == 100 && tile == 100 + 3 * (int)gridDim.x && tid == 0;
    int pi = 0;
#define WSTAMP() do { if (pf) a.prof[pi++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    WSTAMP();
    // ---- requests that do not depend on this tile's arithmetic go out first
    F nxt[XB];
    load_window(min(tile + (int)gridDim.x, ntiles - 1), nxt);
    T4 rv[RES ? NI : 1];
    auto load_res_pass = [&](int pass) {
#pragma unroll
      for (int e = 0; e < (RES ? NI : 0); ++e) {
        const int t = min(t0 + pass * PR + (tid + e * NT) / IPR, a.T_out - 1);
        rv[e] = *(const T4*)((const T*)a.res + (long long)t * a.ldr + ec);
      }
    };
    if (RES) load_res_pass(0);
    f16v acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    for (int tap = 0; tap < a.taps; ++tap) {
      const int buf = tap & 1;
      F wnx[WLOADS];
      const bool more = tap + 1 < a.taps;
      if (more) load_w(tap + 1, wnx);
      const int shift = tap * a.dil;
      const T* wb = ws + (size_t)buf * CT * LDX;
#pragma unroll 2
      for (int ks = 0; ks < CC / KC; ++ks) {
        const int kk = ks * KC + G * h;
        F af[TM], bf[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) af[m] = *(const F*)(wb + (size_t)((wm * TM + m) * 32 + r) * LDX + kk);
#pragma unroll
        for (int n = 0; n < TN; ++n) bf[n] = *(const F*)(xs + (size_t)((wn * TN + n) * 32 + r + shift) * LDX + kk);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n) mma32w(acc[m][n], af[m], bf[n]);
      }
      if (more) {
        store_w(buf ^ 1, wnx);
        lds_barrier();
      }
      WSTAMP();
    }
    // ---- epilogue through LDS (whole channels-last rows per store), one wave column per pass
#pragma unroll
    for (int pass = 0; pass < WN; ++pass) {
      lds_barrier();
      WSTAMP();
      if (wn == pass) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n) {
            const int tl = n * 32 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int cl = (wm * TM + m) * 32 + 8 * g + 4 * h;
              *(f4*)(os + (size_t)tl * LDO + cl) = (f4){acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
            }
          }
      }
      WSTAMP();
      lds_barrier();
      WSTAMP();
      T4 ya[ACCU ? NI : 1];
#pragma unroll
      for (int e = 0; e < (ACCU ? NI : 0); ++e) {
        const int t = min(t0 + pass * PR + (tid + e * NT) / IPR, a.T_out - 1);
        ya[e] = *(const T4*)((const T*)a.y + (long long)t * a.ldy + ec);
      }
      // The activation code is a run-time argument: tested per ELEMENT (as the tile kernel does) it cost ~50 scalar
      // instructions and a branch per value -- 6.7 us of a 24 us tile, measured with in-kernel stamps.  Tested once per pass:
      auto items = [&](auto act_tag) {
#pragma unroll
        for (int e = 0; e < NI; ++e) {
          const int tl = (tid + e * NT) / IPR;
          const int t = t0 + pass * PR + tl;
          if (t >= a.T_out) continue;
          const f4 av = *(const f4*)(os + (size_t)tl * LDO + 4 * ecg);
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float u = av[j] + ebias[j];
            if (RES) u += (float)rv[e][j];
            u *= a.scale;
            u = post_act_c<decltype(act_tag)::value>(a.post_act, u);
            if (ACCU) u += (float)ya[e][j];
            v[j] = u;
          }
          *(T4*)((T*)a.y + (long long)t * a.ldy + ec) = (T4){(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        }
      };
      GSV_ACT_DISPATCH(a.post_act, items);
      if (RES && pass + 1 < WN) load_res_pass(pass + 1);
      WSTAMP();
    }
    F w0[WLOADS];                    // tap 0's slab again (taps 2, 4, ... went through its buffer): an L2 hit, ~1 us per tile
    load_w(0, w0);
    lds_barrier();                 // the fp32 tile (aliasing the window) has been read by every thread
    store_window(nxt);
    store_w(0, w0);
    lds_barrier();
    WSTAMP();
  }
}

}  // namespace

// returns 1 if the problem is not eligible (the caller goes on to the tile kernel), 0 on success, < 0 on error
int launch_conv_wide(int dtype, const ConvArgs& a, hipStream_t s) {
  static const bool off = getenv("GSV_NO_CONV_WIDE") != nullptr;          // A/B switch: one tile per workgroup (conv_lds_kernel)
  if (off || dtype != GSV_F16) return 1;
  if (a.Cin != 128 || a.Cout != 128 || a.Z != 1 || a.stride != 1 || a.ups_u > 0 || a.dil < 1 || a.gate || a.taps < 2) return 1;
  if (a.out_f32 || a.res_f32 || a.y_col0 != 0 || a.T_virt < 16384 || a.T_out != a.T_virt || a.T_in < 1) return 1;
  if (a.ldx % 8 != 0 || a.ldw % 8 != 0 || a.ldy % 4 != 0 || (a.res && a.ldr % 4 != 0) || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return 1;
  const int span = (a.taps - 1) * a.dil;
  if (span > 50 || a.pad < 0 || a.pad > span) return 1;
  const int rows_win = 256 + span;
  const int ntiles = cdiv(a.T_virt, 256);
  const size_t lds = ((size_t)306 + 2 * 128) * 136 * 2;
  const int grid = std::min(ntiles, 256);
  ConvArgs b = a;
  static unsigned long long* d_prof = nullptr;
  static int prof_calls = 0;
  if (getenv("GSV_WIDE_PROF") && !d_prof) { (void)hipMalloc((void**)&d_prof, 64 * 8); (void)hipMemset(d_prof, 0, 64 * 8); }
  b.prof = d_prof;
  const bool res = a.res != nullptr, acc = a.accumulate != 0;
  // 8 waves (round 3): 246 -> 222 us at 7 taps, 309 -> 291 us at 11 taps (tools/conv_probe.py), generator 10.87 -> 10.37 ms per bench
  // step in 2 of 2 alternating pairs, 0 / 10 spilled VGPRs instead of 0 / 22-58; GSV_WIDE_WAVES=4 restores round 2's geometry
  static const int nw = getenv("GSV_WIDE_WAVES") ? atoi(getenv("GSV_WIDE_WAVES")) : 8;
#define GSV_WIDE_NW(R, A, W)                                                                                               \
  do {                                                                                                                     \
    auto kern = conv_wide_f16_kernel<R, A, W>;                                                                             \
    static bool set = false;                                                                                               \
    if (!set) { GSV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(W * 64), lds, s, b, rows_win, ntiles);                                       \
  } while (0)
#define GSV_WIDE(R, A) do { if (nw == 8) GSV_WIDE_NW(R, A, 8); else GSV_WIDE_NW(R, A, 4); } while (0)
  if (res && acc) GSV_WIDE(true, true);
  else if (res) GSV_WIDE(true, false);
  else if (acc) GSV_WIDE(false, true);
  else GSV_WIDE(false, false);
#undef GSV_WIDE
#undef GSV_WIDE_NW
  GSV_HIP(hipGetLastError());
  if (d_prof && ++prof_calls == 3) {
    (void)hipStreamSynchronize(s);
    unsigned long long hp[64];
    (void)hipMemcpy(hp, d_prof, sizeof(hp), hipMemcpyDeviceToHost);
    fprintf(stderr, "[wide prof] taps %d:", a.taps);
    for (int i = 1; i < 24 && hp[i]; ++i) fprintf(stderr, " %.2f", (double)(hp[i] - hp[0]) / 100.0);
    fprintf(stderr, "\n");
  }
  return GSV_OK;
}

}  // namespace gsv

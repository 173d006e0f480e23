// One ResBlock pair of the HiFi-GAN generator's narrow stages in ONE kernel (gfx950, fp16):
//
//     y = ( convs2( lrelu( convs1( lrelu(x) ) ) ) + x ) * scale  [+ y]          (reference module/models.py:262-283: ResBlock1)
//
// convs1 has dilation d, convs2 dilation 1, both k taps, C -> C channels with C = 16 or 32 (the last two generator stages at
// 2 M / 4.1 M samples: 131 MB per tensor).  As two launches a pair moves 5 tensors through HBM (x in, t out | t in, x as
// residual, y out); fused it moves 2 (x once -- the residual rows come out of L2, where the window load left them -- and y).
// Construction = conv_narrow_f16_kernel (conv_lds.hip) twice inside one persistent 256-step tile:
//   convs1's weights in registers (A-fragments, loaded once per wave), convs2's in LDS
//   window  x[t0 - h2 - h1, t0 - h2 + 288 + h1) -> LDS (lrelu while staging), prefetched one tile ahead in registers
//   convs1 over 288 rows (9 MFMA column tiles dealt to the 4 waves) -> + bias -> fp16 -> lrelu -> zero outside [0, T)  -> LDS
//   convs2 over the tile's 256 rows from that LDS image -> fp32 tile transposed through LDS -> + bias + x, * scale, (+ y) -> rows
// Every rounding point of the two-launch path is kept (fp16 intermediate, fp16 lrelu, same tap / k order of the MFMAs): the
// result is bit-identical to it, except that with scale != 1 AND accumulate the compiler contracts the last two operations of
// the two epilogues differently (one fp16 ulp on ~0.02 % of the elements; tests/test_ops_gpu.py::test_conv_pair_matches_two_launches).
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace gsv {

namespace {

typedef _Float16 T;
typedef h8 F;
typedef _Float16 T4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma32p(f16v& acc, const h8& a, const h8& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ h8 lrelu8(h8 v, float s) { h8 t = v * (_Float16)s; return __builtin_elementwise_max(v, t); }
__device__ __forceinline__ T4 lrelu4(T4 v, float s) { T4 t = v * (_Float16)s; return __builtin_elementwise_max(v, t); }

constexpr int ROWS_Y = 288;      // intermediate rows per tile (9 x 32 >= 256 + 2 * 5)
constexpr int ROWS_X = 338;      // window rows at most (288 + 2 * 5 * 5)

// second launch bound = waves per SIMD the register allocation must leave room for: 2 workgroups per CU at C = 32, 3 at C = 16
template <int CC, int TAPS, bool ACCU>
__global__ __launch_bounds__(256, CC == 32 ? 2 : 3) void conv_pair_f16_kernel(ConvPairArgs a, int ntiles) {
  constexpr int G = 8, KC = 16, CT = 32, WN = 4, TN = 2, NT = 256;
  constexpr int LDX = CC + G, VPR = CC / G;
  constexpr int XB = (ROWS_X * VPR + NT - 1) / NT;
  // epilogue passes of 128 (C = 16) / 256 (C = 32) rows: the fp32 tile may take the window AND the intermediate image by then
  constexpr int WPP = CC == 16 ? 2 : 4, NP = WN / WPP;
  constexpr int LDO = CT + 4, PR = TN * 32 * WPP, IPR = CT / 4, NI = PR * IPR / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = (T*)smem;                                   // [ROWS_X][LDX]; the epilogue's fp32 [PR][LDO] tile aliases it
  T* ys = xs + (size_t)ROWS_X * LDX;                  // [ROWS_Y][LDX] lrelu(convs1(...)) of this tile
  T* w2s = ys + (size_t)ROWS_Y * LDX;                 // [TAPS][CT][LDX] convs2's weights; convs1's live in registers (below)
  float* os = (float*)smem;
  static_assert(PR * LDO * 4 <= (ROWS_X + ROWS_Y) * LDX * 2, "epilogue tile must fit in window + intermediate image");
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const T* __restrict__ x = a.x;
  const int h2 = (TAPS - 1) / 2, h1 = h2 * a.dil;
  const int rows_win = ROWS_Y + 2 * h1;
  const int total = rows_win * VPR;
  const int ecg = tid % IPR, ec = 4 * ecg;
  const int env = max(0, min(4, CC - ec));
  f4 ebias = (f4){0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < env; ++j) ebias[j] = a.b2[ec + j];
  // convs1's bias for the channels this lane holds in an accumulator: rows 8 g + 4 h + j
  f4 b1v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = 8 * g + 4 * h + j; b1v[g][j] = c < CC ? a.b1[c] : 0.f; }
  // ---- convs2's weights into LDS, convs1's into registers (every wave multiplies all of them by its own column tiles: TAPS *
  // CC / 16 A-fragments = 44 / 88 VGPRs at 11 taps; keeping them out of LDS is what lets 2 (C = 32) to 4 (C = 16) workgroups
  // share a CU -- with both slabs in LDS the C = 32 kernel ran one workgroup per CU and was slower than two launches)
  {
    const int totw = TAPS * CT * VPR;
    for (int v = tid; v < totw; v += NT) {
      const int tap = v / (CT * VPR), rem = v - tap * (CT * VPR);
      const int row = rem / VPR, col = rem - row * VPR;
      F val = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row < CC) val = *(const F*)(a.w2 + (long long)row * TAPS * CC + (long long)tap * CC + col * G);
      *(F*)(w2s + ((size_t)tap * CT + row) * LDX + col * G) = val;
    }
  }
  F w1r[TAPS * (CC / KC)];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int ks = 0; ks < CC / KC; ++ks) {
      F val = {0, 0, 0, 0, 0, 0, 0, 0};
      if (r < CC) val = *(const F*)(a.w1 + (long long)r * TAPS * CC + (long long)tap * CC + ks * KC + G * h);
      w1r[tap * (CC / KC) + ks] = val;
    }
  auto load_window = [&](int tile, F* regs) {
    const int win_start = tile * 256 - h2 - h1;
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int v = min(tid + i * NT, total - 1);
      const int row = v / VPR, col = v - row * VPR;
      const int ti = win_start + row;
      const F val = *(const F*)(x + (long long)min(max(ti, 0), a.T - 1) * a.ldx + col * G);
      regs[i] = (ti >= 0 && ti < a.T) ? val : (F){0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto store_window = [&](const F* regs) {
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int v = tid + i * NT;
      if (v < total) {
        const int row = v / VPR, col = v - row * VPR;
        *(F*)(xs + (size_t)row * LDX + col * G) = lrelu8(regs[i], 0.1f);
      }
    }
  };
  int tile = blockIdx.x;
  {
    F first[XB];
    load_window(min(tile, ntiles - 1), first);
    store_window(first);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * 256;
    // ---- requests for the NEXT tile's window and THIS tile's epilogue operands go out first
    F nxt[XB];
    load_window(min(tile + (int)gridDim.x, ntiles - 1), nxt);
    T4 rv[NP * NI], yv[ACCU ? NP * NI : 1];
#pragma unroll
    for (int q = 0; q < NP * NI; ++q) {
      const int pass = q / NI, e = q - pass * NI;
      const int t = min(t0 + pass * PR + (tid + e * NT) / IPR, a.T - 1);
      const int cc = min(ec, CC - 4);
      rv[q] = *(const T4*)(x + (long long)t * a.ldx + cc);
      if (ACCU) yv[q] = *(const T4*)(a.y + (long long)t * a.ldy + cc);
    }
    // ---- convs1 (dilation d) over the 288 intermediate rows: column tiles wn, wn + 4 and, wave 0, tile 8
    for (int n1 = wn; n1 < ROWS_Y / 32; n1 += WN) {
      f16v acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int shift = tap * a.dil;
#pragma unroll
        for (int ks = 0; ks < CC / KC; ++ks) {
          const int kk = ks * KC + G * h;
          const F bf = *(const F*)(xs + (size_t)(n1 * 32 + r + shift) * LDX + kk);
          mma32p(acc, w1r[tap * (CC / KC) + ks], bf);
        }
      }
      const int row = n1 * 32 + r, t = t0 - h2 + row;
      const bool inside = t >= 0 && t < a.T;             // convs2 pads its input with zeros, not with convs1 of padding
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 8 * g + 4 * h;
        if (c0 < CC) {
          T4 v = (T4){(T)(acc[4 * g] + b1v[g][0]), (T)(acc[4 * g + 1] + b1v[g][1]), (T)(acc[4 * g + 2] + b1v[g][2]),
                      (T)(acc[4 * g + 3] + b1v[g][3])};
          v = lrelu4(v, 0.1f);
          if (!inside) v = (T4){0, 0, 0, 0};
          *(T4*)(ys + (size_t)row * LDX + c0) = v;
        }
      }
    }
    __syncthreads();                                     // the intermediate image is complete; the window is dead
    // ---- convs2 (dilation 1) over the tile's 256 rows
    f16v acc2[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[n][i] = 0.f;
    for (int tap = 0; tap < TAPS; ++tap) {
      const T* wb = w2s + (size_t)tap * CT * LDX;
#pragma unroll
      for (int ks = 0; ks < CC / KC; ++ks) {
        const int kk = ks * KC + G * h;
        const F af = *(const F*)(wb + (size_t)r * LDX + kk);
        F bf[TN];
#pragma unroll
        for (int n = 0; n < TN; ++n) bf[n] = *(const F*)(ys + (size_t)((wn * TN + n) * 32 + r + tap) * LDX + kk);
#pragma unroll
        for (int n = 0; n < TN; ++n) mma32p(acc2[n], af, bf[n]);
      }
    }
    // ---- epilogue through LDS (whole channels-last rows per store), one wave column per pass
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      __syncthreads();
      if (wn / WPP == pass) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(f4*)(os + (size_t)(((wn % WPP) * TN + n) * 32 + r) * LDO + 8 * g + 4 * h) =
                (f4){acc2[n][4 * g], acc2[n][4 * g + 1], acc2[n][4 * g + 2], acc2[n][4 * g + 3]};
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < NI; ++e) {
        const int q = pass * NI + e;
        const int tl = (tid + e * NT) / IPR;
        const int t = t0 + pass * PR + tl;
        if (!(t < a.T && env > 0)) continue;
        const f4 av = *(const f4*)(os + (size_t)tl * LDO + 4 * ecg);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = av[j] + ebias[j];
          u += (float)rv[q][j];
          u *= a.scale;
          if (ACCU) u += (float)yv[q][j];
          v[j] = u;
        }
        T* yp = a.y + (long long)t * a.ldy + ec;
        if (env == 4) *(T4*)yp = (T4){(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        else for (int j = 0; j < env; ++j) yp[j] = (T)v[j];
      }
    }
    __syncthreads();                 // the fp32 tile (aliasing the window) has been read by every thread
    store_window(nxt);
    __syncthreads();
  }
}

template <int CC, int TAPS>
int launch_pair(const ConvPairArgs& a, hipStream_t s) {
  const int ntiles = cdiv(a.T, 256);
  const size_t lds = ((size_t)ROWS_X + ROWS_Y + (size_t)TAPS * 32) * (CC + 8) * 2;
  static const int cap = getenv("GSV_PAIR_PER_CU") ? std::max(1, atoi(getenv("GSV_PAIR_PER_CU"))) : 3;
  const int per_cu = std::max(1, std::min(cap, (int)((156 * 1024) / lds)));
  const int grid = std::min(ntiles, 256 * per_cu);
#define GSV_PAIR(A)                                                                                                        \
  do {                                                                                                                     \
    auto kern = conv_pair_f16_kernel<CC, TAPS, A>;                                                                         \
    static bool set = false;                                                                                               \
    if (!set) { GSV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, ntiles);                                                    \
  } while (0)
  if (a.accumulate) GSV_PAIR(true);
  else GSV_PAIR(false);
#undef GSV_PAIR
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

template <int CC>
int launch_pair_taps(const ConvPairArgs& a, hipStream_t s) {
  switch (a.taps) {
    case 3: return launch_pair<CC, 3>(a, s);
    case 5: return launch_pair<CC, 5>(a, s);
    case 7: return launch_pair<CC, 7>(a, s);
    case 9: return launch_pair<CC, 9>(a, s);
    default: return launch_pair<CC, 11>(a, s);
  }
}

}  // namespace

bool conv_pair_eligible(int dtype, int C, int taps, int dil, int T) {
  static const bool off = getenv("GSV_NO_CONV_PAIR") != nullptr;       // A/B switch: two launches per pair
  return !off && dtype == GSV_F16 && (C == 16 || C == 32) && (taps & 1) && taps <= 11 && dil >= 1 && ((taps - 1) / 2) * dil <= 25 &&
         T >= 256;
}

int launch_conv_pair(const ConvPairArgs& a, hipStream_t s) {
  GSV_REQUIRE(a.x && a.y && a.w1 && a.w2 && a.b1 && a.b2, "conv_pair: null operand");
  GSV_REQUIRE(conv_pair_eligible(GSV_F16, a.C, a.taps, a.dil, a.T) || getenv("GSV_NO_CONV_PAIR"), "conv_pair: shape C=%d taps=%d dil=%d T=%d not supported",
              a.C, a.taps, a.dil, a.T);
  GSV_REQUIRE(a.ldx % 8 == 0 && a.ldy % 4 == 0 && ((uintptr_t)a.x % 16) == 0 && ((uintptr_t)a.w1 % 16) == 0 && ((uintptr_t)a.w2 % 16) == 0,
              "conv_pair: operands must be 16-byte aligned");
  return a.C == 16 ? launch_pair_taps<16>(a, s) : launch_pair_taps<32>(a, s);
}

}  // namespace gsv

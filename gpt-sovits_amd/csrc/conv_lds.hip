// LDS-staged channels-last conv1d (stride 1) on MFMA -- the SoVITS generator's workhorse.
//
// conv_gemm.hip fetches every MFMA operand straight from L1/L2, which re-reads each input element
// once per tap and each weight once per wave; that is fine for small/odd shapes but leaves the
// generator's 90 ResBlock convs (95 % of SoVITS FLOPs) address-path bound.  Here a
// workgroup owns a (CT output channels x TT time steps) tile:
//   * the input window [TT + (taps-1)*|dil|][CC input channels] is staged ONCE per channel chunk into
//     LDS (leaky-relu applied while staging, zero rows outside the sequence), so every tap is an LDS
//     row shift instead of another global read -> HBM traffic drops to the algorithmic 1 read of x;
//   * the weight slab of one tap [CT][CC] is staged once per workgroup (not per wave) and
//     double-buffered: tap i+1 is in flight from L2 while tap i feeds the MFMAs, one barrier per tap;
//   * rows are padded by 16 B so ds_read_b128 of 16 consecutive rows hits 64 distinct banks;
//   * wave tile 64x128 (TM=2, TN=4 MFMA 32x32 tiles): 6 LDS fragment reads per 8 MFMAs.
// Epilogue (bias, residual, scale, accumulate, tanh/relu, polyphase scatter) is the one of conv_gemm.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "common.h"

namespace gsv {

template <typename T> struct FragL;
template <> struct FragL<_Float16> { typedef h8 type; };
template <> struct FragL<float> { typedef f4 type; };

__device__ __forceinline__ void mma32l(f16v& acc, const h8& a, const h8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32l(f16v& acc, const f4& a, const f4& b) {
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
}

template <typename F> __device__ __forceinline__ F zfrag() {
  F z;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(F) / sizeof(z[0])); ++i) z[i] = 0;
  return z;
}
__device__ __forceinline__ h8 lrelu_l(h8 v, float s) { h8 t = v * (_Float16)s; return __builtin_elementwise_max(v, t); }
__device__ __forceinline__ f4 lrelu_l(f4 v, float s) { f4 t = v * s; return __builtin_elementwise_max(v, t); }
__device__ __forceinline__ h8 relu_l(h8 v) { return __builtin_elementwise_max(v, zfrag<h8>()); }
__device__ __forceinline__ f4 relu_l(f4 v) { return __builtin_elementwise_max(v, zfrag<f4>()); }

template <typename T, int TM, int TN, int WM, int WN, int CC, bool ALLW, bool RES, bool ACCU>
__global__ __launch_bounds__(WM* WN * 64) void conv_lds_kernel(ConvArgs a, int rows_win, int lo) {
  typedef typename FragL<T>::type F;
  constexpr int G = DT<T>::G;        // elements per 16-byte chunk
  constexpr int KC = 2 * G;          // k per MFMA group
  constexpr int CT = WM * TM * 32;   // output channels per workgroup
  constexpr int TT = WN * TN * 32;   // time steps per workgroup
  constexpr int NT = WM * WN * 64;
  constexpr int LDX = CC + G;        // padded LDS row (elements)
  constexpr int VPR = CC / G;        // 16-byte vectors per staged row
  constexpr int WLOADS = (CT * VPR + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = (T*)smem;                                   // [rows_win][LDX]
  T* ws = xs + (size_t)rows_win * LDX;                // [2][CT][LDX]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int t0 = blockIdx.x * TT, cout0 = blockIdx.y * CT;
  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ w = (const T*)a.w;
  const int win_start = t0 - a.pad + lo;              // input row held in window row 0

  f16v acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // ---- epilogue operands are requested FIRST: every global round trip that is serialised inside a
  // tile costs ~2 us with one or two workgroups per CU, so the residual / accumulate tiles and the bias
  // ride in registers through the MFMA loop instead of being fetched (twice per pass) at the end.
  constexpr int LDO = CT + 4;
  constexpr int PR = TN * 32;                         // rows per epilogue pass (one wave column)
  constexpr int IPR = CT / 4;                         // 4-channel items per tile row
  constexpr int NI = PR * IPR / NT;                   // items per thread per pass
  static_assert(PR * IPR % NT == 0 && NT % IPR == 0, "tile/thread mapping");
  typedef T T4 __attribute__((ext_vector_type(4)));
  const bool vec_ok = ((a.ldy & 3) == 0) && ((a.y_col0 & 3) == 0) && ((a.ldr & 3) == 0) &&
                      (a.ups_u == 0 || (a.ups_cout & 3) == 0);
  const int ecg = tid % IPR;                          // this thread's channel group (same for all its items)
  const int ec = cout0 + 4 * ecg;
  int eoc = ec, epp = 0;
  if (a.ups_u > 0) { epp = ec / a.ups_cout; eoc = ec - epp * a.ups_cout; }
  const int env = max(0, min(4, a.Cout - ec));
  f4 ebias = (f4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) for (int j = 0; j < env; ++j) ebias[j] = a.bias[eoc + j];
  // the wide tile (CT = 128) keeps only the residual in registers; its accumulate operand is fetched per
  // pass (that tile has ~19 us of MFMA work, the narrow ones have <2 us and must not stall at all)
  constexpr bool PRE_ACC = ACCU && CT < 128;
  // wide tile: the residual of ONE epilogue pass is resident (pass 0 from kernel entry, pass p+1 requested when pass p has
  // been consumed); holding all WN passes (64 VGPRs beside 128 accumulator registers) spilled 272 B per lane
  constexpr bool RES_LAZY = RES && CT >= 128;
  constexpr int NRV = RES ? (RES_LAZY ? NI : WN * NI) : 1, NYV = PRE_ACC ? WN * NI : 1;
  T4 rv[NRV], yv[NYV];
  auto load_res_pass = [&](int pass) {
#pragma unroll
    for (int e = 0; e < NI; ++e) {
      rv[e] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
      const int tl = (tid + e * NT) / IPR;
      const int t = t0 + pass * PR + tl;
      const int orow = a.ups_u > 0 ? t * a.ups_u + epp - a.ups_pad : t;
      if (!(t < a.T_virt && env > 0 && orow >= 0 && orow < a.T_out)) continue;
      const T* rp = (const T*)a.res + (long long)orow * a.ldr + eoc;
      if (vec_ok && env == 4) rv[e] = *(const T4*)rp;
      else for (int j = 0; j < env; ++j) rv[e][j] = rp[j];
    }
  };
  rv[0] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
  yv[0] = rv[0];
  if (RES_LAZY) load_res_pass(0);
#pragma unroll
  for (int q = 0; q < (((RES && !RES_LAZY) || PRE_ACC) ? WN * NI : 0); ++q) {
    if (RES && !RES_LAZY) rv[q] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    if (PRE_ACC) yv[q] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    const int pass = q / NI, e = q - pass * NI;
    const int tl = (tid + e * NT) / IPR;
    const int t = t0 + pass * PR + tl;
    const int orow = a.ups_u > 0 ? t * a.ups_u + epp - a.ups_pad : t;
    const bool ok = t < a.T_virt && env > 0 && orow >= 0 && orow < a.T_out;
    if (!ok) continue;
    if (RES && !RES_LAZY) {
      const T* rp = (const T*)a.res + (long long)orow * a.ldr + eoc;
      if (vec_ok && env == 4) rv[q] = *(const T4*)rp;
      else for (int j = 0; j < env; ++j) rv[q][j] = rp[j];
    }
    if (PRE_ACC) {
      const T* yp = (const T*)a.y + (long long)orow * a.ldy + a.y_col0 + eoc;
      if (vec_ok && env == 4) yv[q] = *(const T4*)yp;
      else for (int j = 0; j < env; ++j) yv[q][j] = yp[j];
    }
  }

  // weight staging assignment: vector index v -> (row = v / VPR, col = v % VPR)
  auto load_w = [&](int tap, int cc0, F* regs) {
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int v = tid + i * NT;
      const int row = v / VPR, col = v - row * VPR;
      const int co = cout0 + row;
      const int ci = cc0 + col * G;
      regs[i] = (row < CT && co < a.Cout && ci < a.Cin)
                    ? *(const F*)(w + (long long)co * a.ldw + (long long)tap * a.Cin + ci)
                    : zfrag<F>();
    }
  };
  auto store_w = [&](int buf, const F* regs) {
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int v = tid + i * NT;
      const int row = v / VPR, col = v - row * VPR;
      if (row < CT) *(F*)(ws + ((size_t)buf * CT + row) * LDX + col * G) = regs[i];
    }
  };

  for (int cc0 = 0; cc0 < a.Cin; cc0 += CC) {
    __syncthreads();   // every wave is done with the previous chunk's window and weight buffers
    // ---- stage the input window of this channel chunk (pre-activation applied once here).
    // Loads are issued XB at a time before any LDS store so a thread has XB independent 16-byte
    // requests in flight (a load->store loop would serialise one HBM round trip per vector).
    {
      F w0[WLOADS];
      if (!ALLW) load_w(0, cc0, w0);
      constexpr int XB = (306 * VPR + NT - 1) / NT;   // the whole window (<= 306 rows) in one batch of independent loads
      const int total = rows_win * VPR;
      for (int v0 = 0; v0 < total; v0 += XB * NT) {
        F tmp[XB];
#pragma unroll
        for (int i = 0; i < XB; ++i) {
          const int v = v0 + tid + i * NT;
          const int row = v / VPR, col = v - row * VPR;
          const int ti = win_start + row;
          const int ci = cc0 + col * G;
          tmp[i] = (v < total && ti >= 0 && ti < a.T_in && ci < a.Cin) ? *(const F*)(x + (long long)ti * a.ldx + ci) : zfrag<F>();
        }
#pragma unroll
        for (int i = 0; i < XB; ++i) {
          const int v = v0 + tid + i * NT;
          if (v < total) {
            const int row = v / VPR, col = v - row * VPR;
            F val = tmp[i];
            if (a.pre_act == ACT_LRELU) val = lrelu_l(val, a.pre_slope);
            else if (a.pre_act == ACT_RELU) val = relu_l(val);
            *(F*)(xs + (size_t)row * LDX + col * G) = val;
          }
        }
      }
      if (!ALLW) store_w(0, w0);
      else {
        // narrow layers: every tap's weight slab fits in LDS, so the tap loop needs no barrier at all
        const int totw = a.taps * CT * VPR;
        for (int v0 = 0; v0 < totw; v0 += 4 * NT) {
          F tmp[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int v = v0 + tid + i * NT;
            const int tap = v / (CT * VPR), rem = v - tap * (CT * VPR);
            const int row = rem / VPR, col = rem - row * VPR;
            const int co = cout0 + row, ci = cc0 + col * G;
            tmp[i] = (v < totw && co < a.Cout && ci < a.Cin)
                         ? *(const F*)(w + (long long)co * a.ldw + (long long)tap * a.Cin + ci) : zfrag<F>();
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int v = v0 + tid + i * NT;
            if (v < totw) {
              const int tap = v / (CT * VPR), rem = v - tap * (CT * VPR);
              const int row = rem / VPR, col = rem - row * VPR;
              *(F*)(ws + ((size_t)tap * CT + row) * LDX + col * G) = tmp[i];
            }
          }
        }
      }
    }
    __syncthreads();
    const int ksteps = min(CC, a.Cin - cc0) / KC;
    for (int tap = 0; tap < a.taps; ++tap) {
      const int buf = ALLW ? tap : (tap & 1);
      F nxt[WLOADS];
      const bool more = !ALLW && tap + 1 < a.taps;
      if (more) load_w(tap + 1, cc0, nxt);
      const int shift = tap * a.dil - lo;             // window row of output column 0 for this tap
      const T* wb = ws + (size_t)buf * CT * LDX;
      for (int ks = 0; ks < ksteps; ++ks) {
        const int kk = ks * KC + G * h;
        F af[TM], bf[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) af[m] = *(const F*)(wb + (size_t)((wm * TM + m) * 32 + r) * LDX + kk);
#pragma unroll
        for (int n = 0; n < TN; ++n) bf[n] = *(const F*)(xs + (size_t)((wn * TN + n) * 32 + r + shift) * LDX + kk);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n) mma32l(acc[m][n], af[m], bf[n]);
      }
      if (more) {
        store_w(buf ^ 1, nxt);
        __syncthreads();
      }
    }
  }

  // ---- epilogue.  The MFMA accumulator layout gives a lane 4 channels of ONE time step, i.e. a wave
  // store would touch 32 different rows with 8-16 B each.  Instead the fp32 tile goes through LDS (one
  // wave column per pass) and is written as whole channels-last rows: consecutive lanes -> consecutive
  // 8-16 B, so HBM sees full lines.  No global load happens here (operands were preloaded above).
  float* os = (float*)smem;                           // [PR][LDO], reuses the staging buffers
#pragma unroll
  for (int pass = 0; pass < WN; ++pass) {
    __syncthreads();
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
    __syncthreads();
    T4 ya[NI];
    if (ACCU && !PRE_ACC) {
#pragma unroll
      for (int e = 0; e < NI; ++e) {
        ya[e] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
        const int tl = (tid + e * NT) / IPR;
        const int t = t0 + pass * PR + tl;
        const int orow = a.ups_u > 0 ? t * a.ups_u + epp - a.ups_pad : t;
        if (!(t < a.T_virt && env > 0 && orow >= 0 && orow < a.T_out)) continue;
        const T* yp = (const T*)a.y + (long long)orow * a.ldy + a.y_col0 + eoc;
        if (vec_ok && env == 4) ya[e] = *(const T4*)yp;
        else for (int j = 0; j < env; ++j) ya[e][j] = yp[j];
      }
    }
    // the activation code is tested once per pass, not per element (conv_wide.hip: ~50 scalar instructions per value otherwise)
    auto items = [&](auto act_tag) {
  #pragma unroll
      for (int e = 0; e < NI; ++e) {
        const int q = pass * NI + e;
        const int tl = (tid + e * NT) / IPR;
        const int t = t0 + pass * PR + tl;
        const int orow = a.ups_u > 0 ? t * a.ups_u + epp - a.ups_pad : t;
        if (!(t < a.T_virt && env > 0 && orow >= 0 && orow < a.T_out)) continue;
        const f4 av = *(const f4*)(os + (size_t)tl * LDO + 4 * ecg);
        float v[4];
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = av[j] + ebias[j];
          if (RES) u += to_f(rv[RES_LAZY ? e : q][j]);
          u *= a.scale;
          u = post_act_c<decltype(act_tag)::value>(a.post_act, u);
          if (PRE_ACC) u += to_f(yv[q][j]);
          else if (ACCU) u += to_f(ya[e][j]);
          v[j] = u;
        }
        const long long yoff = (long long)orow * a.ldy + a.y_col0 + eoc;
        const bool vec = vec_ok && env == 4;
        if (a.out_f32) {
          float* yp = (float*)a.y + yoff;
          if (vec) *(f4*)yp = (f4){v[0], v[1], v[2], v[3]};
          else for (int j = 0; j < env; ++j) yp[j] = v[j];
        } else {
          T* yp = (T*)a.y + yoff;
          if (vec) *(T4*)yp = (T4){(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
          else for (int j = 0; j < env; ++j) yp[j] = (T)v[j];
        }
      }
    };
    GSV_ACT_DISPATCH(a.post_act, items);
    if (RES_LAZY && pass + 1 < WN) load_res_pass(pass + 1);
  }
}

// ---------------------------------------------------------------------------------------
// Plain GEMM (taps == 1): Y[t][n] = X[t][:] . W[n][:], the prefill projections (M = sum of prompt
// lengths, K = 512 / 2048) and the 1x1 convs of enc_p.  128 (n) x 128 (t) tile, BK = 64, both
// operands double-buffered in LDS through registers: the loads of K-chunk c+1 are in flight while
// chunk c feeds 16 MFMAs per wave, one barrier per chunk; ~74 KB of LDS -> two workgroups per CU.
// ---------------------------------------------------------------------------------------
// NW = waves per workgroup: 4 (64 x 64 outputs per wave; two workgroups per CU when the grid is large enough) or 8 (64 x 32 per
// wave: for grids with fewer tiles than CUs, where a workgroup is alone on its CU and a second wave per SIMD hides its stalls)
#ifndef GSV_GEMM_W8_BK
#define GSV_GEMM_W8_BK 128
#endif
template <typename T, bool RES, bool WNT = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_lds_kernel(ConvArgs a) {
  typedef typename FragL<T>::type F;
  constexpr int G = DT<T>::G, KC = 2 * G;
  // K elements per chunk: 128 B per row (64 fp16 / 32 fp32); the 8-wave fp16 form (one workgroup per CU, 139 KB of LDS) stages
  // 256-B rows: half as many iterations, barriers and load round trips per tile
  constexpr int BK = (NW == 8 && sizeof(T) == 2 ? GSV_GEMM_W8_BK : 64 * 2 / (int)sizeof(T));
  constexpr int LDX = BK + G;
  constexpr int VPR = BK / G;                        // 8 vectors per row
  constexpr int CT = 128, TT = 128, NT = NW * 64;
  constexpr int NLD = CT * VPR / NT;                 // vectors per thread per operand per chunk (4 / 2)
  constexpr int TM = 2, TN = NW == 8 ? 1 : 2, WN = NW == 8 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = (T*)smem;                                  // [2][TT][LDX]
  T* ws = xs + 2 * TT * LDX;                         // [2][CT][LDX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order (unbatched launches): each XCD's L2 keeps a contiguous band of output-channel tiles' weights
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.z == 1 && a.xcd_order) {
    const int vid = xcd_virtual_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    if (a.xcd_order == 2) { by = vid % gridDim.y; bx = vid / gridDim.y; }      // an XCD gets all column tiles of a band of ROW tiles
    else { bx = vid % gridDim.x; by = vid / gridDim.x; }
  }
  const int t0 = bx * TT, cout0 = by * CT;
  const int z = blockIdx.z;                            // batch (attention heads): operand / output offsets
  const T* __restrict__ x = (const T*)a.x + (long long)z * a.xz;
  const T* __restrict__ w = (const T*)a.w + (long long)z * a.wz;
  const long long ybase = (long long)z * a.yz, rbase = (long long)z * a.rz;
  const int K = a.Cin;

  // epilogue operands first (see conv_lds_kernel)
  constexpr int LDO = CT + 4, PR = TN * 32, IPR = CT / 4, NI = PR * IPR / NT;
  typedef T T4 __attribute__((ext_vector_type(4)));
  const bool vec_ok = ((a.ldy & 3) == 0) && ((a.y_col0 & 3) == 0) && ((a.ldr & 3) == 0);
  const int ecg = tid % IPR, ec = cout0 + 4 * ecg;
  const int env = max(0, min(4, a.Cout - ec));
  f4 ebias = (f4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) for (int j = 0; j < env; ++j) ebias[j] = a.bias[z * a.bz + ec + j];
  f4 egate = (f4){1.f, 1.f, 1.f, 1.f};
  if (a.gate) for (int j = 0; j < env; ++j) egate[j] = a.gate[z * a.bz + ec + j];
  T4 rv[RES ? WN * NI : 1];
  rv[0] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
  if (RES) {
#pragma unroll
    for (int q = 0; q < WN * NI; ++q) {
      rv[q] = (T4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
      const int pass = q / NI, e = q - pass * NI;
      const int t = t0 + pass * PR + (tid + e * NT) / IPR;
      if (t < a.T_virt && env > 0) {
        const T* rp = (const T*)a.res + rbase + (long long)t * a.ldr + ec;
        if (vec_ok && env == 4) rv[q] = *(const T4*)rp;
        else for (int j = 0; j < env; ++j) rv[q][j] = rp[j];
      }
    }
  }

  f16v acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  auto load_tiles = [&](int k0, F* xr, F* wr) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int v = tid + i * NT;
      const int row = v / VPR, col = v - row * VPR;
      const int kk = k0 + col * G;
      const int t = t0 + row, co = cout0 + row;
      xr[i] = (t < a.T_in && kk < K) ? *(const F*)(x + (long long)t * a.ldx + kk) : zfrag<F>();
      if (WNT) wr[i] = (co < a.Cout && kk < K) ? __builtin_nontemporal_load((const F*)(w + (long long)co * a.ldw + kk)) : zfrag<F>();
      else wr[i] = (co < a.Cout && kk < K) ? *(const F*)(w + (long long)co * a.ldw + kk) : zfrag<F>();
    }
  };
  auto store_tiles = [&](int buf, F* xr, const F* wr) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int v = tid + i * NT;
      const int row = v / VPR, col = v - row * VPR;
      F val = xr[i];
      if (a.pre_act == ACT_LRELU) val = lrelu_l(val, a.pre_slope);
      else if (a.pre_act == ACT_RELU) val = relu_l(val);
      *(F*)(xs + ((size_t)buf * TT + row) * LDX + col * G) = val;
      *(F*)(ws + ((size_t)buf * CT + row) * LDX + col * G) = wr[i];
    }
  };
  // Two chunks in flight (round 3, the 8-wave variant = ONE workgroup per CU): chunk c is multiplied from LDS buffer c & 1 while chunk c + 1 waits in one register set and
  // chunk c + 2 is being requested into the other.  With ONE chunk of prefetch distance an iteration (16-32 MFMAs per wave,
  // ~0.2 us) could not be shorter than a global-load round trip (1-2 us under load): the K = 512 GEMMs of the AR prefill ran 8
  // such iterations per tile, 9 x their MFMA time.  The loop is unrolled by two so that the register sets keep static names.
  const int nchunks = (K + BK - 1) / BK;
  auto compute = [&](int buf) {
    const T* xb = xs + (size_t)buf * TT * LDX;
    const T* wb = ws + (size_t)buf * CT * LDX;
#pragma unroll
    for (int ks = 0; ks < BK / KC; ++ks) {
      const int kk = ks * KC + G * h;
      F af[TM], bf[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) af[m] = *(const F*)(wb + (size_t)((wm * TM + m) * 32 + r) * LDX + kk);
#pragma unroll
      for (int n = 0; n < TN; ++n) bf[n] = *(const F*)(xb + (size_t)((wn * TN + n) * 32 + r) * LDX + kk);
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) mma32l(acc[m][n], af[m], bf[n]);
    }
  };
  if constexpr (NW == 8) {
    F xa[NLD], wa[NLD], xq[NLD], wq[NLD];
    load_tiles(0, xa, wa);
    store_tiles(0, xa, wa);
    load_tiles(BK, xa, wa);                            // chunk 1 (zeros beyond K: load_tiles tests every element)
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      load_tiles((c + 2) * BK, xq, wq);
      compute(0);
      if (c + 1 >= nchunks) break;
      store_tiles(1, xa, wa);
      __syncthreads();
      load_tiles((c + 3) * BK, xa, wa);
      compute(1);
      if (c + 2 < nchunks) {
        store_tiles(0, xq, wq);
        __syncthreads();
      }
    }
  } else {
    // 4 waves: two workgroups per CU already keep two chunks in flight per CU, and the second register set would cost the second
    // workgroup (118 -> 214 VGPRs + 64 accumulators); measured: 39.3 vs 40.1 us on the prefill's QKV / FFN1 launches
    {
      F xr[NLD], wr[NLD];
      load_tiles(0, xr, wr);
      store_tiles(0, xr, wr);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const int buf = c & 1;
      F xr[NLD], wr[NLD];
      const bool more = c + 1 < nchunks;
      if (more) load_tiles((c + 1) * BK, xr, wr);
      compute(buf);
      if (more) {
        store_tiles(buf ^ 1, xr, wr);
        __syncthreads();
      }
    }
  }

  float* os = (float*)smem;                           // [PR][LDO]
#pragma unroll
  for (int pass = 0; pass < WN; ++pass) {
    __syncthreads();
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
    __syncthreads();
    if constexpr (sizeof(T) == 2 && !RES) {
      if (a.vt_out && cout0 >= a.vt_col0) {
        // V column tile: stored transposed, vt[col][t] (what attn.hip's vt_kernel produced from y); item = (column, 4 steps)
        constexpr int TG = PR / 4;                    // 4-step groups per pass
#pragma unroll
        for (int e = 0; e < NI; ++e) {
          const int idx = tid + e * NT;
          const int c = idx / TG, tg = idx - c * TG;
          const int co = cout0 + c;
          const int t = t0 + pass * PR + 4 * tg;
          if (co >= a.Cout || t >= a.vt_ld) continue;
          const float bz = a.bias ? a.bias[co] : 0.f;
          T4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (t + j < a.T_virt) ? (T)((os[(size_t)(4 * tg + j) * LDO + c] + bz) * a.scale) : (T)0.f;
          *(T4*)((T*)a.vt_out + (long long)(co - a.vt_col0) * a.vt_ld + t) = o;
        }
        continue;
      }
    }
    const bool rope_here = sizeof(T) == 2 && a.rope_cs != nullptr &&
                           ((ec >= a.rope_q0 && ec < a.rope_q0 + 2 * a.rope_half) || (ec >= a.rope_k0 && ec < a.rope_k0 + 2 * a.rope_half));
    // the activation code is tested once per pass, not per element (conv_wide.hip: ~50 scalar instructions per value otherwise)
    auto items = [&](auto act_tag) {
  #pragma unroll
      for (int e = 0; e < NI; ++e) {
        const int q = pass * NI + e;
        const int tl = (tid + e * NT) / IPR;
        const int t = t0 + pass * PR + tl;
        if (!(t < a.T_virt && env > 0)) continue;
        const f4 av = *(const f4*)(os + (size_t)tl * LDO + 4 * ecg);
        float v[4];
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = (av[j] + ebias[j]) * egate[j];
          if (RES) u += to_f(rv[q][j]);
          u *= a.scale;
          u = post_act_c<decltype(act_tag)::value>(a.post_act, u);
          v[j] = u;
        }
        if (rope_here) {
          // rotary embedding on the fp16-rounded values, as the separate kernel applied it to the stored projection
          const int p0 = (ec - (ec >= a.rope_k0 ? a.rope_k0 : a.rope_q0)) >> 1;
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const float cs = a.rope_cs[((long long)t * a.rope_half + p0 + pp) * 2], sn = a.rope_cs[((long long)t * a.rope_half + p0 + pp) * 2 + 1];
            const float x0 = (float)(T)v[2 * pp], x1 = (float)(T)v[2 * pp + 1];
            v[2 * pp] = x0 * cs - x1 * sn;
            v[2 * pp + 1] = x1 * cs + x0 * sn;
          }
        }
        const long long yoff = ybase + (long long)t * a.ldy + a.y_col0 + ec;
        const bool vec = vec_ok && env == 4 && ((ybase & 3) == 0);
        if (a.out_f32) {
          float* yp = (float*)a.y + yoff;
          if (vec) *(f4*)yp = (f4){v[0], v[1], v[2], v[3]};
          else for (int j = 0; j < env; ++j) yp[j] = v[j];
        } else {
          T* yp = (T*)a.y + yoff;
          if (vec) *(T4*)yp = (T4){(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
          else for (int j = 0; j < env; ++j) yp[j] = (T)v[j];
        }
      }
    };
    GSV_ACT_DISPATCH(a.post_act, items);
  }
}

template <typename T> static int try_launch_gemm(const ConvArgs& a, hipStream_t s) {
  constexpr int G = DT<T>::G;
  constexpr int BK = 64 * 2 / (int)sizeof(T);
  if ((a.vt_out || a.rope_cs) && (sizeof(T) != 2 || a.res || a.Z != 1 || a.vt_col0 % 128 != 0 || a.vt_ld % 4 != 0 || a.rope_q0 % 4 != 0 ||
                                  a.rope_k0 % 4 != 0 || a.out_f32)) {
    set_error("gemm: fused rotary / V^T epilogue needs fp16, no residual, tile-aligned V columns");
    return GSV_ERR_ARG;
  }
  if (a.taps != 1 || a.stride != 1 || a.ups_u > 0 || a.accumulate || a.pad != 0) return 1;
  if (a.T_virt < 512 || a.Cout < 96 || a.Cin % (2 * G) != 0 || a.Cin < BK) return 1;
  if (a.Z > 1 && ((a.xz % G) || (a.wz % G) || a.res)) return 1;      // batched: head slices must stay 16-byte aligned
  if (a.res && a.res_f32) return 1;
  if (a.ldx % G != 0 || a.ldw % G != 0 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return 1;
  size_t lds = (size_t)2 * (128 + 128) * (BK + G) * sizeof(T);           // 73.7 KB; epilogue tile 33.8 KB fits inside
  dim3 grid(cdiv(a.T_virt, 128), cdiv(a.Cout, 128), a.Z);
  static const bool xcd = !(getenv("GSV_GEMM_XCD") && getenv("GSV_GEMM_XCD")[0] == '0');     // A/B switch
  ConvArgs b = a;
  b.xcd_order = xcd ? 1 : 0;
  // Which operand an XCD's L2 should keep: every tile streams one activation panel [128][K] and one weight panel [128][K].
  // Order 1 gives an XCD a band of weight panels (all row tiles of a few column tiles); when the activations are the larger
  // operand and ALL weights fit an L2 anyway (prefill: 5760 x 2048 activations = 23.6 MB against 2 MB of weights), order 2 gives
  // it a band of row tiles with all their column tiles, so that an activation panel is fetched from memory once per XCD
  // instead of once per column tile (tools/gemm_probe.py, GSV_GEMM_XCD=1 restores order 1).
  static const bool only1 = getenv("GSV_GEMM_XCD") && getenv("GSV_GEMM_XCD")[0] == '1';
  if (xcd && !only1 && (long long)a.T_virt > 2LL * a.Cout && (size_t)a.Cout * a.Cin * sizeof(T) <= (size_t)3 << 20) b.xcd_order = 2;
  // 8 waves per workgroup where the grid has fewer tiles than the chip has CUs (prefill out-projection / FFN2: 180 tiles, the
  // DiT's QKV: 192, enc_p 1 x 1 convs): the workgroup is alone on its CU; GSV_GEMM_WAVES=4 restores round 2's geometry
  static const int gemm_waves = getenv("GSV_GEMM_WAVES") ? atoi(getenv("GSV_GEMM_WAVES")) : 8;
  static const long long w8_max_tiles = getenv("GSV_GEMM_W8_MAX_TILES") ? atoll(getenv("GSV_GEMM_W8_MAX_TILES")) : 256;   // A/B
  const bool w8 = sizeof(T) == 2 && gemm_waves == 8 && (long long)grid.x * grid.y * grid.z <= w8_max_tiles && a.Cin >= GSV_GEMM_W8_BK;
  if (w8) lds = (size_t)2 * (128 + 128) * (GSV_GEMM_W8_BK + G) * sizeof(T);       // 139 KB at 128-wide chunks
#define GSV_GEMM_LAUNCH(R, NTW, W)                                                                                         \
  do {                                                                                                                     \
    auto kern = gemm_lds_kernel<T, R, NTW, W>;                                                                             \
    static bool set = false;                                                                                               \
    if (!set) { GSV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
    hipLaunchKernelGGL(kern, grid, dim3(W * 64), lds, s, b);                                                               \
  } while (0)
  if (a.res) { if (w8) GSV_GEMM_LAUNCH(true, false, 8); else GSV_GEMM_LAUNCH(true, false, 4); }
  else if (a.w_nt) { if (w8) GSV_GEMM_LAUNCH(false, true, 8); else GSV_GEMM_LAUNCH(false, true, 4); }
  else { if (w8) GSV_GEMM_LAUNCH(false, false, 8); else GSV_GEMM_LAUNCH(false, false, 4); }
#undef GSV_GEMM_LAUNCH
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

template <typename T, int TM, int TN, int WM, int WN, int CC, bool ALLW, bool RES, bool ACCU>
static int launch_inst2(const ConvArgs& a, int rows_win, int lo, hipStream_t s) {
  constexpr int G = DT<T>::G;
  constexpr int CT = WM * TM * 32, TT = WN * TN * 32;
  size_t lds = ((size_t)rows_win + (ALLW ? a.taps : 2) * CT) * (CC + G) * sizeof(T);
  const size_t lds_epi = (size_t)TN * 32 * (CT + 4) * sizeof(float);
  if (lds_epi > lds) lds = lds_epi;
  auto kern = conv_lds_kernel<T, TM, TN, WM, WN, CC, ALLW, RES, ACCU>;
  static bool attr_set = false;
  if (!attr_set) {
    GSV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (lds > 160 * 1024) { set_error("conv_lds: window needs %zu B of LDS", lds); return GSV_ERR_ARG; }
  dim3 grid(cdiv(a.T_virt, TT), cdiv(a.Cout, CT), 1);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, a, rows_win, lo);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

template <typename T, int TM, int TN, int WM, int WN, int CC, bool ALLW>
static int launch_inst(const ConvArgs& a, int rows_win, int lo, hipStream_t s) {
  const bool res = a.res != nullptr, acc = a.accumulate != 0;
  if (res && acc) return launch_inst2<T, TM, TN, WM, WN, CC, ALLW, true, true>(a, rows_win, lo, s);
  if (res) return launch_inst2<T, TM, TN, WM, WN, CC, ALLW, true, false>(a, rows_win, lo, s);
  if (acc) return launch_inst2<T, TM, TN, WM, WN, CC, ALLW, false, true>(a, rows_win, lo, s);
  return launch_inst2<T, TM, TN, WM, WN, CC, ALLW, false, false>(a, rows_win, lo, s);
}

template <typename T, int CT, int CC> static bool fits(int rows, int wslabs = 2) {
  return ((size_t)rows + (size_t)wslabs * CT) * (CC + DT<T>::G) * sizeof(T) <= 160 * 1024;
}

// ---------------------------------------------------------------------------------------
// Narrow layers (C_in = C_out <= 32: the last two generator stages, 4.1 M and 2 M time steps of 16 / 32 channels).
// They are pure HBM streaming (131 MB per tensor, ~nothing to multiply), and one 256-step tile per workgroup made
// every tile a serial chain  load window -> LDS -> MFMA -> LDS transpose -> store  plus a reload of all 11 tap slabs:
// 184-244 us per conv against 52-79 us of HBM time.  Here a workgroup is PERSISTENT over tiles: the weights of all
// taps are staged once, and the NEXT tile's input window and epilogue operands are requested (unconditional, clamped
// addresses; zeros selected afterwards) before the current tile's MFMAs and epilogue, then written to LDS when the
// current tile is done -- the load round trip hides behind the previous tile's work.
// ---------------------------------------------------------------------------------------
template <int CC, int TM, int TN, int WN, bool RES, bool ACCU>
__global__ __launch_bounds__(64 * WN) void conv_narrow_f16_kernel(ConvArgs a, int rows_win, int ntiles) {
  typedef _Float16 T;
  typedef h8 F;
  typedef h4 T4;
  constexpr int G = 8, KC = 16, CT = 32 * TM, TT = 32 * TN * WN, NT = 64 * WN;
  static_assert(TT == 256, "tiles are 256 time steps");
  constexpr int LDX = CC + G, VPR = CC / G;
  constexpr int XB = (306 * VPR + NT - 1) / NT;
  // epilogue passes: as many wave columns per pass as the fp32 tile may take of the window's LDS (the smallest window is 258
  // rows).  One column per pass meant 2 barriers per column -- 16 per tile at 8 waves; with 128-row passes it is 4.
  constexpr int WPP = CC == 16 ? 1 : 128 / (TN * 32), NP = WN / WPP;
  constexpr int LDO = CT + 4, PR = TN * 32 * WPP, IPR = CT / 4, NI = PR * IPR / NT;
  static_assert(WN % WPP == 0 && (size_t)PR * LDO * 4 <= (size_t)258 * (CC + 8) * 2, "epilogue tile must fit in the smallest window");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = (T*)smem;                                   // [rows_win][LDX]; the epilogue's fp32 [PR][LDO] tile aliases it
  T* ws = xs + (size_t)rows_win * LDX;                // [taps][CT][LDX], staged once
  float* os = (float*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ w = (const T*)a.w;
  const int total = rows_win * VPR;
  const bool vec_ok = ((a.ldy & 3) == 0) && ((a.y_col0 & 3) == 0) && ((a.ldr & 3) == 0);
  const int ecg = tid % IPR, ec = 4 * ecg;
  const int env = max(0, min(4, a.Cout - ec));
  f4 ebias = (f4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) for (int j = 0; j < env; ++j) ebias[j] = a.bias[ec + j];
  // ---- all taps' weights, once per workgroup
  {
    const int totw = a.taps * CT * VPR;
    for (int v = tid; v < totw; v += NT) {
      const int tap = v / (CT * VPR), rem = v - tap * (CT * VPR);
      const int row = rem / VPR, col = rem - row * VPR;
      F val = zfrag<F>();
      if (row < a.Cout) val = *(const F*)(w + (long long)row * a.ldw + (long long)tap * a.Cin + col * G);
      *(F*)(ws + ((size_t)tap * CT + row) * LDX + col * G) = val;
    }
  }
  // window of tile `tile` -> registers: clamped (always valid) addresses, zero rows outside the sequence selected after
  auto load_window = [&](int tile, F* regs) {
    const int win_start = tile * TT - a.pad;
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int v = min(tid + i * NT, total - 1);
      const int row = v / VPR, col = v - row * VPR;
      const int ti = win_start + row;
      const F val = *(const F*)(x + (long long)min(max(ti, 0), a.T_in - 1) * a.ldx + col * G);
      regs[i] = (ti >= 0 && ti < a.T_in) ? val : zfrag<F>();
    }
  };
  auto store_window = [&](const F* regs) {
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int v = tid + i * NT;
      if (v < total) {
        const int row = v / VPR, col = v - row * VPR;
        F val = regs[i];
        if (a.pre_act == ACT_LRELU) val = lrelu_l(val, a.pre_slope);
        else if (a.pre_act == ACT_RELU) val = relu_l(val);
        *(F*)(xs + (size_t)row * LDX + col * G) = val;
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
    const int t0 = tile * TT;
    const bool pf = a.prof && blockIdx.x == 100 && tile == 100 + 3 * (int)gridDim.x && tid == 0;
    int pi = 0;
#define NSTAMP() do { if (pf) a.prof[pi++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    NSTAMP();
    // ---- requests for the NEXT tile's window and THIS tile's epilogue operands go out first
    F nxt[XB];
    load_window(min(tile + (int)gridDim.x, ntiles - 1), nxt);
    T4 rv[RES ? NP * NI : 1], yv[ACCU ? NP * NI : 1];
#pragma unroll
    for (int q = 0; q < ((RES || ACCU) ? NP * NI : 0); ++q) {
      const int pass = q / NI, e = q - pass * NI;
      const int t = min(t0 + pass * PR + (tid + e * NT) / IPR, a.T_out - 1);
      const int cc = min(ec, max(a.Cout - 4, 0));          // clamped channel group: loads stay in bounds; masked by env at use
      if (RES) rv[q] = *(const T4*)((const T*)a.res + (long long)t * a.ldr + cc);
      if (ACCU) yv[q] = *(const T4*)((const T*)a.y + (long long)t * a.ldy + a.y_col0 + cc);
    }
    NSTAMP();
    f16v acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    for (int tap = 0; tap < a.taps; ++tap) {
      const int shift = tap * a.dil;
      const T* wb = ws + (size_t)tap * CT * LDX;
#pragma unroll
      for (int ks = 0; ks < CC / KC; ++ks) {
        const int kk = ks * KC + G * h;
        F af[TM], bf[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) af[m] = *(const F*)(wb + (size_t)(m * 32 + r) * LDX + kk);
#pragma unroll
        for (int n = 0; n < TN; ++n) bf[n] = *(const F*)(xs + (size_t)((wn * TN + n) * 32 + r + shift) * LDX + kk);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n) mma32l(acc[m][n], af[m], bf[n]);
      }
    }
    NSTAMP();
    // ---- epilogue through LDS (whole channels-last rows per store), one wave column per pass
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      __syncthreads();
      NSTAMP();
      if (wn / WPP == pass) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *(f4*)(os + (size_t)(((wn % WPP) * TN + n) * 32 + r) * LDO + m * 32 + 8 * g + 4 * h) =
                  (f4){acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
      }
      NSTAMP();
      __syncthreads();
      NSTAMP();
      auto items = [&](auto act_tag) {      // activation code tested once per pass, not per element
  #pragma unroll
        for (int e = 0; e < NI; ++e) {
          const int q = pass * NI + e;
          const int tl = (tid + e * NT) / IPR;
          const int t = t0 + pass * PR + tl;
          if (!(t < a.T_virt && t < a.T_out && env > 0)) continue;
          const f4 av = *(const f4*)(os + (size_t)tl * LDO + 4 * ecg);
          float v[4];
  #pragma unroll
          for (int j = 0; j < 4; ++j) {
            float u = av[j] + ebias[j];
            if (RES) u += (float)rv[q][j];
            u *= a.scale;
            u = post_act_c<decltype(act_tag)::value>(a.post_act, u);
            if (ACCU) u += (float)yv[q][j];
            v[j] = u;
          }
          if (a.out_f32) {                 // conv_post: one fp32 output channel
            float* yp = (float*)a.y + (long long)t * a.ldy + a.y_col0 + ec;
            for (int j = 0; j < env; ++j) yp[j] = v[j];
          } else {
            T* yp = (T*)a.y + (long long)t * a.ldy + a.y_col0 + ec;
            if (vec_ok && env == 4) *(T4*)yp = (T4){(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
            else for (int j = 0; j < env; ++j) yp[j] = (T)v[j];
          }
        }
      };
      GSV_ACT_DISPATCH(a.post_act, items);
    }
    NSTAMP();
    __syncthreads();                 // the fp32 tile (aliasing the window) has been read by every thread
    NSTAMP();
    store_window(nxt);
    __syncthreads();
    NSTAMP();
  }
}

template <int CC, int TM, int TN, int WN>
static int launch_narrow(const ConvArgs& a, int rows_win, hipStream_t s) {
  const int ntiles = cdiv(a.T_virt, 256);
  const size_t lds = ((size_t)rows_win + (size_t)a.taps * 32 * TM) * (CC + 8) * 2;
  // resident workgroups per CU: as many as the LDS footprint allows, capped (GSV_NARROW_PER_CU, default 3).  These stages
  // are HBM-bound and every workgroup keeps one tile's window + operands in flight, so residency = bytes in flight.
  static const int cap = getenv("GSV_NARROW_PER_CU") ? std::max(1, atoi(getenv("GSV_NARROW_PER_CU"))) : 3;
  const int per_cu = std::max(1, std::min(cap, (int)((156 * 1024) / lds)));
  const int grid = std::min(ntiles, 256 * per_cu);
  const bool res = a.res != nullptr, acc = a.accumulate != 0;
  static unsigned long long* d_prof = nullptr;          // measurement runs (GSV_NARROW_PROF=1): in-kernel stamps of one tile
  static int prof_calls = 0;
  if (getenv("GSV_NARROW_PROF") && !d_prof) { (void)hipMalloc((void**)&d_prof, 64 * 8); (void)hipMemset(d_prof, 0, 64 * 8); }
  ConvArgs ap = a;
  ap.prof = d_prof;
#define GSV_NARROW(R, A)                                                                                                   \
  do {                                                                                                                     \
    auto kern = conv_narrow_f16_kernel<CC, TM, TN, WN, R, A>;                                                                         \
    static bool set = false;                                                                                               \
    if (!set) { GSV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN), lds, s, ap, rows_win, ntiles);                                         \
  } while (0)
  if (res && acc) GSV_NARROW(true, true);
  else if (res) GSV_NARROW(true, false);
  else if (acc) GSV_NARROW(false, true);
  else GSV_NARROW(false, false);
#undef GSV_NARROW
  GSV_HIP(hipGetLastError());
  if (d_prof && ++prof_calls == 3) {
    (void)hipStreamSynchronize(s);
    unsigned long long hp[64];
    (void)hipMemcpy(hp, d_prof, sizeof(hp), hipMemcpyDeviceToHost);
    fprintf(stderr, "[narrow prof] C %d taps %d (start | requests issued | taps done | epilogue done | barrier | window stored):", CC, a.taps);
    for (int i = 1; i < 20 && hp[i]; ++i) fprintf(stderr, " %.2f", (double)(hp[i] - hp[0]) / 100.0);
    fprintf(stderr, "\n");
  }
  return GSV_OK;
}

// returns 1 if the problem is not eligible (caller falls back to conv_gemm), 0 on success, <0 on error
template <typename T> static int try_launch(const ConvArgs& a, hipStream_t s) {
  constexpr int G = DT<T>::G;
  constexpr int KC = 2 * G;
  constexpr int CCBIG = 256 / (int)sizeof(T);   // 128 fp16 / 64 fp32 input channels per chunk
  if (a.Z != 1 || a.stride != 1 || a.Cin % KC != 0 || a.T_virt < 256 || a.gate) return 1;   // gate: 1x1 (gemm) epilogues only
  if ((a.res && a.res_f32) || (a.accumulate && a.out_f32)) return 1;   // preloaded operands are engine-dtype tiles
  if (a.ldx % G != 0 || a.ldw % G != 0 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return 1;
  const int span = (a.taps - 1) * (a.dil < 0 ? -a.dil : a.dil);
  if (span > 50) return 1;                      // staging batch is sized for windows of <= 306 rows
  const int lo = a.dil < 0 ? (a.taps - 1) * a.dil : 0;
  const int rows = 256 + span;                  // every configuration owns 256 time steps
  if (a.Cout > 64) {
    // mid-size problems (enc_p / flow convs over a few thousand frames: 50-150 tiles of 256 steps) cannot fill the
    // chip with 256-step tiles and each tile is a long dependent chain; halve the tile to double the workgroups
    // (SoVITS 23.9 -> 23.2 ms per bench step; 64-step tiles for the smallest grids measured no further gain)
    static const bool no_half = getenv("GSV_CONV_NO_HALF_TILE") != nullptr;      // A/B switch
    const long long wgs256 = (long long)cdiv(a.T_virt, 256) * cdiv(a.Cout, 128);
    static const bool half_always = getenv("GSV_CONV_HALF_ALWAYS") != nullptr;    // experiment switch
    // 8 waves (64 channels x 32 steps per wave) for the half tiles too: these grids have fewer workgroups than CUs, so a
    // workgroup is alone on its CU -- SoVITS device time 13.62 -> 13.2 ms per bench step; GSV_CONV_HALF_WAVES=4 restores
    static const int half_waves = getenv("GSV_CONV_HALF_WAVES") ? atoi(getenv("GSV_CONV_HALF_WAVES")) : 8;
    if (!no_half && (wgs256 < 192 || half_always)) {
      const int rows128 = 128 + span;
      if (sizeof(T) == 2 && half_waves == 8) {
        if (a.Cin >= CCBIG && fits<T, 128, CCBIG>(rows128)) return launch_inst<T, 2, 1, 2, 4, CCBIG, false>(a, rows128, lo, s);
        if (a.Cin >= CCBIG / 2 && fits<T, 128, CCBIG / 2>(rows128)) return launch_inst<T, 2, 1, 2, 4, CCBIG / 2, false>(a, rows128, lo, s);
      }
      if (a.Cin >= CCBIG && fits<T, 128, CCBIG>(rows128)) return launch_inst<T, 2, 2, 2, 2, CCBIG, false>(a, rows128, lo, s);
      if (a.Cin >= CCBIG / 2 && fits<T, 128, CCBIG / 2>(rows128)) return launch_inst<T, 2, 2, 2, 2, CCBIG / 2, false>(a, rows128, lo, s);
    }
    // 8 waves (64 channels x 64 steps per wave, two waves per SIMD: one wave's LDS reads and waits under the other's MFMAs) instead
    // of 4 (64 x 128): 256-channel generator convs 76.9 / 108.5 / 145.5 -> 58.6 / 89.6 / 127.2 us at 3 / 7 / 11 taps
    // (tools/conv_probe.py), generator 10.37 -> 10.05 ms per bench step; GSV_CONV_TILE_WAVES=4 restores round 2's geometry
    static const int tile_waves = getenv("GSV_CONV_TILE_WAVES") ? atoi(getenv("GSV_CONV_TILE_WAVES")) : 8;
    if (sizeof(T) == 2 && tile_waves == 8) {
      if (a.Cin >= CCBIG && fits<T, 128, CCBIG>(rows)) return launch_inst<T, 2, 2, 2, 4, CCBIG, false>(a, rows, lo, s);
      if (a.Cin >= CCBIG / 2 && fits<T, 128, CCBIG / 2>(rows)) return launch_inst<T, 2, 2, 2, 4, CCBIG / 2, false>(a, rows, lo, s);
    }
    if (a.Cin >= CCBIG && fits<T, 128, CCBIG>(rows)) return launch_inst<T, 2, 4, 2, 2, CCBIG, false>(a, rows, lo, s);
    if (a.Cin >= CCBIG / 2 && fits<T, 128, CCBIG / 2>(rows)) return launch_inst<T, 2, 4, 2, 2, CCBIG / 2, false>(a, rows, lo, s);
    return 1;
  }
  if (a.Cout > 32) {
    if (sizeof(T) == 2) {
      // persistent variant for the 64-channel stage: all 11 tap slabs resident = 145 KB of LDS = ONE workgroup per CU; run
      // with 8 waves (2 per SIMD, 32 columns each) so one wave's MFMAs overlap another's LDS reads
      static const bool no_persist64 = getenv("GSV_CONV_NO_PERSIST") != nullptr || getenv("GSV_CONV_NO_PERSIST64") != nullptr;
      const bool plain = a.ups_u == 0 && a.dil >= 1 && !a.out_f32 && !a.res_f32 && a.T_virt >= 16384 && a.T_out >= a.T_virt &&
                         a.T_in >= 1 && (a.Cout % 4 == 0) && (!a.res || a.ldr % 4 == 0) && a.ldy % 4 == 0 && a.y_col0 % 4 == 0;
      const size_t lds = ((size_t)rows + (size_t)a.taps * 64) * (64 + 8) * 2;
      if (!no_persist64 && plain && a.Cin == 64 && a.Cout <= 64 && lds <= 160 * 1024) return launch_narrow<64, 2, 1, 8>(a, rows, s);
    }
    if (a.Cin % 64 == 0 && fits<T, 64, 64>(rows)) return launch_inst<T, 2, 2, 1, 4, 64, false>(a, rows, lo, s);
    if (a.Cin % 32 == 0 && fits<T, 64, 32>(rows)) return launch_inst<T, 2, 2, 1, 4, 32, false>(a, rows, lo, s);
    return 1;
  }
  // persistent variant (fp16, one input chunk, plain stride-1 conv writing T-dtype rows): see conv_narrow_f16_kernel
  if (sizeof(T) == 2) {
    static const bool no_persist = getenv("GSV_CONV_NO_PERSIST") != nullptr;      // A/B switch
    const bool epi_free = !a.res && !a.accumulate;       // no vector operand loads: any Cout / fp32 output (conv_post) is fine
    const bool plain = a.ups_u == 0 && a.dil >= 1 && a.T_virt >= 4096 && a.T_out >= a.T_virt && a.T_in >= 1 &&
                       (epi_free || (!a.out_f32 && !a.res_f32 && a.Cout % 4 == 0 && (!a.res || a.ldr % 4 == 0) &&
                                     a.ldy % 4 == 0 && a.y_col0 % 4 == 0));
    if (!no_persist && plain && (a.Cin == 16 || a.Cin == 32)) {
      const size_t lds = ((size_t)rows + (size_t)a.taps * 32) * (a.Cin + 8) * 2;
      if (lds <= 64 * 1024) return a.Cin == 16 ? launch_narrow<16, 1, 2, 4>(a, rows, s) : launch_narrow<32, 1, 2, 4>(a, rows, s);
    }
  }
  // narrow layers (HBM-bound): all taps' weights resident in LDS, no barrier in the tap loop; small
  // register / LDS footprint so that several workgroups per CU overlap their single load round trip
  if (a.Cin % 64 == 0 && fits<T, 32, 64>(rows, a.taps)) return launch_inst<T, 1, 2, 1, 4, 64, true>(a, rows, lo, s);
  if (a.Cin % 32 == 0 && fits<T, 32, 32>(rows, a.taps)) return launch_inst<T, 1, 2, 1, 4, 32, true>(a, rows, lo, s);
  if (a.Cin % 16 == 0 && fits<T, 32, 16>(rows, a.taps)) return launch_inst<T, 1, 2, 1, 4, 16, true>(a, rows, lo, s);
  return 1;
}

int launch_conv_lds(int dtype, const ConvArgs& a, hipStream_t s) {
  int rc = 1;
  if (dtype == GSV_F16) rc = try_launch_gemm<_Float16>(a, s);
  else if (dtype == GSV_F32) rc = try_launch_gemm<float>(a, s);
  if (rc != 1) return rc;
  if (a.vt_out || a.rope_cs) { set_error("gemm: the fused rotary / V^T epilogue exists in gemm_lds_kernel only (shape not eligible)"); return GSV_ERR_ARG; }
  if (dtype == GSV_F16) return try_launch<_Float16>(a, s);
  if (dtype == GSV_F32) return try_launch<float>(a, s);
  return 1;
}

}  // namespace gsv

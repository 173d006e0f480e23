// Channels-last implicit-GEMM conv1d / GEMM on MFMA (gfx950).
//
//   y[t][co] = post( (sum_{tap,ci} pre(x[t*stride + tap*dil - pad][ci]) * w[co][tap*Cin+ci]
//                     + bias[co] + res[t][co]) * scale )
//
// One kernel serves every dense contraction outside the AR decode step: prefill
// projections, all SoVITS convs (dilated, strided, 1x1, and transposed convs restated as
// polyphase convs with a scatter epilogue), and the materialised attention products.
// Orientation: MFMA A = weights (rows = output channels), B = activations (cols = time), so
// a lane ends up holding 4 consecutive output channels of one time step -> vector stores
// into the channels-last output.  fp16 uses v_mfma_f32_32x32x16_f16 (fp32 accumulate);
// fp32 uses the exact-f32 v_mfma_f32_32x32x2_f32 with a permuted k order so that both
// operands are still fetched as 16-byte contiguous chunks.
#include <stdlib.h>

#include <stdio.h>

#include "common.h"

namespace gsv {

template <typename T> struct Frag;
template <> struct Frag<_Float16> { typedef h8 type; };
template <> struct Frag<float> { typedef f4 type; };

__device__ __forceinline__ void mma32(f16v& acc, const h8& a, const h8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f16v& acc, const f4& a, const f4& b) {
  // lane half h holds k = k0 + 4h + i; MFMA i contracts {k0+i, k0+4+i}: all 8 k covered once.
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
}

template <typename F> __device__ __forceinline__ F zero_frag() {
  F z;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(F) / sizeof(z[0])); ++i) z[i] = 0;
  return z;
}

__device__ __forceinline__ h8 lrelu_frag(h8 v, float slope) {
  h8 s = v * (_Float16)slope;
  return __builtin_elementwise_max(v, s);
}
__device__ __forceinline__ f4 lrelu_frag(f4 v, float slope) {
  f4 s = v * slope;
  return __builtin_elementwise_max(v, s);
}
__device__ __forceinline__ h8 relu_frag(h8 v) { return __builtin_elementwise_max(v, zero_frag<h8>()); }
__device__ __forceinline__ f4 relu_frag(f4 v) { return __builtin_elementwise_max(v, zero_frag<f4>()); }

template <typename T, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(ConvArgs a) {
  typedef typename Frag<T>::type F;
  constexpr int G = DT<T>::G;
  constexpr int KC = 2 * G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int z = blockIdx.z;
  const int cout0 = (blockIdx.y * WM + wm) * TM * 32;
  const int t0 = (blockIdx.x * WN + wn) * TN * 32;
  if (cout0 >= a.Cout || t0 >= a.T_virt) return;

  const T* __restrict__ x = (const T*)a.x + (long long)z * a.xz;
  const T* __restrict__ w = (const T*)a.w + (long long)z * a.wz;
  const int Kflat = a.taps * a.Cin;

  f16v acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // per-lane constant parts
  const T* wrow[TM];
  bool wok[TM];
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    int row = cout0 + 32 * m + r;
    wok[m] = row < a.Cout;
    wrow[m] = w + (long long)(wok[m] ? row : 0) * a.ldw;
  }
  int tbase[TN];
  bool tok[TN];
#pragma unroll
  for (int n = 0; n < TN; ++n) {
    int t = t0 + 32 * n + r;
    tok[n] = t < a.T_virt;
    tbase[n] = t * a.stride - a.pad;
  }

  int tap = 0, cin = G * h;
  while (cin >= a.Cin) { cin -= a.Cin; ++tap; }
  for (int kk = 0; kk < Kflat; kk += KC) {
    const int k = kk + G * h;
    const bool kok = k < Kflat;
    F af[TM], bf[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
      af[m] = (kok && wok[m]) ? *(const F*)(wrow[m] + k) : zero_frag<F>();
    const int shift = tap * a.dil;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      int ti = tbase[n] + shift;
      bool ok = kok && tok[n] && ti >= 0 && ti < a.T_in;
      F v = ok ? *(const F*)(x + (long long)ti * a.ldx + cin) : zero_frag<F>();
      if (a.pre_act == ACT_LRELU) v = lrelu_frag(v, a.pre_slope);
      else if (a.pre_act == ACT_RELU) v = relu_frag(v);
      bf[n] = v;
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) mma32(acc[m][n], af[m], bf[n]);
    cin += KC;
    while (cin >= a.Cin) { cin -= a.Cin; ++tap; }
  }

  // ---- epilogue: lane holds column t (time), rows (i&3)+8(i>>2)+4h (output channels)
  const bool has_act = a.post_act != ACT_NONE;        // one uniform test per value instead of post_act_f's chain of five
  const bool vec_ok = ((a.ldy & 3) == 0) && ((a.y_col0 & 3) == 0) && ((a.ldr & 3) == 0) &&
                      (a.ups_u == 0 || (a.ups_cout & 3) == 0);
#pragma unroll
  for (int m = 0; m < TM; ++m) {
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int t = t0 + 32 * n + r;
      if (t >= a.T_virt) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cout0 + 32 * m + 8 * g + 4 * h;
        if (c >= a.Cout) continue;
        int orow = t, oc = c;
        if (a.ups_u > 0) {
          int p = c / a.ups_cout;
          oc = c - p * a.ups_cout;
          orow = t * a.ups_u + p - a.ups_pad;
        }
        if (orow < 0 || orow >= a.T_out) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * g + j];
        const int nvalid = min(4, a.Cout - c);
        const long long yoff = (long long)z * a.yz + (long long)orow * a.ldy + a.y_col0 + oc;
        const long long roff = (long long)z * a.rz + (long long)orow * a.ldr + oc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j >= nvalid) break;
          float u = v[j];
          if (a.bias) u += a.bias[z * a.bz + oc + j];
          if (a.gate) u *= a.gate[z * a.bz + oc + j];
          if (a.res) u += a.res_f32 ? ((const float*)a.res)[roff + j] : to_f(((const T*)a.res)[roff + j]);
          u *= a.scale;
          if (has_act) u = post_act_f(a.post_act, u);
          if (a.accumulate) u += a.out_f32 ? ((float*)a.y)[yoff + j] : to_f(((T*)a.y)[yoff + j]);
          v[j] = u;
        }
        if (a.out_f32) {
          float* yp = (float*)a.y + yoff;
          if (vec_ok && nvalid == 4) *(f4*)yp = (f4){v[0], v[1], v[2], v[3]};
          else for (int j = 0; j < nvalid; ++j) yp[j] = v[j];
        } else {
          T* yp = (T*)a.y + yoff;
          if (vec_ok && nvalid == 4) {
            typedef T T4 __attribute__((ext_vector_type(4)));
            *(T4*)yp = (T4){(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
          } else {
            for (int j = 0; j < nvalid; ++j) yp[j] = (T)v[j];
          }
        }
      }
    }
  }
}

static long long small_tiles_below() {
  static const long long v = getenv("GSV_CONV_GEMM_SMALL_TILES") ? atoll(getenv("GSV_CONV_GEMM_SMALL_TILES")) : 64;   // A/B switch (0 = off)
  return v;
}

template <typename T> static int launch_t(const ConvArgs& a, hipStream_t s) {
  constexpr int G = DT<T>::G;
  GSV_REQUIRE(a.Cin % G == 0, "conv_gemm: Cin=%d must be a multiple of %d", a.Cin, G);
  GSV_REQUIRE(a.ldx % G == 0 && a.ldw % G == 0, "conv_gemm: ldx=%d / ldw=%d must be multiples of %d", a.ldx, a.ldw, G);
  GSV_REQUIRE(a.ldw >= a.taps * a.Cin, "conv_gemm: ldw too small");
  GSV_REQUIRE(((uintptr_t)a.x % 16) == 0 && ((uintptr_t)a.w % 16) == 0, "conv_gemm: x/w must be 16-byte aligned");
  GSV_REQUIRE(a.T_virt > 0 && a.Cout > 0 && a.Z > 0, "conv_gemm: empty problem");
  if (a.Cout <= 32) {
    dim3 grid(cdiv(a.T_virt, 512), cdiv(a.Cout, 32), a.Z);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 1, 4, 1, 4>), grid, dim3(256), 0, s, a);
  } else if (a.Cout <= 64 && (long long)cdiv(a.T_virt, 256) * a.Z < 192) {
    // under-filled grid (e.g. the DiT's grouped position conv: 16 groups x 4 time tiles): 64 x 64 tiles instead
    dim3 grid(cdiv(a.T_virt, 64), cdiv(a.Cout, 64), a.Z);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 1, 1, 2, 2>), grid, dim3(256), 0, s, a);
  } else if (a.Cout <= 64) {
    dim3 grid(cdiv(a.T_virt, 256), cdiv(a.Cout, 64), a.Z);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 2, 2, 1, 4>), grid, dim3(256), 0, s, a);
  } else if ((long long)cdiv(a.T_virt, 128) * cdiv(a.Cout, 128) * a.Z < small_tiles_below()) {
    // a handful of 128 x 128 tiles (single-utterance enc_p / flow convs: 200 frames x 384 channels = 6 workgroups, each
    // a serial chain over taps x Cin): 64 x 64 tiles put four times as many CUs on the same chain length
    dim3 grid(cdiv(a.T_virt, 64), cdiv(a.Cout, 64), a.Z);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 1, 1, 2, 2>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid(cdiv(a.T_virt, 128), cdiv(a.Cout, 128), a.Z);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 2, 2, 2, 2>), grid, dim3(256), 0, s, a);
  }
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int launch_conv_gemm(int dtype, const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  if (a.T_virt == 0) a.T_virt = a.T_out;
  if (a.ups_u > 0 && a.ups_cout == 0) { set_error("conv_gemm: ups_cout missing"); return GSV_ERR_ARG; }
  static const bool no_lds = getenv("GSV_NO_CONV_LDS") != nullptr;   // A/B switch for profiling
  {
    const int rc = launch_gemm_sk(dtype, a, s);
    if (rc <= 0) return rc;
  }
  if (no_lds && (a.vt_out || a.rope_cs)) { set_error("conv_gemm: fused QKV epilogue needs the LDS GEMM path"); return GSV_ERR_ARG; }
  if (!no_lds) {
    int rc = launch_conv_wide(dtype, a, s);
    if (rc <= 0) return rc;
    rc = launch_conv_lds(dtype, a, s);
    if (rc <= 0) return rc;
  }
  static const bool trace = getenv("GSV_TRACE_CONV_GEMM") != nullptr;      // which shapes take the direct-from-global fallback
  if (trace)
    fprintf(stderr, "[conv_gemm fallback] T_in %d T_virt %d Cin %d Cout %d taps %d stride %d dil %d Z %d res %d res_f32 %d acc %d out_f32 %d gate %d ups %d\n",
            a.T_in, a.T_virt, a.Cin, a.Cout, a.taps, a.stride, a.dil, a.Z, a.res != nullptr, a.res_f32, a.accumulate, a.out_f32,
            a.gate != nullptr, a.ups_u);
  if (dtype == GSV_F16) return launch_t<_Float16>(a, s);
  if (dtype == GSV_F32) return launch_t<float>(a, s);
  set_error("conv_gemm: bad dtype %d", dtype);
  return GSV_ERR_ARG;
}

}  // namespace gsv

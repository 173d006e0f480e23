// gsv HIP library -- shared helpers (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/gsv.h"

namespace gsv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);

#define GSV_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      gsv::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return GSV_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define GSV_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      gsv::set_error(__VA_ARGS__);             \
      return GSV_ERR_ARG;                      \
    }                                          \
  } while (0)

#define GSV_RC(e)            \
  do {                       \
    int _rc = (e);           \
    if (_rc) return _rc;     \
  } while (0)

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int id = GSV_F32; static constexpr int G = 4; };
template <> struct DT<_Float16> { static constexpr int id = GSV_F16; static constexpr int G = 8; };

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(_Float16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// XCD-aware workgroup order: the dispatcher deals linear workgroup ids round-robin to the 8 XCDs (id % 8), each with its
// own 4 MB L2.  This bijection gives XCD x the CONTIGUOUS run of virtual ids [x * n/8, (x+1) * n/8), so a kernel that
// decodes (tile row, tile column) from the virtual id keeps each XCD on its own slice of the weights / heads.
__device__ __forceinline__ int xcd_virtual_id(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
static inline size_t dt_size(int dt) { return dt == GSV_F16 ? 2 : 4; }

// ------------------------------------------------------------------------------------
// Workgroup barrier for LDS producer / consumer hand-offs.  __syncthreads() also emits `s_waitcnt vmcnt(0)`: it waits for every
// outstanding GLOBAL load and store of the wave -- in the persistent conv kernels that is the next tile's window prefetch and
// the previous tile's output stores, i.e. exactly the HBM time the persistence was meant to hide (same lesson as t2s_mega.hip).
// Only LDS traffic has to be complete here; results of global loads are waited for where they are used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// generic channels-last implicit-GEMM conv (conv_gemm.hip)
// ------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_LRELU = 3, ACT_MISH = 4, ACT_CLAMP1 = 5, ACT_SILU = 6, ACT_GELU = 7,
       ACT_GELU_TANH = 8, ACT_LRELU01 = 9, ACT_LOGCLAMP = 10 /* log(max(u, 1e-5)): mel_processing.py:8-14 */,
       ACT_RELU20 = 11 /* Hardtanh(0, 20): eres2net/ERes2NetV2.py:19-21 */, ACT_LOG_EPS = 12 /* log(max(u, FLT_EPSILON)): eres2net/kaldi.py:654 */ };

// Transcendental activations live in ONE out-of-line copy per translation unit: the conv epilogues are unrolled
// 16-64x, and inlining tanhf / expf / log1pf / erff into every instance grew the conv kernels by 26 % and made the
// accumulate / residual variants 2-3x slower (instruction-cache misses), measured with rocprofv3 on MI355X.
static __device__ __noinline__ float post_act_slow(int act, float u) {
  switch (act) {
    case ACT_TANH: { const float t = __expf(-2.f * fabsf(u)); return copysignf(__fdividef(1.f - t, 1.f + t), u); }   // = fast_tanh below
    case ACT_MISH: return u * tanhf(u > 20.f ? u : log1pf(expf(u)));   // x * tanh(softplus(x))
    case ACT_SILU: return u / (1.f + expf(-u));
    case ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
    case ACT_GELU_TANH: return 0.5f * u * (1.f + tanhf(0.79788456080286536f * (u + 0.044715f * u * u * u)));
    case ACT_LOGCLAMP: return logf(fmaxf(u, 1e-5f));
    case ACT_LOG_EPS: return logf(fmaxf(u, 1.1920928955078125e-07f));
    default: return u;
  }
}

__device__ __forceinline__ float post_act_f(int act, float u) {
  if (act == ACT_NONE) return u;
  if (act == ACT_RELU) return fmaxf(u, 0.f);
  if (act == ACT_CLAMP1) return fminf(fmaxf(u, -1.f), 1.f);
  if (act == ACT_LRELU01) return u > 0.f ? u : 0.01f * u;   // nn.LeakyReLU() default slope
  if (act == ACT_RELU20) return fminf(fmaxf(u, 0.f), 20.f);
  return post_act_slow(act, u);
}

// Epilogue activation with the code fixed at compile time for the two cases that matter for speed (none: every ResBlock
// conv; tanh: conv_post over 4.1 M samples), -1 = decide per value from the run-time code.  The kernels test the run-time
// code ONCE per epilogue pass and call the matching instance (conv_wide.hip has the measurement that led here).
__device__ __forceinline__ float fast_tanh(float u) {
  const float t = __expf(-2.f * fabsf(u));
  return copysignf(__fdividef(1.f - t, 1.f + t), u);
}
template <int ACTC> __device__ __forceinline__ float post_act_c(int act, float u) {
  if (ACTC == ACT_NONE) return u;
  if (ACTC == ACT_TANH) return fast_tanh(u);
  return post_act_f(act, u);
}
#define GSV_ACT_DISPATCH(act, fn)                                         \
  do {                                                                    \
    if ((act) == ACT_NONE) fn(std::integral_constant<int, ACT_NONE>{});   \
    else if ((act) == ACT_TANH) fn(std::integral_constant<int, ACT_TANH>{}); \
    else fn(std::integral_constant<int, -1>{});                           \
  } while (0)

struct ConvArgs {
  const void* x = nullptr;   // [Z][T_in][ldx] activations, channels-last
  const void* w = nullptr;   // [Z][Cout][ldw]  weights, K index = tap*Cin + cin (cin fastest)
  const float* bias = nullptr;  // [Cout_real] fp32 or null
  const float* gate = nullptr;  // [Cout_real] fp32 or null: v = ((acc + bias) * gate + res) * scale
  void* y = nullptr;         // [Z][T_out][ldy]
  const void* res = nullptr; // residual, same dtype/shape convention as y (ldr)
  int T_in = 0, T_out = 0;   // valid input rows / output rows actually stored
  int T_virt = 0;            // number of GEMM columns computed (== T_out unless ups_u > 0)
  int Cin = 0, Cout = 0, taps = 1;
  int stride = 1, dil = 1, pad = 0;  // input row = t*stride + tap*dil - pad
  int ldx = 0, ldw = 0, ldy = 0, ldr = 0;
  int y_col0 = 0;            // column offset into y rows (for writing channel slices)
  int pre_act = ACT_NONE;    // activation applied to x at load (ACT_LRELU uses pre_slope)
  float pre_slope = 0.1f;
  int post_act = ACT_NONE;   // activation applied after bias (+res)
  float scale = 1.0f;        // v = (acc + bias + res) * scale
  int accumulate = 0;        // y += v instead of y = v
  int out_f32 = 0;           // y is fp32 regardless of T
  int res_f32 = 0;           // res is fp32 regardless of T
  int ups_u = 0, ups_pad = 0, ups_cout = 0;  // transposed-conv scatter: row = s*u + n/ups_cout - pad
  int Z = 1;
  long long xz = 0, wz = 0, yz = 0, rz = 0;  // batch strides in elements
  unsigned long long* prof = nullptr;   // conv_wide.hip measurement runs (GSV_WIDE_PROF): s_memrealtime stamps of one tile
  int bz = 0;                                // batch stride of bias / gate (grouped convs)
  int xcd_order = 0;                         // set by the launcher: 1 = decode tile ids through xcd_virtual_id
  int w_nt = 0;                              // weights are loaded non-temporal (streamed once per step: keep them out of the
                                             // Infinity Cache so that OTHER layers' weights stay resident), GEMM paths only
  // DiT QKV projection (gemm_lds_kernel, fp16 only): the epilogue of the q / k column tiles applies the rotary embedding to the
  // first 2 * rope_half channels of q (columns rope_q0 ..) and k (rope_k0 ..), and the v column tiles (columns >= vt_col0) are
  // stored TRANSPOSED as vt[(col - vt_col0)][t] (leading dim vt_ld >= T_virt rounded up to 32, zeros beyond T_virt) instead of
  // into y -- what attn.hip's vt_kernel did in a launch of its own
  void* vt_out = nullptr;
  int vt_col0 = 0, vt_ld = 0;
  const float* rope_cs = nullptr;            // [T][rope_half][2] cos, sin
  int rope_half = 0, rope_q0 = 0, rope_k0 = 0;
};
int launch_conv_gemm(int dtype, const ConvArgs& a, hipStream_t s);
// split-K-in-workgroup streaming GEMM for under-filled grids (gemm_sk.hip): 0 = launched, 1 = not eligible, <0 = error
int launch_gemm_sk(int dtype, const ConvArgs& a, hipStream_t s);
// LDS-staged variant for stride-1 convs (conv_lds.hip): 0 = launched, 1 = not eligible, <0 = error
int launch_conv_lds(int dtype, const ConvArgs& a, hipStream_t s);
// persistent 128-channel tile convolution (conv_wide.hip): 1 = not eligible, 0 = launched, < 0 error
int launch_conv_wide(int dtype, const ConvArgs& a, hipStream_t s);

// fused ResBlock pair of the generator's narrow stages (conv_pair.hip): y = (convs2(lrelu(convs1(lrelu(x)))) + x) * scale [+ y]
struct ConvPairArgs {
  const _Float16* x = nullptr;     // [T][ldx] fp16, channels-last
  const _Float16* w1 = nullptr;    // [C][taps * C] tap-major, dilation `dil`
  const float* b1 = nullptr;
  const _Float16* w2 = nullptr;    // [C][taps * C], dilation 1
  const float* b2 = nullptr;
  _Float16* y = nullptr;           // [T][ldy]
  int T = 0, C = 0, taps = 0, dil = 1, ldx = 0, ldy = 0;
  float scale = 1.f;
  int accumulate = 0;
};
bool conv_pair_eligible(int dtype, int C, int taps, int dil, int T);
int launch_conv_pair(const ConvPairArgs& a, hipStream_t s);

// fused attention, fp16, head dim 64 (attn.hip); vt_buf: heads * 64 * ceil32(T) halfs of scratch
// rope_cs != null: rotate the first 2*rope_half channels of q and k in place first (cos|sin table [T][rope_half][2])
int launch_flash_attn64_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* vt_buf, int T, int heads,
                            float scale, void* out, int ldo, hipStream_t s, const float* rope_cs = nullptr, int rope_half = 0,
                            bool vt_ready = false);   // vt_ready: V^T (and the rotary embedding) already produced by the QKV GEMM's epilogue

// enc_p self-attention with window-4 relative positions, fp16, head dim 96 (attn.hip)
int launch_flash_rel96_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* vt_buf, int T, int heads,
                           float scale, const float* rel_k, const float* rel_v, void* out, int ldo, hipStream_t s);

// elementwise / small ops (ops.hip)
int launch_layernorm(int dtype, const void* x, int x_f32, const void* res, int res_f32, const float* gamma,
                     const float* beta, void* y, int y_f32, int rows, int C, float eps, hipStream_t s);
int launch_convert(const float* src, void* dst, int dtype, long long n, hipStream_t s);

}  // namespace gsv

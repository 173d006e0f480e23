// BigVGAN anti-aliased snake activation for gfx950 -- the HIP counterpart of the reference's one
// native kernel (BigVGAN/alias_free_activation/cuda/anti_alias_activation_cuda.cu:44-179).
//
//   up[n]  = 2 * sum_i xp[i] * uf[n + 15 - 2i]         xp = replicate-pad(x, 5, 5), 12-tap FIR, zero-stuffed x2
//   a[n]   = up[n] + sin^2(alpha * up[n]) / (beta + 1e-9)      alpha = exp(log_alpha[c]), beta = exp(log_beta[c])
//   y[t]   = sum_f df[f] * ap[2t + f]                  ap = replicate-pad(a, 5, 6), stride 2
//
// The reference gives each thread 32 consecutive outputs (lane stride 32 elements: uncoalesced) and
// keeps everything in registers.  Here a workgroup owns a (row, 2048-sample tile): x is read once
// with unit-stride lanes into LDS, the 2x-rate intermediate lives only in LDS, and y is written with
// unit-stride lanes, so HBM traffic is the algorithmic 1 read + 1 write per element (+0.8 % halo).
// Arithmetic is fp32 regardless of the I/O dtype (the reference computes in the I/O type).
#include "common.h"

namespace gsv {

constexpr int AA_TT = 2048;  // outputs per workgroup

template <typename T>
__global__ __launch_bounds__(256) void aa_act_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ up12,
                                                     const T* __restrict__ dn12, const T* __restrict__ log_alpha,
                                                     const T* __restrict__ log_beta, int C, int Tn) {
  __shared__ float xs[AA_TT + 16];
  __shared__ float as[2 * AA_TT + 16];
  __shared__ float uf[12], df[12];
  const int row = blockIdx.y;                 // b * C + c
  const int c = row % C;
  const int t0 = blockIdx.x * AA_TT;
  const int tid = threadIdx.x;
  const T* xr = x + (long long)row * Tn;
  if (tid < 12) { uf[tid] = to_f(up12[tid]); df[tid] = to_f(dn12[tid]); }
  const float alpha = expf(to_f(log_alpha[c]));
  const float inv_beta = 1.f / (expf(to_f(log_beta[c])) + 1e-9f);
  // x window: original indices t0-8 .. t0+TT+7 (clamped = replicate padding)
  for (int i = tid; i < AA_TT + 16; i += 256) {
    int t = min(max(t0 - 8 + i, 0), Tn - 1);
    xs[i] = to_f(xr[t]);
  }
  __syncthreads();
  // intermediate a[n] for n = 2*t0-8 .. 2*t0+2*TT+7 ; a index clamped to [0, 2T-1] (replicate padding of a)
  const int n0 = 2 * t0 - 8;
  for (int k = tid; k < 2 * AA_TT + 16; k += 256) {
    int n = min(max(n0 + k, 0), 2 * Tn - 1);
    // up[n] = 2 * sum_{i : 0 <= n+15-2i <= 11} xp[i] * uf[n+15-2i],  xp[i] = x[clamp(i-5)]
    const int ilo = (n + 5) >> 1;          // ceil((n+4)/2)
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = ilo + j;
      const int f = n + 15 - 2 * i;
      if (f >= 0 && f < 12) {
        int xo = min(max(i - 5, 0), Tn - 1);
        acc += xs[xo - (t0 - 8)] * uf[f];
      }
    }
    const float u = 2.f * acc;
    const float sn = sinf(u * alpha);
    as[k] = u + inv_beta * sn * sn;
  }
  __syncthreads();
  T* yr = y + (long long)row * Tn;
  for (int i = tid; i < AA_TT; i += 256) {
    const int t = t0 + i;
    if (t >= Tn) break;
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      // ap[2t+f] = a[clamp(2t+f-5)] ; as[] is indexed by (n - n0) and already holds clamped values
      const int n = 2 * t + f - 5;
      acc += df[f] * as[n - n0];
    }
    yr[t] = (T)acc;
  }
}

}  // namespace gsv

extern "C" int gsv_aa_act_forward(const void* x, void* y, const void* up12, const void* dn12, const void* log_alpha,
                                  const void* log_beta, int B, int C, int T, int dtype, gsv_stream_t stream) {
  using namespace gsv;
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "aa_act_forward: bad dtype %d", dtype);
  GSV_REQUIRE(B >= 0 && C >= 0 && T >= 0, "aa_act_forward: negative shape");
  if (B == 0 || C == 0 || T == 0) return GSV_OK;   // reference: silent return on seq_len == 0
  GSV_REQUIRE(x && y && up12 && dn12 && log_alpha && log_beta, "aa_act_forward: null pointer");
  GSV_REQUIRE((long long)B * C <= 65535, "aa_act_forward: B*C=%lld exceeds the grid limit", (long long)B * C);
  dim3 grid(cdiv(T, AA_TT), B * C);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSV_F16)
    hipLaunchKernelGGL(aa_act_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)x, (_Float16*)y, (const _Float16*)up12,
                       (const _Float16*)dn12, (const _Float16*)log_alpha, (const _Float16*)log_beta, C, T);
  else
    hipLaunchKernelGGL(aa_act_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)y, (const float*)up12,
                       (const float*)dn12, (const float*)log_alpha, (const float*)log_beta, C, T);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

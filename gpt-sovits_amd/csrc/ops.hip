// Small shared kernels: error state, dtype conversion, row LayerNorm.
#include "common.h"

namespace gsv {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

template <typename T> __global__ void convert_kernel(const float* __restrict__ s, T* __restrict__ d, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = (T)s[i];
}

int launch_convert(const float* src, void* dst, int dtype, long long n, hipStream_t s) {
  if (n <= 0) return GSV_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == GSV_F16) hipLaunchKernelGGL(convert_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, src, (_Float16*)dst, n);
  else hipLaunchKernelGGL(convert_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, n);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

// One wave per row.  y = LN(x + res) * gamma + beta, statistics in fp32 (two-pass).
template <typename TX, typename TR, typename TY>
__global__ void layernorm_kernel(const TX* __restrict__ x, const TR* __restrict__ res, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, TY* __restrict__ y, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TX* xr = x + (long long)row * C;
  const TR* rr = res ? res + (long long)row * C : nullptr;
  if (C <= 512) {
    // the row fits in 8 registers per lane: ONE pass over global memory instead of three dependent ones
    float v[8];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < C ? to_f(xr[c]) + (rr ? to_f(rr[c]) : 0.f) : 0.f;
      s1 += v[i];
    }
    const float mean1 = wave_sum(s1) / (float)C;
    float q1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = (lane + 64 * i < C) ? v[i] - mean1 : 0.f;
      q1 += d * d;
    }
    const float rstd1 = rsqrtf(wave_sum(q1) / (float)C + eps);
    TY* yr1 = y + (long long)row * C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      if (c < C) yr1[c] = (TY)((v[i] - mean1) * rstd1 * gamma[c] + beta[c]);
    }
    return;
  }
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += to_f(xr[c]) + (rr ? to_f(rr[c]) : 0.f);
  const float mean = wave_sum(sum) / (float)C;
  float var = 0.f;
  for (int c = lane; c < C; c += 64) {
    float d = to_f(xr[c]) + (rr ? to_f(rr[c]) : 0.f) - mean;
    var += d * d;
  }
  const float rstd = rsqrtf(wave_sum(var) / (float)C + eps);
  TY* yr = y + (long long)row * C;
  for (int c = lane; c < C; c += 64) {
    float v = to_f(xr[c]) + (rr ? to_f(rr[c]) : 0.f);
    yr[c] = (TY)((v - mean) * rstd * gamma[c] + beta[c]);
  }
}

template <typename TX, typename TR, typename TY>
static int ln_launch(const void* x, const void* res, const float* g, const float* b, void* y, int rows, int C, float eps,
                     hipStream_t s) {
  hipLaunchKernelGGL((layernorm_kernel<TX, TR, TY>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const TX*)x, (const TR*)res,
                     g, b, (TY*)y, rows, C, eps);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int launch_layernorm(int dtype, const void* x, int x_f32, const void* res, int res_f32, const float* gamma,
                     const float* beta, void* y, int y_f32, int rows, int C, float eps, hipStream_t s) {
  if (rows <= 0) return GSV_OK;
  const bool h = dtype == GSV_F16;
  const bool xf = x_f32 || !h, rf = res_f32 || !h, yf = y_f32 || !h;
  if (xf && rf && yf) return ln_launch<float, float, float>(x, res, gamma, beta, y, rows, C, eps, s);
  if (xf && rf && !yf) return ln_launch<float, float, _Float16>(x, res, gamma, beta, y, rows, C, eps, s);
  if (xf && !rf && !yf) return ln_launch<float, _Float16, _Float16>(x, res, gamma, beta, y, rows, C, eps, s);
  if (!xf && !rf && !yf) return ln_launch<_Float16, _Float16, _Float16>(x, res, gamma, beta, y, rows, C, eps, s);
  if (!xf && !rf && yf) return ln_launch<_Float16, _Float16, float>(x, res, gamma, beta, y, rows, C, eps, s);
  if (!xf && rf && !yf) return ln_launch<_Float16, float, _Float16>(x, res, gamma, beta, y, rows, C, eps, s);
  set_error("layernorm: unsupported dtype combination");
  return GSV_ERR_ARG;
}


// ---------------------------------------------------------------------------------------------------------------
// Reference-audio front-end helpers: framing of a waveform into a GEMM operand, STFT
// magnitude, per-channel normalisation over time (HuBERT's GroupNorm(512, 512)).  The GEMMs themselves are conv_gemm.
// ---------------------------------------------------------------------------------------------------------------
// out[t][k] = x[reflect(t * hop + k - pad)] for k < flen, 0 for flen <= k < ld   (torch.stft / F.pad(mode="reflect") framing;
// pad = 0: plain strided frames, the operand of a Conv1d(1, C, flen, stride = hop))
template <typename T>
__global__ void frame_kernel(const float* __restrict__ x, int n, int flen, int hop, int pad, int ld, int T_out, T* __restrict__ out) {
  const int t = blockIdx.x;
  if (t >= T_out) return;
  for (int k = threadIdx.x; k < ld; k += blockDim.x) {
    float v = 0.f;
    if (k < flen) {
      int i = t * hop + k - pad;
      if (i < 0) i = -i;                       // reflect without repeating the edge sample
      if (i >= n) i = 2 * (n - 1) - i;
      v = (i >= 0 && i < n) ? x[i] : 0.f;
    }
    out[(long long)t * ld + k] = (T)v;
  }
}

// spec[b][t] = sqrt(re^2 + im^2 + eps), ri [T][2 * bins] (re | im column blocks) -> spec [bins][T]  (mel_processing.py:73)
__global__ void magnitude_kernel(const float* __restrict__ ri, int T, int bins, float eps, float* __restrict__ spec) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, b = b0 + tx;
    float v = 0.f;
    if (t < T && b < bins) {
      const float re = ri[(long long)t * 2 * bins + b], im = ri[(long long)t * 2 * bins + bins + b];
      v = eps < 0.f ? re * re + im * im : sqrtf(re * re + im * im + eps);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int b = b0 + i, t = t0 + tx;
    if (b < bins && t < T) spec[(long long)b * T + t] = tile[tx][i];
  }
}

// same magnitude kept frame-major [T][ld] (zero up to ld): the left operand of the mel-filterbank GEMM
__global__ void magnitude_tm_kernel(const float* __restrict__ ri, int T, int bins, int ld, float eps, float* __restrict__ spec) {
  const int b = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  if (b >= ld) return;
  float v = 0.f;
  if (b < bins) {
    const float re = ri[(long long)t * 2 * bins + b], im = ri[(long long)t * 2 * bins + bins + b];
    v = eps < 0.f ? re * re + im * im : sqrtf(re * re + im * im + eps);
  }
  spec[(long long)t * ld + b] = v;
}

// attentional feature fusion mix (eres2net/fusion.py:22-27 with t = tanh(local_att)): out = x (1 + t) + y (1 - t)
__global__ void aff_mix_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ t, long long n,
                               float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float a = t[i]; out[i] = x[i] * (1.f + a) + y[i] * (1.f - a); }
}

// out[c] = mean over t of x[t][c]  (ERes2NetV2.forward3's temporal mean, ERes2NetV2.py:258); one thread per column, 64-column
// workgroups x 16 time slices reduced through LDS
__global__ __launch_bounds__(1024) void time_mean_kernel(const float* __restrict__ x, int T, int ld, float* __restrict__ out) {
  __shared__ float part[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < ld) for (int t = sl; t < T; t += 16) s += x[(long long)t * ld + c];
  part[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < ld) {
    float v = 0.f;
    for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x & 63];
    out[c] = v / (float)T;
  }
}

// y[t][c] = act((x[t][c] - mean_c) * rstd_c * gamma[c] + beta[c]), statistics over t (biased variance), channels-last.
// Two kernels: per-(channel block, time slice) partial sums -> finalize + apply.
template <typename T>
__global__ void cnorm_stats_kernel(const T* __restrict__ x, int Tn, int C, int slices, float* __restrict__ part /*[slices][2][C]*/) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, sl = blockIdx.y;
  if (c >= C) return;
  const int per = (Tn + slices - 1) / slices, t0 = sl * per, t1 = min(Tn, t0 + per);
  float s = 0.f, q = 0.f;
  for (int t = t0; t < t1; ++t) { const float v = to_f(x[(long long)t * C + c]); s += v; q += v * v; }
  part[((long long)sl * 2 + 0) * C + c] = s;
  part[((long long)sl * 2 + 1) * C + c] = q;
}
template <typename T>
__global__ void cnorm_apply_kernel(const T* __restrict__ x, int Tn, int C, int slices, const float* __restrict__ part,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act, T* __restrict__ y) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int sl = 0; sl < slices; ++sl) { s += part[((long long)sl * 2 + 0) * C + c]; q += part[((long long)sl * 2 + 1) * C + c]; }
  const float mean = (float)(s / Tn);
  const float var = fmaxf((float)(q / Tn - (s / Tn) * (s / Tn)), 0.f);
  const float a = rsqrtf(var + eps) * gamma[c], b = beta[c] - mean * a;
  const int per = (Tn + gridDim.y - 1) / gridDim.y, t0 = blockIdx.y * per, t1 = min(Tn, t0 + per);
  for (int t = t0; t < t1; ++t) y[(long long)t * C + c] = (T)post_act_f(act, to_f(x[(long long)t * C + c]) * a + b);
}

}  // namespace gsv

extern "C" {
const char* gsv_last_error(void) { return gsv::get_error(); }
int gsv_abi_version(void) { return 1; }
int gsv_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    gsv::set_error("gsv_init: no HIP device (%s)", hipGetErrorString(e));
    return GSV_ERR_HIP;
  }
  if (device < 0 || device >= n) {
    gsv::set_error("gsv_init: device %d out of range (%d devices)", device, n);
    return GSV_ERR_ARG;
  }
  GSV_HIP(hipSetDevice(device));
  hipDeviceProp_t p;
  GSV_HIP(hipGetDeviceProperties(&p, device));
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    gsv::set_error("gsv_init: device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
    return GSV_ERR_HIP;
  }
  return GSV_OK;
}

int gsv_op_layernorm(const void* x, const void* res, const float* gamma, const float* beta, void* y, int rows, int C,
                     float eps, int dtype, gsv_stream_t stream) {
  return gsv::launch_layernorm(dtype, x, 0, res, 0, gamma, beta, y, 0, rows, C, eps, (hipStream_t)stream);
}

int gsv_op_flash_attn64(const void* qkv, int T, int heads, float scale, void* vt_scratch, void* out, gsv_stream_t stream) {
  const int inner = heads * 64;
  return gsv::launch_flash_attn64_f16(qkv, 3 * inner, (const _Float16*)qkv + inner, 3 * inner, (const _Float16*)qkv + 2 * inner, 3 * inner,
                                      vt_scratch, T, heads, scale, out, inner, (hipStream_t)stream);
}

int gsv_op_flash_rel96(const void* qkv, int T, int heads, float scale, const float* rel_k, const float* rel_v, void* vt_scratch,
                       void* out, gsv_stream_t stream) {
  const int H = heads * 96;
  return gsv::launch_flash_rel96_f16(qkv, 3 * H, (const _Float16*)qkv + H, 3 * H, (const _Float16*)qkv + 2 * H, 3 * H, vt_scratch, T, heads,
                                     scale, rel_k, rel_v, out, H, (hipStream_t)stream);
}

int gsv_op_conv1d(const gsv_conv_desc* d, int dtype, gsv_stream_t stream) {
  gsv::ConvArgs a;
  a.x = d->x; a.w = d->w; a.bias = d->bias; a.y = d->y; a.res = d->res; a.gate = d->gate;
  a.T_in = d->T_in; a.T_out = d->T_out; a.Cin = d->Cin; a.Cout = d->Cout; a.taps = d->taps;
  a.stride = d->stride; a.dil = d->dil; a.pad = d->pad;
  a.ldx = d->Cin; a.ldw = d->taps * d->Cin;
  a.pre_act = d->pre_act; a.pre_slope = d->pre_slope; a.post_act = d->post_act; a.scale = d->scale;
  a.accumulate = d->accumulate; a.out_f32 = d->out_f32; a.res_f32 = d->out_f32;
  if (d->ups_u > 0) {
    // transposed conv restated as a polyphase conv: Cout = u * real_cout virtual channels
    a.ups_u = d->ups_u; a.ups_pad = d->ups_pad; a.ups_cout = d->Cout / d->ups_u;
    a.ldy = a.ups_cout; a.ldr = a.ups_cout;
    a.T_virt = d->T_in + d->taps - 1;
  } else {
    a.ldy = d->Cout; a.ldr = d->Cout;
    a.T_virt = d->T_out;
  }
  if (d->ldx > 0) a.ldx = d->ldx;
  if (d->ldw > 0) a.ldw = d->ldw;
  if (d->ldy > 0) { a.ldy = d->ldy; a.ldr = d->ldy; }
  if (d->ldr > 0) a.ldr = d->ldr;
  if (d->Z > 1) { a.Z = d->Z; a.xz = d->xz; a.wz = d->wz; a.yz = d->yz; a.bz = d->bz; a.rz = d->rz ? d->rz : d->yz; }
  return gsv::launch_conv_gemm(dtype, a, (hipStream_t)stream);
}

int gsv_op_frame(const float* x, int n, int frame_len, int hop, int pad, int ld, int T_out, void* out, int dtype, gsv_stream_t stream) {
  GSV_REQUIRE(x && out && n > 0 && frame_len > 0 && hop > 0 && ld >= frame_len && T_out > 0, "op_frame: bad argument");
  GSV_REQUIRE(pad >= 0 && pad < n, "op_frame: reflect padding %d needs more than %d samples", pad, pad);
  GSV_REQUIRE((long long)(T_out - 1) * hop + frame_len - pad <= (long long)n + pad, "op_frame: frames run past the (padded) signal");
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "op_frame: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSV_F16) hipLaunchKernelGGL(gsv::frame_kernel<_Float16>, dim3(T_out), dim3(256), 0, s, x, n, frame_len, hop, pad, ld, T_out, (_Float16*)out);
  else hipLaunchKernelGGL(gsv::frame_kernel<float>, dim3(T_out), dim3(256), 0, s, x, n, frame_len, hop, pad, ld, T_out, (float*)out);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_op_magnitude(const float* re_im, int T, int bins, float eps, int frame_ld, float* spec, gsv_stream_t stream) {
  GSV_REQUIRE(re_im && spec && T > 0 && bins > 0, "op_magnitude: bad argument");
  GSV_REQUIRE(frame_ld == 0 || frame_ld >= bins, "op_magnitude: frame_ld %d is smaller than %d bins", frame_ld, bins);
  if (frame_ld == 0)
    hipLaunchKernelGGL(gsv::magnitude_kernel, dim3(gsv::cdiv(T, 32), gsv::cdiv(bins, 32)), dim3(256), 0, (hipStream_t)stream, re_im, T, bins, eps, spec);
  else
    hipLaunchKernelGGL(gsv::magnitude_tm_kernel, dim3(gsv::cdiv(frame_ld, 256), T), dim3(256), 0, (hipStream_t)stream, re_im, T, bins, frame_ld, eps, spec);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_op_conv_pair(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int T, int C, int taps,
                     int dil, float scale, int accumulate, gsv_stream_t stream) {
  GSV_REQUIRE(gsv::conv_pair_eligible(GSV_F16, C, taps, dil, T), "op_conv_pair: C must be 16 or 32, taps odd <= 11, (taps - 1) / 2 * dil <= 25, T >= 256");
  gsv::ConvPairArgs a;
  a.x = (const _Float16*)x; a.w1 = (const _Float16*)w1; a.b1 = b1; a.w2 = (const _Float16*)w2; a.b2 = b2; a.y = (_Float16*)y;
  a.T = T; a.C = C; a.taps = taps; a.dil = dil; a.ldx = C; a.ldy = C; a.scale = scale; a.accumulate = accumulate;
  return gsv::launch_conv_pair(a, (hipStream_t)stream);
}

int gsv_op_aff_mix(const float* x, const float* y, const float* t, long long n, float* out, gsv_stream_t stream) {
  GSV_REQUIRE(x && y && t && out && n > 0, "op_aff_mix: bad argument");
  hipLaunchKernelGGL(gsv::aff_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, t, n, out);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_op_time_mean(const float* x, int T, int ld, float* out, gsv_stream_t stream) {
  GSV_REQUIRE(x && out && T > 0 && ld > 0, "op_time_mean: bad argument");
  hipLaunchKernelGGL(gsv::time_mean_kernel, dim3(gsv::cdiv(ld, 64)), dim3(1024), 0, (hipStream_t)stream, x, T, ld, out);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_op_channel_norm(const void* x, int T, int C, const float* gamma, const float* beta, float eps, int act, float* scratch,
                        void* y, int dtype, gsv_stream_t stream) {
  GSV_REQUIRE(x && y && gamma && beta && scratch && T > 0 && C > 0, "op_channel_norm: bad argument");
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "op_channel_norm: bad dtype");
  const int slices = 64;                      // scratch: 64 * 2 * C floats
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(gsv::cdiv(C, 64), slices);
  if (dtype == GSV_F16) {
    hipLaunchKernelGGL(gsv::cnorm_stats_kernel<_Float16>, grid, dim3(64), 0, s, (const _Float16*)x, T, C, slices, scratch);
    hipLaunchKernelGGL(gsv::cnorm_apply_kernel<_Float16>, grid, dim3(64), 0, s, (const _Float16*)x, T, C, slices, scratch, gamma, beta, eps, act, (_Float16*)y);
  } else {
    hipLaunchKernelGGL(gsv::cnorm_stats_kernel<float>, grid, dim3(64), 0, s, (const float*)x, T, C, slices, scratch);
    hipLaunchKernelGGL(gsv::cnorm_apply_kernel<float>, grid, dim3(64), 0, s, (const float*)x, T, C, slices, scratch, gamma, beta, eps, act, (float*)y);
  }
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}
}

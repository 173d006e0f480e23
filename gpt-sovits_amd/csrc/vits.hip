// SoVITS v2 waveform decoder engine (H6-H12) for gfx950.
//
// Every activation is channels-last [time][channel] in the engine dtype, so every conv /
// 1x1 / Linear / attention product is one call of the MFMA implicit-GEMM kernel
// (conv_gemm.hip).  Fusions: leaky-relu applied on operand load, bias / residual / scale /
// accumulate / tanh in the epilogue (ResBlock1 adds and the MRF mean never run as separate
// passes), transposed convs as polyphase convs with a scatter epilogue, weight-norm folded
// once at load (the reference re-materialises it every forward), speaker-conditioning
// terms (cond(ge), WN cond_layer(ge)) folded into biases once per reference audio.
// A single sequence is decoded per call (as in the reference, TTS.py:1266-1273 folds the
// batch into the time axis), so all x_mask terms are identically one and are dropped.
#include "engine.h"

namespace gsv {

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void gather_rows_kernel(const int* __restrict__ idx, const float* __restrict__ table, int C, int rep, int n,
                                   T* __restrict__ out) {
  // out[(i*rep + r)][c] = table[idx[i]][c]
  const int row = blockIdx.x;
  if (row >= n * rep) return;
  const float* src = table + (long long)idx[row / rep] * C;
  T* dst = out + (long long)row * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = (T)src[c];
}

// fp32 channels-first [C_total][T] -> T channels-last [T][C] (first C channels)
template <typename T>
__global__ void cf_to_cl_kernel(const float* __restrict__ src, int Tn, int C, T* __restrict__ dst, int ldd = 0) {
  if (ldd == 0) ldd = C;
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < Tn) ? src[(long long)c * Tn + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int t = t0 + i, c = c0 + tx;
    if (t < Tn && c < C) dst[(long long)t * ldd + c] = (T)tile[tx][i];
  }
}

// Anti-aliased snake / snakebeta on channels-last activations [T][C] (BigVGAN Activation1d,
// alias_free_activation/torch/act.py:25-30): 2x zero-stuffed 12-tap up-FIR -> x + sin^2(a x)/(b+1e-9)
// -> 12-tap stride-2 down-FIR, replicate padding as in aa.hip.  A workgroup owns 64 time steps x 64
// channels: rows are read/written 128 B wide (lane = channel), the 2x-rate intermediate lives in LDS.
template <typename T>
__global__ __launch_bounds__(256) void aa_act_cl_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, int C, int ld,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        int logscale, const float* __restrict__ up12,
                                                        const float* __restrict__ dn12) {
  constexpr int TT = 64, CW = 64;
  __shared__ float xs[TT + 16][CW];
  __shared__ float as[2 * TT + 16][CW];
  __shared__ float uf[12], df[12];
  const int t0 = blockIdx.x * TT, c0 = blockIdx.y * CW;
  const int cl = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const int c = c0 + cl;
  const bool cok = c < C;
  if (threadIdx.x < 12) { uf[threadIdx.x] = up12[threadIdx.x]; df[threadIdx.x] = dn12[threadIdx.x]; }
  float a = 1.f, ib = 1.f;
  if (cok) {
    a = logscale ? expf(alpha[c]) : alpha[c];
    ib = 1.f / ((logscale ? expf(beta[c]) : beta[c]) + 1e-9f);
  }
  for (int i = tq; i < TT + 16; i += 4) {
    const int t = min(max(t0 - 8 + i, 0), Tn - 1);
    xs[i][cl] = cok ? to_f(x[(long long)t * ld + c]) : 0.f;
  }
  __syncthreads();
  const int n0 = 2 * t0 - 8;
  for (int k = tq; k < 2 * TT + 16; k += 4) {
    const int n = min(max(n0 + k, 0), 2 * Tn - 1);
    const int ilo = (n + 5) >> 1;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = ilo + j;
      const int f = n + 15 - 2 * i;
      if (f >= 0 && f < 12) {
        const int xo = min(max(i - 5, 0), Tn - 1);
        acc += xs[xo - (t0 - 8)][cl] * uf[f];
      }
    }
    const float u = 2.f * acc;
    const float sn = sinf(u * a);
    as[k][cl] = u + ib * sn * sn;
  }
  __syncthreads();
  for (int i = tq; i < TT; i += 4) {
    const int t = t0 + i;
    if (t >= Tn || !cok) continue;
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < 12; ++f) acc += df[f] * as[2 * i + f + 3][cl];
    y[(long long)t * ld + c] = (T)acc;
  }
}

// T channels-last [T][ld] (cols col0..col0+C) -> fp32 channels-first [C][T]
template <typename TS>
__global__ void cl_to_cf_kernel(const TS* __restrict__ src, int Tn, int ld, int col0, int C, float* __restrict__ dst) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Tn * C) return;
  int c = (int)(i / Tn), t = (int)(i % Tn);
  dst[i] = to_f(src[(long long)t * ld + col0 + c]);
}

// Vt[z][c][j] = src[j][col0 + z*kc + c], zero padded to ldv columns
template <typename T>
__global__ void transpose_v_kernel(const T* __restrict__ src, int ld, int col0, int kc, int Tk, int ldv, T* __restrict__ vt) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z;
  const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    int j = j0 + i, c = c0 + tx;
    tile[i][tx] = (j < Tk && c < kc) ? to_f(src[(long long)j * ld + col0 + z * kc + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, j = j0 + tx;
    if (c < kc && j < ldv) vt[((long long)z * kc + c) * ldv + j] = (T)(j < Tk ? tile[tx][i] : 0.f);
  }
}

// Row softmax of fp32 scores [Z][Tq][Tk] -> P (T, row stride ldp, zero padded).  With a relative
// window (w > 0): scores[i][j] += qs_i . rel_k[j-i+w] for |j-i| <= w before the softmax
// (attentions.py:238-243, qs = q/sqrt(kc) is folded via `qscale`), and the 2w+1 band
// probabilities are kept in `band` for the relative-value term (attentions.py:253-256).
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ scores, int Tq, int Tk, int ldp,
                                                           T* __restrict__ P, const T* __restrict__ q, int ldq, int kc,
                                                           const float* __restrict__ rel_k, int w, float qscale,
                                                           float* __restrict__ band) {
  const int i = blockIdx.x, z = blockIdx.y;
  const float* srow = scores + ((long long)z * Tq + i) * Tk;
  T* prow = P + ((long long)z * Tq + i) * ldp;
  __shared__ float s_bias[16];
  __shared__ float s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (w > 0) {
    // 2w+1 dot products of length kc: one wave per offset (strided)
    for (int r = wave; r < 2 * w + 1; r += 4) {
      float s = 0.f;
      const T* qp = q + (long long)i * ldq + z * kc;
      for (int c = lane; c < kc; c += 64) s += to_f(qp[c]) * rel_k[r * kc + c];
      s = wave_sum(s);
      if (lane == 0) s_bias[r] = s * qscale;
    }
    __syncthreads();
  }
  // the biased row is kept in LDS (<= 12288 keys) so the fp32 scores are read from HBM once, not three times
  extern __shared__ float s_row[];
  const bool cached = Tk <= 12288;
  float m = -INFINITY;
  for (int j = tid; j < Tk; j += 256) {
    float v = srow[j];
    if (w > 0) { int r = j - i + w; if (r >= 0 && r <= 2 * w) v += s_bias[r]; }
    if (cached) s_row[j] = v;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float sum = 0.f;
  for (int j = tid; j < Tk; j += 256) {
    float v;
    if (cached) v = s_row[j];
    else {
      v = srow[j];
      if (w > 0) { int r = j - i + w; if (r >= 0 && r <= 2 * w) v += s_bias[r]; }
    }
    const float e = expf(v - m);
    if (cached) s_row[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) s_red[4 + wave] = sum;
  __syncthreads();
  sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
  const float inv = 1.f / sum;
  for (int j = tid; j < ldp; j += 256) {
    float p = 0.f;
    if (j < Tk) {
      int r = j - i + w;
      const bool inband = w > 0 && r >= 0 && r <= 2 * w;
      if (cached) p = s_row[j] * inv;
      else {
        float v = srow[j];
        if (inband) v += s_bias[r];
        p = expf(v - m) * inv;
      }
      if (inband) band[((long long)z * Tq + i) * (2 * w + 1) + r] = p;
    }
    prow[j] = (T)p;
  }
  if (w > 0 && tid < 2 * w + 1) {
    int j = i + tid - w;
    if (j < 0 || j >= Tk) band[((long long)z * Tq + i) * (2 * w + 1) + tid] = 0.f;
  }
}

// out[i][z*kc + c] += sum_r band[z][i][r] * rel_v[r][c]
template <typename T>
__global__ void relv_add_kernel(const float* __restrict__ band, const float* __restrict__ rel_v, int Tq, int kc, int nz,
                                int w, T* __restrict__ out, int ldo) {
  const int i = blockIdx.x;
  for (int e = threadIdx.x; e < nz * kc; e += blockDim.x) {
    int z = e / kc, c = e - z * kc;
    const float* b = band + ((long long)z * Tq + i) * (2 * w + 1);
    float s = 0.f;
    for (int r = 0; r < 2 * w + 1; ++r) s += b[r] * rel_v[r * kc + c];
    T* o = out + (long long)i * ldo + e;
    *o = (T)(to_f(*o) + s);
  }
}

// acts[t][c] = tanh(a[t][c]) * sigmoid(a[t][H + c])      (commons.py:96-103)
template <typename T>
__global__ void gate_kernel(const T* __restrict__ a, long long n, int H, T* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long t = i / H; int c = (int)(i - t * H);
  float x = to_f(a[t * 2 * H + c]), y = to_f(a[t * 2 * H + H + c]);
  out[i] = (T)(tanhf(x) * (1.f / (1.f + expf(-y))));
}

// x[t][c] += y[t][c] * sigmoid(y[t][H + c])                 (Conv1dGLU, modules.py:551-557)
template <typename T>
__global__ void glu_res_kernel(const T* __restrict__ y, long long n, int H, T* __restrict__ x) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long t = i / H; int c = (int)(i - t * H);
  float a = to_f(y[t * 2 * H + c]), b = to_f(y[t * 2 * H + H + c]);
  x[i] = (T)(to_f(x[i]) + a * (1.f / (1.f + expf(-b))));
}

template <typename T>
__global__ void flip_channels_kernel(const T* __restrict__ src, long long n, int C, T* __restrict__ dst) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long t = i / C; int c = (int)(i - t * C);
  dst[i] = src[t * C + (C - 1 - c)];
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// z_p[t][c] = m + noise * exp(logs) * scale   (models.py:1000); noise fp32 channels-first or counter RNG
template <typename T>
__global__ void zp_kernel(const float* __restrict__ stats, int F, int C, const float* __restrict__ noise, float scale,
                          unsigned long long seed, T* __restrict__ z) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)F * C) return;
  int t = (int)(i / C), c = (int)(i - (long long)t * C);
  float m = stats[(long long)t * 2 * C + c], ls = stats[(long long)t * 2 * C + C + c];
  float n;
  if (noise) n = noise[(long long)c * F + t];
  else {
    unsigned long long h1 = mix64(seed ^ mix64((unsigned long long)i * 2 + 1)), h2 = mix64(seed ^ mix64((unsigned long long)i * 2 + 2));
    float u1 = ((float)(h1 >> 40) + 1.0f) * (1.0f / 16777217.0f), u2 = (float)(h2 >> 40) * (1.0f / 16777216.0f);
    n = sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
  }
  z[i] = (T)(m + n * expf(ls) * scale);
}

// out[c] (+)= mean_t x[t][c] * wgt
template <typename T>
__global__ void mean_time_kernel(const T* __restrict__ x, int Tn, int C, float wgt, int accumulate, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int t = 0; t < Tn; ++t) s += to_f(x[(long long)t * C + c]) / (float)Tn;
  out[c] = (accumulate ? out[c] : 0.f) + s * wgt;
}

// F.interpolate(mode="linear", align_corners=False) along time on channels-last rows (models.py:226-228)
template <typename T>
__global__ void interp_linear_kernel(const T* __restrict__ x, int Tin, int Tout, int C, T* __restrict__ y) {
  const int t = blockIdx.x;
  const float scale = (float)Tin / (float)Tout;
  float src = scale * ((float)t + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  const int i0 = min((int)src, Tin - 1);
  const int i1 = min(i0 + 1, Tin - 1);
  const float w1 = src - (float)i0, w0 = 1.f - w1;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    y[(long long)t * C + c] = (T)(w0 * to_f(x[(long long)i0 * C + c]) + w1 * to_f(x[(long long)i1 * C + c]));
}

// F.interpolate(mode="nearest", scale_factor=sf) along time: src = min(floor(dst * (float)(1/sf)), Tin - 1)
template <typename T>
__global__ void interp_nearest_kernel(const T* __restrict__ x, int Tin, int Tout, int C, float scale, T* __restrict__ y) {
  const int t = blockIdx.x;
  const int src = min((int)floorf((float)t * scale), Tin - 1);
  for (int c = threadIdx.x; c < C; c += blockDim.x) y[(long long)t * C + c] = x[(long long)src * C + c];
}

// v2Pro: ge (+)= PReLU(ge_ref + sv_proj) * wgt   (module/models.py:971-975, then the mean over references)
__global__ void sv_prelu_acc_kernel(const float* __restrict__ ge_ref, const float* __restrict__ sv_proj, const float* __restrict__ a,
                                    float wgt, int accumulate, int n, float* __restrict__ ge) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = ge_ref[i] + sv_proj[i];
  const float u = (v >= 0.f ? v : a[i] * v) * wgt;
  ge[i] = accumulate ? ge[i] + u : u;
}

__global__ void vec_add_kernel(const float* a, const float* b, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// nearest codeword: argmax_j -(xx - 2 x.e_j + ee_j)   (core_vq.py:172-176)
__global__ void argmax_code_kernel(const float* __restrict__ dots, const float* __restrict__ x, const float* __restrict__ ee,
                                   int D, int NB, int* __restrict__ codes) {
  const int t = blockIdx.x, lane = threadIdx.x;
  float xx = 0.f;
  for (int c = lane; c < D; c += 64) { float v = x[(long long)t * D + c]; xx += v * v; }
  xx = wave_sum(xx);
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int j = lane; j < NB; j += 64) {
    float d = -(xx - 2.f * dots[(long long)t * NB + j] + ee[j]);
    if (d > best) { best = d; bi = j; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) codes[t] = bi;
}

}  // namespace gsv

using namespace gsv;

// =======================================================================================
// engine
// =======================================================================================

namespace gsveng {


int dalloc(gsv_vits* h, void** p, size_t bytes) {
  GSV_HIP(hipMalloc(p, bytes ? bytes : 16));
  h->allocs.push_back(*p);
  return GSV_OK;
}

int up_f32(gsv_vits* h, const float* v, size_t n, float** out) {
  GSV_RC(dalloc(h, (void**)out, n * 4));
  GSV_HIP(hipMemcpy(*out, v, n * 4, hipMemcpyHostToDevice));
  return GSV_OK;
}

int up_t(gsv_vits* h, const std::vector<float>& v, void** out) {
  if (h->dtype == GSV_F32) return up_f32(h, v.data(), v.size(), (float**)out);
  std::vector<_Float16> tmp(v.size());
  for (size_t i = 0; i < v.size(); ++i) tmp[i] = (_Float16)v[i];
  GSV_RC(dalloc(h, out, tmp.size() * 2));
  GSV_HIP(hipMemcpy(*out, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  return GSV_OK;
}

// fetch a (possibly weight-normed) tensor as fp32 host vector
bool fetch(gsv_vits* h, const std::string& name, size_t n, int dim0, std::vector<float>& out) {
  auto it = h->staged.find(name);
  if (it != h->staged.end()) {
    if (it->second.size() != n) { set_error("vits: tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.size(), n); return false; }
    out = it->second;
    return true;
  }
  // weight norm: name ends with ".weight" -> weight_g / weight_v (torch.nn.utils.weight_norm, dim=0)
  if (name.size() > 7 && name.compare(name.size() - 7, 7, ".weight") == 0) {
    auto ig = h->staged.find(name + "_g"), iv = h->staged.find(name + "_v");
    if (ig != h->staged.end() && iv != h->staged.end()) {
      if (iv->second.size() != n || (int)ig->second.size() != dim0) { set_error("vits: bad weight-norm pair for '%s'", name.c_str()); return false; }
      out.resize(n);
      const size_t per = n / dim0;
      for (int r = 0; r < dim0; ++r) {
        double ss = 0.0;
        const float* v = iv->second.data() + (size_t)r * per;
        for (size_t i = 0; i < per; ++i) ss += (double)v[i] * v[i];
        const float sc = ig->second[r] / (float)sqrt(ss);
        for (size_t i = 0; i < per; ++i) out[(size_t)r * per + i] = v[i] * sc;
      }
      return true;
    }
  }
  set_error("vits: missing tensor '%s'", name.c_str());
  return false;
}

// torch Conv1d weight [cout][cin][k] (+bias) -> Conv
int make_conv(gsv_vits* h, const std::string& name, int cout, int cin, int k, bool bias, Conv* c) {
  std::vector<float> w, b;
  if (!fetch(h, name + ".weight", (size_t)cout * cin * k, cout, w)) return GSV_ERR_ARG;
  std::vector<float> p((size_t)cout * k * cin);
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i)
      for (int j = 0; j < k; ++j) p[((size_t)o * k + j) * cin + i] = w[((size_t)o * cin + i) * k + j];
  GSV_RC(up_t(h, p, &c->w));
  if (bias) {
    if (!fetch(h, name + ".bias", cout, cout, b)) return GSV_ERR_ARG;
    GSV_RC(up_f32(h, b.data(), b.size(), &c->b));
  }
  c->cin = cin; c->cout = cout; c->taps = k;
  return GSV_OK;
}

// several 1x1 convs / Linears stacked along the output dim
int make_stacked(gsv_vits* h, const std::vector<std::string>& names, int cout_each, int cin, Conv* c) {
  std::vector<float> W, B;
  for (auto& n : names) {
    std::vector<float> w, b;
    if (!fetch(h, n + ".weight", (size_t)cout_each * cin, cout_each, w)) return GSV_ERR_ARG;
    if (!fetch(h, n + ".bias", cout_each, cout_each, b)) return GSV_ERR_ARG;
    W.insert(W.end(), w.begin(), w.end());
    B.insert(B.end(), b.begin(), b.end());
  }
  GSV_RC(up_t(h, W, &c->w));
  GSV_RC(up_f32(h, B.data(), B.size(), &c->b));
  c->cin = cin; c->cout = cout_each * (int)names.size(); c->taps = 1;
  return GSV_OK;
}

// ConvTranspose1d weight [cin][cout][k], stride u, padding (k-u)/2 -> polyphase conv with
// ceil(k/u) taps producing u*cout virtual channels (row p*cout+co), input x[s - q]
int make_ups(gsv_vits* h, const std::string& name, int cin, int cout, int k, int u, Conv* c) {
  std::vector<float> w, b;
  if (!fetch(h, name + ".weight", (size_t)cin * cout * k, cin, w)) return GSV_ERR_ARG;
  if (!fetch(h, name + ".bias", cout, cout, b)) return GSV_ERR_ARG;
  const int taps = (k + u - 1) / u;
  std::vector<float> p((size_t)u * cout * taps * cin, 0.f);
  for (int ph = 0; ph < u; ++ph)
    for (int q = 0; q < taps; ++q) {
      const int j = q * u + ph;
      if (j >= k) continue;
      for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
          p[(((size_t)ph * cout + co) * taps + q) * cin + ci] = w[((size_t)ci * cout + co) * k + j];
    }
  GSV_RC(up_t(h, p, &c->w));
  GSV_RC(up_f32(h, b.data(), b.size(), &c->b));
  c->cin = cin; c->cout = u * cout; c->taps = taps; c->ups_u = u; c->ups_pad = (k - u) / 2; c->ups_cout = cout;
  return GSV_OK;
}

int make_vec(gsv_vits* h, const std::string& name, size_t n, float** out) {
  std::vector<float> v;
  if (!fetch(h, name, n, (int)n, v)) return GSV_ERR_ARG;
  return up_f32(h, v.data(), n, out);
}

int make_encoder(gsv_vits* h, const std::string& prefix, int n_layers, std::vector<AttnLayerW>* out) {
  const auto& c = h->cfg;
  const int H = c.hidden_channels, FC = c.filter_channels, kc = H / c.n_heads;
  out->resize(n_layers);
  for (int i = 0; i < n_layers; ++i) {
    AttnLayerW& L = (*out)[i];
    const std::string a = prefix + ".attn_layers." + std::to_string(i) + ".";
    GSV_RC(make_stacked(h, {a + "conv_q", a + "conv_k", a + "conv_v"}, H, H, &L.qkv));
    GSV_RC(make_conv(h, a + "conv_o", H, H, 1, true, &L.o));
    GSV_RC(make_vec(h, a + "emb_rel_k", (size_t)9 * kc, &L.rel_k));
    GSV_RC(make_vec(h, a + "emb_rel_v", (size_t)9 * kc, &L.rel_v));
    GSV_RC(make_vec(h, prefix + ".norm_layers_1." + std::to_string(i) + ".gamma", H, &L.g1));
    GSV_RC(make_vec(h, prefix + ".norm_layers_1." + std::to_string(i) + ".beta", H, &L.b1));
    GSV_RC(make_vec(h, prefix + ".norm_layers_2." + std::to_string(i) + ".gamma", H, &L.g2));
    GSV_RC(make_vec(h, prefix + ".norm_layers_2." + std::to_string(i) + ".beta", H, &L.b2));
    const std::string f = prefix + ".ffn_layers." + std::to_string(i) + ".";
    GSV_RC(make_conv(h, f + "conv_1", FC, H, c.kernel_size, true, &L.f1));
    GSV_RC(make_conv(h, f + "conv_2", H, FC, c.kernel_size, true, &L.f2));
  }
  return GSV_OK;
}

int need(gsv_vits* h, const char* name, size_t bytes, void** out) {
  Buf& b = h->bufs[name];
  if (b.cap < bytes) {
    if (b.p) { GSV_HIP(hipDeviceSynchronize()); GSV_HIP(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t cap = bytes + bytes / 8 + 256;
    GSV_HIP(hipMalloc(&b.p, cap));
    b.cap = cap;
  }
  *out = b.p;
  return GSV_OK;
}


int conv(gsv_vits* h, hipStream_t s, const Conv& c, const void* x, int ldx, int T_in, void* y, int T_out, const ConvOpt& o) {
  ConvArgs a;
  a.x = x; a.y = y; a.res = o.res;
  const int cout = o.cout >= 0 ? o.cout : c.cout;
  a.w = (const char*)c.w + (size_t)o.w_row0 * c.taps * c.cin * esz(h);
  a.bias = o.no_bias ? nullptr : (o.bias_override ? o.bias_override : (c.b ? c.b + (c.ups_u ? 0 : o.w_row0) : nullptr));
  a.gate = o.gate;
  a.w_nt = o.w_nt;
  a.T_in = T_in; a.T_out = T_out; a.Cin = c.cin; a.Cout = cout; a.taps = c.taps;
  a.stride = o.stride; a.dil = o.dil;
  a.pad = o.pad >= 0 ? o.pad : (c.taps * o.dil - o.dil) / 2;
  a.ldx = ldx; a.ldw = c.taps * c.cin;
  a.pre_act = o.pre_act; a.pre_slope = o.pre_slope; a.post_act = o.post_act; a.scale = o.scale;
  a.accumulate = o.accumulate; a.out_f32 = o.out_f32; a.res_f32 = o.res_f32;
  if (c.ups_u > 0) {
    a.ups_u = c.ups_u; a.ups_pad = c.ups_pad; a.ups_cout = c.ups_cout;
    a.dil = -1; a.pad = 0; a.stride = 1;
    a.T_virt = T_in + c.taps - 1;
    a.ldy = o.ldy ? o.ldy : c.ups_cout;
  } else {
    a.T_virt = T_out;
    a.ldy = o.ldy ? o.ldy : cout;
  }
  a.ldr = o.ldr ? o.ldr : a.ldy;
  a.y_col0 = o.y_col0;
  a.vt_out = o.vt_out; a.vt_col0 = o.vt_col0; a.vt_ld = o.vt_ld;
  a.rope_cs = o.rope_cs; a.rope_half = o.rope_half; a.rope_q0 = o.rope_q0; a.rope_k0 = o.rope_k0;
  return launch_conv_gemm(h->dtype, a, s);
}


// materialised multi-head attention: q [Tq][ldq] cols qcol0.., k/v [Tk][ldkv] cols kcol0/vcol0..
// -> out [Tq][ldo] (heads concatenated).  rel_k/rel_v non-null: window-4 relative positions.
int attention(gsv_vits* h, hipStream_t s, const void* q, int ldq, int qcol0, const void* kv, int ldkv, int kcol0, int vcol0,
              int Tq, int Tk, int nh, int kc, float scale, const float* rel_k, const float* rel_v, void* out, int ldo) {
  const size_t es = esz(h);
  static const bool no_flash = getenv("GSV_MATERIALIZED_ENC_ATTN") != nullptr;    // A/B switch
  if (!no_flash && h->dtype == GSV_F16 && kc == 96 && rel_k && rel_v && Tq == Tk && q == kv && ldq == ldkv) {
    void* vtb;
    GSV_RC(need(h, "att_vt96", (size_t)nh * 96 * ((Tk + 31) / 32 * 32) * 2, &vtb));
    return launch_flash_rel96_f16((const _Float16*)q + qcol0, ldq, (const _Float16*)kv + kcol0, ldkv, (const _Float16*)kv + vcol0, ldkv,
                                  vtb, Tq, nh, scale, rel_k, rel_v, out, ldo, s);
  }
  const int G = h->dtype == GSV_F16 ? 8 : 4;
  const int ldp = (Tk + G - 1) / G * G;
  void *scores, *P, *Vt, *band;
  GSV_RC(need(h, "att_scores", (size_t)nh * Tq * Tk * 4, &scores));
  GSV_RC(need(h, "att_P", (size_t)nh * Tq * ldp * es, &P));
  GSV_RC(need(h, "att_Vt", (size_t)nh * kc * ldp * es, &Vt));
  GSV_RC(need(h, "att_band", (size_t)nh * Tq * 9 * 4 + 64, &band));
  ConvArgs a;
  a.x = (const char*)q + (size_t)qcol0 * es; a.w = (const char*)kv + (size_t)kcol0 * es; a.y = scores;
  a.T_in = Tq; a.T_out = Tq; a.T_virt = Tq; a.Cin = kc; a.Cout = Tk; a.taps = 1;
  a.ldx = ldq; a.ldw = ldkv; a.ldy = Tk; a.out_f32 = 1; a.scale = scale;
  a.Z = nh; a.xz = kc; a.wz = kc; a.yz = (long long)Tq * Tk;
  GSV_RC(launch_conv_gemm(h->dtype, a, s));
  const int w = rel_k ? 4 : 0;
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(softmax_rows_kernel<_Float16>, dim3(Tq, nh), dim3(256), Tk <= 12288 ? (size_t)Tk * 4 : 0, s, (const float*)scores, Tq, Tk, ldp, (_Float16*)P,
                       (const _Float16*)q + qcol0, ldq, kc, rel_k, w, scale, (float*)band),
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(Tq, nh), dim3(256), Tk <= 12288 ? (size_t)Tk * 4 : 0, s, (const float*)scores, Tq, Tk, ldp, (float*)P,
                       (const float*)q + qcol0, ldq, kc, rel_k, w, scale, (float*)band));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(transpose_v_kernel<_Float16>, dim3(cdiv(ldp, 32), cdiv(kc, 32), nh), dim3(256), 0, s, (const _Float16*)kv, ldkv,
                       vcol0, kc, Tk, ldp, (_Float16*)Vt),
    hipLaunchKernelGGL(transpose_v_kernel<float>, dim3(cdiv(ldp, 32), cdiv(kc, 32), nh), dim3(256), 0, s, (const float*)kv, ldkv, vcol0,
                       kc, Tk, ldp, (float*)Vt));
  ConvArgs b;
  b.x = P; b.w = Vt; b.y = out;
  b.T_in = Tq; b.T_out = Tq; b.T_virt = Tq; b.Cin = ldp; b.Cout = kc; b.taps = 1;
  b.ldx = ldp; b.ldw = ldp; b.ldy = ldo;
  b.Z = nh; b.xz = (long long)Tq * ldp; b.wz = (long long)kc * ldp; b.yz = kc;
  GSV_RC(launch_conv_gemm(h->dtype, b, s));
  if (rel_v) {
    GSV_DISPATCH(h,
      hipLaunchKernelGGL(relv_add_kernel<_Float16>, dim3(Tq), dim3(256), 0, s, (const float*)band, rel_v, Tq, kc, nh, 4, (_Float16*)out, ldo),
      hipLaunchKernelGGL(relv_add_kernel<float>, dim3(Tq), dim3(256), 0, s, (const float*)band, rel_v, Tq, kc, nh, 4, (float*)out, ldo));
  }
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

// attentions.Encoder.forward (attentions.py:64-84) on x [Tn][H] in place
int run_encoder(gsv_vits* h, hipStream_t s, std::vector<AttnLayerW>& layers, void* x, int Tn) {
  const auto& c = h->cfg;
  const int H = c.hidden_channels, FC = c.filter_channels, kc = H / c.n_heads;
  const size_t es = esz(h);
  void *qkv, *ao, *y, *ff;
  GSV_RC(need(h, "enc_qkv", (size_t)Tn * 3 * H * es, &qkv));
  GSV_RC(need(h, "enc_ao", (size_t)Tn * H * es, &ao));
  GSV_RC(need(h, "enc_y", (size_t)Tn * H * es, &y));
  GSV_RC(need(h, "enc_ff", (size_t)Tn * FC * es, &ff));
  for (auto& L : layers) {
    ConvOpt o;
    GSV_RC(conv(h, s, L.qkv, x, H, Tn, qkv, Tn, o));
    GSV_RC(attention(h, s, qkv, 3 * H, 0, qkv, 3 * H, H, 2 * H, Tn, Tn, c.n_heads, kc, 1.f / sqrtf((float)kc), L.rel_k, L.rel_v, ao, H));
    GSV_RC(conv(h, s, L.o, ao, H, Tn, y, Tn, o));
    GSV_RC(launch_layernorm(h->dtype, x, 0, y, 0, L.g1, L.b1, x, 0, Tn, H, 1e-5f, s));
    ConvOpt o1; o1.post_act = ACT_RELU;
    GSV_RC(conv(h, s, L.f1, x, H, Tn, ff, Tn, o1));
    GSV_RC(conv(h, s, L.f2, ff, FC, Tn, y, Tn, o));
    GSV_RC(launch_layernorm(h->dtype, x, 0, y, 0, L.g2, L.b2, x, 0, Tn, H, 1e-5f, s));
  }
  return GSV_OK;
}

void free_ctx(gsv_vits* h) {
  for (void* p : h->allocs) (void)hipFree(p);
  for (auto& b : h->bufs) if (b.second.p) (void)hipFree(b.second.p);
  h->allocs.clear();
  h->bufs.clear();
}

}  // namespace gsveng
using namespace gsveng;

namespace gsveng {

// quantizer.decode + nearest x2 (H8) and TextEncoder.forward up to (and including) the speed interpolation (H10, reference
// module/models.py:199-231): returns the hidden sequence y [F][hidden] (what `enc_p` returns as its first value)
int run_enc_p(gsv_vits* h, hipStream_t s, const int32_t* codes, int T, const int32_t* phones, int L, double speed, void** y_out,
              int* F_out) {
  const auto& c = h->cfg;
  const size_t es = esz(h);
  const int H = c.hidden_channels, SSL = c.ssl_dim, MH = 512;
  const int F0 = 2 * T;
  const int F = (speed == 1.0) ? F0 : (int)((double)F0 / speed) + 1;   // frames after the speed interpolation
  // ---- H8: codebook gather + nearest x2
  void *q768, *y, *tx;
  GSV_RC(need(h, "q768", (size_t)F0 * SSL * es, &q768));
  GSV_RC(need(h, "enc_x", (size_t)F0 * H * es, &y));
  GSV_RC(need(h, "enc_tx", (size_t)L * H * es, &tx));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(gather_rows_kernel<_Float16>, dim3(F0), dim3(128), 0, s, codes, h->codebook, SSL, 2, T, (_Float16*)q768),
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(F0), dim3(128), 0, s, codes, h->codebook, SSL, 2, T, (float*)q768));
  // ---- H10: enc_p
  ConvOpt o;
  GSV_RC(conv(h, s, h->ssl_proj_enc, q768, SSL, F0, y, F0, o));
  GSV_RC(run_encoder(h, s, h->enc_ssl, y, F0));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(gather_rows_kernel<_Float16>, dim3(L), dim3(128), 0, s, phones, h->text_emb, H, 1, L, (_Float16*)tx),
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(L), dim3(128), 0, s, phones, h->text_emb, H, 1, L, (float*)tx));
  GSV_RC(run_encoder(h, s, h->enc_text, tx, L));
  {  // MRTE (mrte_model.py:25-44)
    void *s512, *t512, *q512, *kv512, *o512, *x512;
    GSV_RC(need(h, "m_s", (size_t)F0 * MH * es, &s512));
    GSV_RC(need(h, "m_t", (size_t)L * MH * es, &t512));
    GSV_RC(need(h, "m_q", (size_t)F0 * MH * es, &q512));
    GSV_RC(need(h, "m_kv", (size_t)L * 2 * MH * es, &kv512));
    GSV_RC(need(h, "m_o", (size_t)F0 * MH * es, &o512));
    GSV_RC(need(h, "m_x", (size_t)F0 * MH * es, &x512));
    GSV_RC(conv(h, s, h->c_pre, y, H, F0, s512, F0, o));
    GSV_RC(conv(h, s, h->text_pre, tx, H, L, t512, L, o));
    GSV_RC(conv(h, s, h->mq, s512, MH, F0, q512, F0, o));
    GSV_RC(conv(h, s, h->mkv, t512, MH, L, kv512, L, o));
    GSV_RC(attention(h, s, q512, MH, 0, kv512, 2 * MH, 0, MH, F0, L, 4, MH / 4, 1.f / sqrtf((float)(MH / 4)), nullptr, nullptr, o512, MH));
    ConvOpt om; om.res = s512; om.ldr = MH; om.bias_override = h->mo_bias_eff;
    GSV_RC(conv(h, s, h->mo, o512, MH, F0, x512, F0, om));
    GSV_RC(conv(h, s, h->c_post, x512, MH, F0, y, F0, o));
  }
  GSV_RC(run_encoder(h, s, h->enc2, y, F0));
  if (F != F0) {
    void* yi;
    GSV_RC(need(h, "enc_x_speed", (size_t)F * H * es, &yi));
    GSV_DISPATCH(h,
      hipLaunchKernelGGL(interp_linear_kernel<_Float16>, dim3(F), dim3(64), 0, s, (const _Float16*)y, F0, F, H, (_Float16*)yi),
      hipLaunchKernelGGL(interp_linear_kernel<float>, dim3(F), dim3(64), 0, s, (const float*)y, F0, F, H, (float*)yi));
    y = yi;
  }
  *y_out = y;
  *F_out = F;
  return GSV_OK;
}

}  // namespace gsveng

extern "C" {

int gsv_vits_create(const gsv_vits_config* cfg, int dtype, gsv_vits_t** out) {
  GSV_REQUIRE(cfg && out, "vits_create: null argument");
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "vits_create: bad dtype");
  GSV_REQUIRE(cfg->flavor >= 0 && cfg->flavor <= 2, "vits_create: flavor must be 0 (v1/v2), 1 (v3) or 2 (v4)");
  GSV_REQUIRE(cfg->flavor != 0 || (cfg->n_ups >= 1 && cfg->n_ups <= 8 && cfg->n_resblocks >= 1 && cfg->n_resblocks <= 4), "vits_create: bad generator shape");
  GSV_REQUIRE(cfg->hidden_channels % cfg->n_heads == 0 && (cfg->hidden_channels / cfg->n_heads) % 8 == 0, "vits_create: head dim must be a multiple of 8");
  GSV_REQUIRE(cfg->inter_channels % 16 == 0, "vits_create: inter_channels must be a multiple of 16");
  int n = 0;
  GSV_HIP(hipGetDeviceCount(&n));
  gsv_vits* h = new gsv_vits();
  h->cfg = *cfg;
  h->dtype = dtype;
  *out = h;
  return GSV_OK;
}

void gsv_vits_destroy(gsv_vits_t* h) {
  if (!h) return;
  for (void* p : h->allocs) (void)hipFree(p);
  for (auto& b : h->bufs) if (b.second.p) (void)hipFree(b.second.p);
  for (auto e : h->ev) if (e) (void)hipEventDestroy(e);
  delete h;
}

int gsv_vits_load_tensor(gsv_vits_t* h, const char* name, const float* data, int64_t numel) {
  GSV_REQUIRE(h && name && data && numel > 0, "vits_load_tensor: bad argument");
  GSV_REQUIRE(!h->finalized, "vits_load_tensor: handle already finalized");
  h->staged[name].assign(data, data + numel);
  return GSV_OK;
}

int gsv_vits_finalize(gsv_vits_t* h) {
  GSV_REQUIRE(h && !h->finalized, "vits_finalize: bad handle");
  const auto& c = h->cfg;
  const int H = c.hidden_channels, IC = c.inter_channels, GIN = c.gin_channels, SSL = c.ssl_dim;
  const int MH = 512;  // MRTE hidden (mrte_model.py:13)
  GSV_REQUIRE(GIN == MH || c.v2pro, "vits: gin_channels must equal the MRTE width 512 (mrte_model.py:36 adds ge to it)");
  // enc_p
  GSV_RC(make_conv(h, "enc_p.ssl_proj", H, SSL, 1, true, &h->ssl_proj_enc));
  GSV_RC(make_encoder(h, "enc_p.encoder_ssl", c.n_layers / 2, &h->enc_ssl));
  GSV_RC(make_encoder(h, "enc_p.encoder_text", c.n_layers, &h->enc_text));
  GSV_RC(make_encoder(h, "enc_p.encoder2", c.n_layers / 2, &h->enc2));
  GSV_RC(make_vec(h, "enc_p.text_embedding.weight", (size_t)c.n_symbols * H, &h->text_emb));
  GSV_RC(make_conv(h, "enc_p.mrte.c_pre", MH, H, 1, true, &h->c_pre));
  GSV_RC(make_conv(h, "enc_p.mrte.text_pre", MH, H, 1, true, &h->text_pre));
  GSV_RC(make_conv(h, "enc_p.mrte.c_post", H, MH, 1, true, &h->c_post));
  GSV_RC(make_conv(h, "enc_p.mrte.cross_attention.conv_q", MH, MH, 1, true, &h->mq));
  GSV_RC(make_stacked(h, {"enc_p.mrte.cross_attention.conv_k", "enc_p.mrte.cross_attention.conv_v"}, MH, MH, &h->mkv));
  GSV_RC(make_conv(h, "enc_p.mrte.cross_attention.conv_o", MH, MH, 1, true, &h->mo));
  GSV_RC(make_conv(h, "enc_p.proj", 2 * IC, H, 1, true, &h->proj));
  // codebook + top-level ssl_proj
  {
    std::vector<float> e;
    if (!fetch(h, "quantizer.vq.layers.0._codebook.embed", (size_t)c.n_bins * SSL, c.n_bins, e)) return GSV_ERR_ARG;
    GSV_RC(up_f32(h, e.data(), e.size(), &h->codebook));
    GSV_RC(up_t(h, e, &h->codebook_t));
    std::vector<float> ee(c.n_bins);
    for (int j = 0; j < c.n_bins; ++j) {
      float s = 0.f;   // fp32 sum of squares in index order, like embed.pow(2).sum(0)
      for (int k = 0; k < SSL; ++k) s += e[(size_t)j * SSL + k] * e[(size_t)j * SSL + k];
      ee[j] = s;
    }
    GSV_RC(up_f32(h, ee.data(), ee.size(), &h->code_ee));
    GSV_RC(make_conv(h, "ssl_proj", SSL, SSL, 2, true, &h->top_ssl_proj));
  }
  if (c.flavor == 0) {
  // flow
  for (int fi = 0; fi < 4; ++fi) {
    FlowW& f = h->flows[fi];
    const std::string p = "flow.flows." + std::to_string(2 * fi);
    GSV_RC(make_conv(h, p + ".pre", H, IC / 2, 1, true, &f.pre));
    GSV_RC(make_conv(h, p + ".post", IC / 2, H, 1, true, &f.post));
    for (int li = 0; li < 4; ++li) {
      GSV_RC(make_conv(h, p + ".enc.in_layers." + std::to_string(li), 2 * H, H, 5, true, &f.wn.in[li]));
      const int rs = li < 3 ? 2 * H : H;
      GSV_RC(make_conv(h, p + ".enc.res_skip_layers." + std::to_string(li), rs, H, 1, true, &f.wn.res[li]));
      GSV_RC(dalloc(h, (void**)&f.wn.in_bias_eff[li], (size_t)2 * H * 4));
    }
    GSV_RC(make_conv(h, p + ".enc.cond_layer", 2 * H * 4, GIN, 1, true, &f.wn.cond));
  }
  // generator
  const int UIC = c.upsample_initial_channel;
  GSV_RC(make_conv(h, "dec.conv_pre", UIC, IC, 7, true, &h->conv_pre));
  GSV_RC(make_conv(h, "dec.cond", UIC, GIN, 1, true, &h->cond));
  GSV_RC(dalloc(h, (void**)&h->conv_pre_bias_eff, (size_t)UIC * 4));
  h->ups.resize(c.n_ups);
  int ch = UIC;
  for (int i = 0; i < c.n_ups; ++i) {
    const int cin = UIC >> i, cout = UIC >> (i + 1);
    GSV_REQUIRE(cout % 8 == 0, "vits: generator channel count %d must be a multiple of 8", cout);
    GSV_RC(make_ups(h, "dec.ups." + std::to_string(i), cin, cout, c.up_kernels[i], c.up_rates[i], &h->ups[i]));
    ch = cout;
    for (int j = 0; j < c.n_resblocks; ++j) {
      const std::string r = "dec.resblocks." + std::to_string(i * c.n_resblocks + j);
      for (int k = 0; k < 3; ++k) {
        Conv c1, c2;
        GSV_RC(make_conv(h, r + ".convs1." + std::to_string(k), ch, ch, c.rb_kernels[j], true, &c1));
        GSV_RC(make_conv(h, r + ".convs2." + std::to_string(k), ch, ch, c.rb_kernels[j], true, &c2));
        h->rb1.push_back(c1);
        h->rb2.push_back(c2);
      }
    }
  }
  GSV_RC(make_conv(h, "dec.conv_post", 1, ch, 7, false, &h->conv_post));
  } else {
    // v3 / v4 (SynthesizerTrnV3, module/models.py:1203-1206): bridge + wns1 = Encoder(512, 512, 512, 5, 1, 8, gin)
    const int W = 512, NL = 8;
    GSV_REQUIRE(IC == H, "vits: v3/v4 need inter_channels == hidden_channels (bridge = Conv1d(inter, 512) on the hidden sequence)");
    GSV_RC(make_conv(h, "bridge.0", W, IC, 1, true, &h->bridge));
    GSV_RC(make_conv(h, "wns1.pre", W, W, 1, true, &h->w1_pre));
    GSV_RC(make_conv(h, "wns1.proj", W, W, 1, true, &h->w1_proj));
    GSV_RC(make_conv(h, "wns1.enc.cond_layer", 2 * W * NL, GIN, 1, true, &h->w1_cond));
    h->w1_in.resize(NL); h->w1_res.resize(NL); h->w1_in_bias_eff.resize(NL);
    for (int li = 0; li < NL; ++li) {
      GSV_RC(make_conv(h, "wns1.enc.in_layers." + std::to_string(li), 2 * W, W, 5, true, &h->w1_in[li]));
      GSV_RC(make_conv(h, "wns1.enc.res_skip_layers." + std::to_string(li), li < NL - 1 ? 2 * W : W, W, 1, true, &h->w1_res[li]));
      GSV_RC(dalloc(h, (void**)&h->w1_in_bias_eff[li], (size_t)2 * W * 4));
    }
  }
  // ref_enc
  const int RH = 128;
  GSV_RC(make_conv(h, "ref_enc.spectral.0.fc", RH, c.ref_bins, 1, true, &h->r_sp0));
  GSV_RC(make_conv(h, "ref_enc.spectral.3.fc", RH, RH, 1, true, &h->r_sp3));
  GSV_RC(make_conv(h, "ref_enc.temporal.0.conv1.conv", 2 * RH, RH, 5, true, &h->r_t0));
  GSV_RC(make_conv(h, "ref_enc.temporal.1.conv1.conv", 2 * RH, RH, 5, true, &h->r_t1));
  GSV_RC(make_stacked(h, {"ref_enc.slf_attn.w_qs", "ref_enc.slf_attn.w_ks", "ref_enc.slf_attn.w_vs"}, RH, RH, &h->r_qkv));
  GSV_RC(make_conv(h, "ref_enc.slf_attn.fc", RH, RH, 1, true, &h->r_fc));
  GSV_RC(make_conv(h, "ref_enc.fc.fc", GIN, RH, 1, true, &h->r_out));
  GSV_RC(dalloc(h, (void**)&h->ge, (size_t)GIN * 4));
  GSV_RC(dalloc(h, &h->ge_t, (size_t)GIN * esz(h)));
  GSV_RC(dalloc(h, (void**)&h->mo_bias_eff, (size_t)MH * 4));
  if (c.v2pro) {
    GSV_RC(make_conv(h, "sv_emb", GIN, 20480, 1, true, &h->sv_emb));
    GSV_RC(make_conv(h, "ge_to512", MH, GIN, 1, true, &h->ge_to512));
    GSV_RC(make_vec(h, "prelu.weight", GIN, &h->prelu_w));
    GSV_RC(dalloc(h, (void**)&h->ge_ref, (size_t)GIN * 4));
    GSV_RC(dalloc(h, (void**)&h->sv_proj, (size_t)GIN * 4));
    GSV_RC(dalloc(h, (void**)&h->ge512, (size_t)MH * 4));
    GSV_RC(dalloc(h, &h->sv_t, (size_t)20480 * esz(h)));
  }
  for (auto& e : h->ev) GSV_HIP(hipEventCreate(&e));
  h->staged.clear();
  h->finalized = true;
  return GSV_OK;
}

static int set_refer_impl(gsv_vits_t* h, const float* const* specs, const int* frames, int bins, const float* const* sv_embs, int n_refs,
                          gsv_stream_t stream);

int gsv_vits_set_refer(gsv_vits_t* h, const float* const* specs, const int* frames, int bins, int n_refs, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized, "vits_set_refer: handle not finalized");
  GSV_REQUIRE(!h->cfg.v2pro, "vits_set_refer: a v2Pro model needs gsv_vits_set_refer_sv (one sv embedding per reference)");
  return set_refer_impl(h, specs, frames, bins, nullptr, n_refs, stream);
}

int gsv_vits_set_refer_sv(gsv_vits_t* h, const float* const* specs, const int* frames, int bins, const float* const* sv_embs,
                          int n_refs, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized, "vits_set_refer_sv: handle not finalized");
  GSV_REQUIRE(h->cfg.v2pro && sv_embs, "vits_set_refer_sv: not a v2Pro model, or no sv embeddings");
  for (int r = 0; r < n_refs; ++r) GSV_REQUIRE(sv_embs[r], "vits_set_refer_sv: missing sv embedding %d", r);
  return set_refer_impl(h, specs, frames, bins, sv_embs, n_refs, stream);
}

static int set_refer_impl(gsv_vits_t* h, const float* const* specs, const int* frames, int bins, const float* const* sv_embs, int n_refs,
                          gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized, "vits_set_refer: handle not finalized");
  GSV_REQUIRE(specs && frames && n_refs >= 1, "vits_set_refer: no reference spectrogram");
  GSV_REQUIRE(bins >= h->cfg.ref_bins, "vits_set_refer: spectrogram has %d bins, need >= %d", bins, h->cfg.ref_bins);
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const size_t es = esz(h);
  const int RB = c.ref_bins, RH = 128, GIN = c.gin_channels;
  for (int r = 0; r < n_refs; ++r) {
    const int Tr = frames[r];
    GSV_REQUIRE(Tr >= 1 && specs[r], "vits_set_refer: empty reference %d", r);
    void *x0, *a, *b, *y2, *qkv, *ao;
    GSV_RC(need(h, "ref_x0", (size_t)Tr * RB * es, &x0));
    GSV_RC(need(h, "ref_a", (size_t)Tr * RH * es, &a));
    GSV_RC(need(h, "ref_b", (size_t)Tr * RH * es, &b));
    GSV_RC(need(h, "ref_y2", (size_t)Tr * 2 * RH * es, &y2));
    GSV_RC(need(h, "ref_qkv", (size_t)Tr * 3 * RH * es, &qkv));
    GSV_RC(need(h, "ref_ao", (size_t)Tr * GIN * es, &ao));
    GSV_DISPATCH(h,
      hipLaunchKernelGGL(cf_to_cl_kernel<_Float16>, dim3(cdiv(Tr, 32), cdiv(RB, 32)), dim3(256), 0, s, specs[r], Tr, RB, (_Float16*)x0),
      hipLaunchKernelGGL(cf_to_cl_kernel<float>, dim3(cdiv(Tr, 32), cdiv(RB, 32)), dim3(256), 0, s, specs[r], Tr, RB, (float*)x0));
    ConvOpt om; om.post_act = ACT_MISH;
    GSV_RC(conv(h, s, h->r_sp0, x0, RB, Tr, a, Tr, om));
    GSV_RC(conv(h, s, h->r_sp3, a, RH, Tr, b, Tr, om));
    ConvOpt o;
    for (int t = 0; t < 2; ++t) {
      GSV_RC(conv(h, s, t == 0 ? h->r_t0 : h->r_t1, b, RH, Tr, y2, Tr, o));
      GSV_DISPATCH(h,
        hipLaunchKernelGGL(glu_res_kernel<_Float16>, dim3(nblk((long long)Tr * RH)), dim3(256), 0, s, (const _Float16*)y2, (long long)Tr * RH, RH, (_Float16*)b),
        hipLaunchKernelGGL(glu_res_kernel<float>, dim3(nblk((long long)Tr * RH)), dim3(256), 0, s, (const float*)y2, (long long)Tr * RH, RH, (float*)b));
    }
    GSV_RC(conv(h, s, h->r_qkv, b, RH, Tr, qkv, Tr, o));
    // 2 heads x 64, temperature sqrt(d_model) (modules.py:610)
    GSV_RC(attention(h, s, qkv, 3 * RH, 0, qkv, 3 * RH, RH, 2 * RH, Tr, Tr, 2, RH / 2, 1.f / sqrtf((float)RH), nullptr, nullptr, a, RH));
    ConvOpt orr; orr.res = b; orr.ldr = RH;
    GSV_RC(conv(h, s, h->r_fc, a, RH, Tr, y2, Tr, orr));          // fc(attn) + residual -> y2 [Tr][RH]
    GSV_RC(conv(h, s, h->r_out, y2, RH, Tr, ao, Tr, o));            // [Tr][GIN]
    if (!sv_embs) {
      GSV_DISPATCH(h,
        hipLaunchKernelGGL(mean_time_kernel<_Float16>, dim3(cdiv(GIN, 64)), dim3(64), 0, s, (const _Float16*)ao, Tr, GIN, 1.f / n_refs, r > 0, h->ge),
        hipLaunchKernelGGL(mean_time_kernel<float>, dim3(cdiv(GIN, 64)), dim3(64), 0, s, (const float*)ao, Tr, GIN, 1.f / n_refs, r > 0, h->ge));
    } else {
      GSV_DISPATCH(h,
        hipLaunchKernelGGL(mean_time_kernel<_Float16>, dim3(cdiv(GIN, 64)), dim3(64), 0, s, (const _Float16*)ao, Tr, GIN, 1.f, 0, h->ge_ref),
        hipLaunchKernelGGL(mean_time_kernel<float>, dim3(cdiv(GIN, 64)), dim3(64), 0, s, (const float*)ao, Tr, GIN, 1.f, 0, h->ge_ref));
      GSV_RC(launch_convert(sv_embs[r], h->sv_t, h->dtype, 20480, s));
      ConvOpt osv; osv.out_f32 = 1;
      GSV_RC(conv(h, s, h->sv_emb, h->sv_t, 20480, 1, h->sv_proj, 1, osv));
      hipLaunchKernelGGL(sv_prelu_acc_kernel, dim3(cdiv(GIN, 256)), dim3(256), 0, s, (const float*)h->ge_ref, (const float*)h->sv_proj,
                         (const float*)h->prelu_w, 1.f / n_refs, r > 0, GIN, h->ge);
    }
  }
  GSV_RC(launch_convert(h->ge, h->ge_t, h->dtype, GIN, s));
  // fold conditioning into biases: conv_pre + cond(ge); MRTE conv_o bias + ge; WN in_layers + cond_layer(ge)
  float* tmp;
  GSV_RC(need(h, "cond_tmp", (size_t)2048 * 4 * 4, (void**)&tmp));
  {
    ConvOpt o; o.out_f32 = 1;
    if (c.v2pro) {     // the MRTE adds ge_to512(ge) (models.py:997)
      GSV_RC(conv(h, s, h->ge_to512, h->ge_t, GIN, 1, h->ge512, 1, o));
      hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(512, 256)), dim3(256), 0, s, h->ge512, h->mo.b, h->mo_bias_eff, 512);
    } else {
      hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(GIN, 256)), dim3(256), 0, s, h->ge, h->mo.b, h->mo_bias_eff, GIN);
    }
    if (c.flavor != 0) {
      GSV_RC(conv(h, s, h->w1_cond, h->ge_t, GIN, 1, tmp, 1, o));
      const int n = 2 * 512;
      for (size_t li = 0; li < h->w1_in.size(); ++li)
        hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, tmp + li * n, h->w1_in[li].b, h->w1_in_bias_eff[li], n);
    } else {
    GSV_RC(conv(h, s, h->cond, h->ge_t, GIN, 1, tmp, 1, o));
    hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(h->conv_pre.cout, 256)), dim3(256), 0, s, tmp, h->conv_pre.b, h->conv_pre_bias_eff, h->conv_pre.cout);
    for (int fi = 0; fi < 4; ++fi) {
      WNW& w = h->flows[fi].wn;
      GSV_RC(conv(h, s, w.cond, h->ge_t, GIN, 1, tmp, 1, o));
      const int n = 2 * c.hidden_channels;
      for (int li = 0; li < 4; ++li)
        hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, tmp + li * n, w.in[li].b, w.in_bias_eff[li], n);
    }
    }
  }
  GSV_HIP(hipGetLastError());
  h->has_ref = true;
  return GSV_OK;
}

int gsv_vits_decode(gsv_vits_t* h, const int32_t* codes, int T, const int32_t* phones, int L, const float* noise,
                    float noise_scale, double speed, uint64_t seed, float* wav, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized, "vits_decode: handle not finalized");
  GSV_REQUIRE(h->has_ref, "vits_decode: call gsv_vits_set_refer first");
  GSV_REQUIRE(codes && phones && wav && T >= 1 && L >= 1, "vits_decode: empty input (T=%d, L=%d)", T, L);
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const size_t es = esz(h);
  const int H = c.hidden_channels, IC = c.inter_channels;
  GSV_REQUIRE(speed > 0.0, "vits_decode: speed must be positive");
  GSV_REQUIRE(c.flavor == 0, "vits_decode: this handle is a v3/v4 model (use gsv_vits_decode_encp + gsv_cfm_inference + a vocoder)");
  GSV_HIP(hipEventRecord(h->ev[0], s));
  void* y = nullptr;
  int F = 0;
  GSV_RC(run_enc_p(h, s, codes, T, phones, L, speed, &y, &F));
  ConvOpt o;
  float* stats;
  GSV_RC(need(h, "stats", (size_t)F * 2 * IC * 4, (void**)&stats));
  { ConvOpt of; of.out_f32 = 1; GSV_RC(conv(h, s, h->proj, y, H, F, stats, F, of)); }
  // ---- z_p, H11: flow reverse
  void *z, *zf, *hb, *xin, *acts, *wout;
  GSV_RC(need(h, "z", (size_t)F * IC * es, &z));
  GSV_RC(need(h, "zf", (size_t)F * IC * es, &zf));
  GSV_RC(need(h, "wn_h", (size_t)F * H * es, &hb));
  GSV_RC(need(h, "wn_xin", (size_t)F * 2 * H * es, &xin));
  GSV_RC(need(h, "wn_acts", (size_t)F * H * es, &acts));
  GSV_RC(need(h, "wn_out", (size_t)F * H * es, &wout));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(zp_kernel<_Float16>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, stats, F, IC, noise, noise_scale, (unsigned long long)seed, (_Float16*)z),
    hipLaunchKernelGGL(zp_kernel<float>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, stats, F, IC, noise, noise_scale, (unsigned long long)seed, (float*)z));
  const int half = IC / 2;
  for (int fi = 3; fi >= 0; --fi) {
    FlowW& f = h->flows[fi];
    GSV_DISPATCH(h,
      hipLaunchKernelGGL(flip_channels_kernel<_Float16>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, (const _Float16*)z, (long long)F * IC, IC, (_Float16*)zf),
      hipLaunchKernelGGL(flip_channels_kernel<float>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, (const float*)z, (long long)F * IC, IC, (float*)zf));
    std::swap(z, zf);
    GSV_RC(conv(h, s, f.pre, z, IC, F, hb, F, o));   // x0 = channels [0, half)
    for (int li = 0; li < 4; ++li) {
      ConvOpt oi; oi.bias_override = f.wn.in_bias_eff[li];
      GSV_RC(conv(h, s, f.wn.in[li], hb, H, F, xin, F, oi));
      GSV_DISPATCH(h,
        hipLaunchKernelGGL(gate_kernel<_Float16>, dim3(nblk((long long)F * H)), dim3(256), 0, s, (const _Float16*)xin, (long long)F * H, H, (_Float16*)acts),
        hipLaunchKernelGGL(gate_kernel<float>, dim3(nblk((long long)F * H)), dim3(256), 0, s, (const float*)xin, (long long)F * H, H, (float*)acts));
      if (li < 3) {
        ConvOpt ores; ores.cout = H; ores.w_row0 = 0; ores.accumulate = 1;        // h += rs[:H]
        GSV_RC(conv(h, s, f.wn.res[li], acts, H, F, hb, F, ores));
        ConvOpt osk; osk.cout = H; osk.w_row0 = H; osk.accumulate = li > 0;       // out (+)= rs[H:]
        GSV_RC(conv(h, s, f.wn.res[li], acts, H, F, wout, F, osk));
      } else {
        ConvOpt osk; osk.accumulate = 1;
        GSV_RC(conv(h, s, f.wn.res[li], acts, H, F, wout, F, osk));
      }
    }
    ConvOpt op; op.scale = -1.f; op.accumulate = 1; op.ldy = IC; op.y_col0 = half;  // x1 -= post(h)
    GSV_RC(conv(h, s, f.post, wout, H, F, z, F, op));
  }
  h->lastF = F;
  // keep a stable pointer to the final z for the debug hook
  {
    void* zkeep;
    GSV_RC(need(h, "z_keep", (size_t)F * IC * es, &zkeep));
    GSV_HIP(hipMemcpyAsync(zkeep, z, (size_t)F * IC * es, hipMemcpyDeviceToDevice, s));
  }
  GSV_HIP(hipEventRecord(h->ev[1], s));
  // ---- H12: generator
  size_t maxel = (size_t)F * c.upsample_initial_channel;
  {
    long long Tn = F; int ch = c.upsample_initial_channel;
    for (int i = 0; i < c.n_ups; ++i) { Tn *= c.up_rates[i]; ch >>= 1; maxel = std::max(maxel, (size_t)Tn * ch); }
  }
  void* gb[5];
  const char* gnames[5] = {"g0", "g1", "g2", "g3", "g4"};
  for (int i = 0; i < 5; ++i) GSV_RC(need(h, gnames[i], maxel * es, &gb[i]));
  void* cur = gb[3];
  { ConvOpt op; op.bias_override = h->conv_pre_bias_eff; GSV_RC(conv(h, s, h->conv_pre, z, IC, F, cur, F, op)); }
  int Tn = F, ch = c.upsample_initial_channel;
  for (int i = 0; i < c.n_ups; ++i) {
    const int Tout = Tn * c.up_rates[i];
    ch >>= 1;
    void* xup = gb[0]; void* xt = gb[1]; void* R = gb[2]; void* xs = (cur == gb[3]) ? gb[4] : gb[3];
    { ConvOpt ou; ou.pre_act = ACT_LRELU; ou.pre_slope = 0.1f; GSV_RC(conv(h, s, h->ups[i], cur, ch * 2, Tn, xup, Tout, ou)); }
    for (int j = 0; j < c.n_resblocks; ++j) {
      const void* xr = xup;
      for (int k = 0; k < 3; ++k) {
        const Conv& c1 = h->rb1[(i * c.n_resblocks + j) * 3 + k];
        const Conv& c2 = h->rb2[(i * c.n_resblocks + j) * 3 + k];
        if (c1.b && c2.b && c1.taps == c2.taps && conv_pair_eligible(h->dtype, ch, c1.taps, c.rb_dilations[j][k], Tout)) {
          // narrow stages: the pair in one kernel, the intermediate tensor never leaves the CU (conv_pair.hip)
          ConvPairArgs pa;
          pa.x = (const _Float16*)xr; pa.w1 = (const _Float16*)c1.w; pa.b1 = c1.b; pa.w2 = (const _Float16*)c2.w; pa.b2 = c2.b;
          pa.T = Tout; pa.C = ch; pa.taps = c1.taps; pa.dil = c.rb_dilations[j][k]; pa.ldx = ch; pa.ldy = ch;
          if (k < 2) { pa.y = (_Float16*)R; }
          else { pa.y = (_Float16*)xs; pa.scale = 1.f / (float)c.n_resblocks; pa.accumulate = j > 0; }
          // the pair reads x as window AND residual: it must not be overwritten in place
          if ((const void*)pa.y == xr) { pa.y = (_Float16*)xt; }
          GSV_RC(launch_conv_pair(pa, s));
          if (k < 2) { if (pa.y == (_Float16*)xt) { std::swap(xt, R); } xr = R; }
          continue;
        }
        ConvOpt o1; o1.pre_act = ACT_LRELU; o1.pre_slope = 0.1f; o1.dil = c.rb_dilations[j][k];
        GSV_RC(conv(h, s, c1, xr, ch, Tout, xt, Tout, o1));
        ConvOpt o2; o2.pre_act = ACT_LRELU; o2.pre_slope = 0.1f; o2.res = xr; o2.ldr = ch;
        if (k < 2) {
          GSV_RC(conv(h, s, c2, xt, ch, Tout, R, Tout, o2));
          xr = R;
        } else {
          o2.scale = 1.f / (float)c.n_resblocks; o2.accumulate = j > 0;
          GSV_RC(conv(h, s, c2, xt, ch, Tout, xs, Tout, o2));
        }
      }
    }
    cur = xs; Tn = Tout;
  }
  { ConvOpt op; op.pre_act = ACT_LRELU; op.pre_slope = 0.01f; op.post_act = ACT_TANH; op.out_f32 = 1;
    GSV_RC(conv(h, s, h->conv_post, cur, ch, Tn, wav, Tn, op)); }
  GSV_HIP(hipEventRecord(h->ev[2], s));
  return GSV_OK;
}

int gsv_vits_encp_frames(gsv_vits_t* h, int T, double speed) {
  if (!h || T < 1 || !(speed > 0.0)) return -1;
  const int F0 = 2 * T;
  const int Fs = (speed == 1.0) ? F0 : (int)((double)F0 / speed) + 1;
  const double sf = h->cfg.flavor == 1 ? 1.875 : 2.0;
  return (int)floor((double)Fs * sf);
}

int gsv_vits_decode_encp(gsv_vits_t* h, const int32_t* codes, int T, const int32_t* phones, int L, double speed, float* fea,
                         gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized, "vits_decode_encp: handle not finalized");
  GSV_REQUIRE(h->cfg.flavor != 0, "vits_decode_encp: this handle is a v1/v2 model (use gsv_vits_decode)");
  GSV_REQUIRE(h->has_ref, "vits_decode_encp: call gsv_vits_set_refer first");
  GSV_REQUIRE(codes && phones && fea && T >= 1 && L >= 1, "vits_decode_encp: empty input (T=%d, L=%d)", T, L);
  GSV_REQUIRE(speed > 0.0, "vits_decode_encp: speed must be positive");
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const size_t es = esz(h);
  const int H = c.hidden_channels, W = 512, NL = (int)h->w1_in.size();
  void* y = nullptr;
  int Fs = 0;
  GSV_RC(run_enc_p(h, s, codes, T, phones, L, speed, &y, &Fs));
  const double sf = c.flavor == 1 ? 1.875 : 2.0;
  const int F = (int)floor((double)Fs * sf);
  // wns1's mask length (models.py:1252-1258): frames >= Lm are zeroed at every masked point of Encoder / WN
  const double per = c.flavor == 1 ? 3.875 : 4.0;
  const int sizee = (speed == 1.0) ? (int)((double)T * per) : (int)((double)T * per / speed) + 1;
  const int Lm = std::min(sizee, F);
  void *br, *up, *hb, *xin, *acts, *wout, *st;
  GSV_RC(need(h, "e_br", (size_t)Fs * W * es, &br));
  GSV_RC(need(h, "e_up", (size_t)F * W * es, &up));
  GSV_RC(need(h, "e_h", (size_t)F * W * es, &hb));
  GSV_RC(need(h, "e_xin", (size_t)F * 2 * W * es, &xin));
  GSV_RC(need(h, "e_acts", (size_t)F * W * es, &acts));
  GSV_RC(need(h, "e_out", (size_t)F * W * es, &wout));
  GSV_RC(need(h, "e_st", (size_t)F * W * es, &st));
  auto mask_tail = [&](void* p) -> int {
    if (Lm < F) GSV_HIP(hipMemsetAsync((char*)p + (size_t)Lm * W * es, 0, (size_t)(F - Lm) * W * es, s));
    return GSV_OK;
  };
  { ConvOpt ob; ob.post_act = ACT_LRELU01; GSV_RC(conv(h, s, h->bridge, y, H, Fs, br, Fs, ob)); }   // bridge: 1x1 + LeakyReLU(0.01)
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(interp_nearest_kernel<_Float16>, dim3(F), dim3(128), 0, s, (const _Float16*)br, Fs, F, W, (float)(1.0 / sf), (_Float16*)up),
    hipLaunchKernelGGL(interp_nearest_kernel<float>, dim3(F), dim3(128), 0, s, (const float*)br, Fs, F, W, (float)(1.0 / sf), (float*)up));
  ConvOpt o;
  GSV_RC(conv(h, s, h->w1_pre, up, W, F, hb, F, o));
  GSV_RC(mask_tail(hb));
  for (int li = 0; li < NL; ++li) {                        // modules.WN.forward (modules.py:  in -> gate -> res/skip)
    ConvOpt oi; oi.bias_override = h->w1_in_bias_eff[li];
    GSV_RC(conv(h, s, h->w1_in[li], hb, W, F, xin, F, oi));
    GSV_DISPATCH(h,
      hipLaunchKernelGGL(gate_kernel<_Float16>, dim3(nblk((long long)F * W)), dim3(256), 0, s, (const _Float16*)xin, (long long)F * W, W, (_Float16*)acts),
      hipLaunchKernelGGL(gate_kernel<float>, dim3(nblk((long long)F * W)), dim3(256), 0, s, (const float*)xin, (long long)F * W, W, (float*)acts));
    if (li < NL - 1) {
      ConvOpt ores; ores.cout = W; ores.w_row0 = 0; ores.accumulate = 1;
      GSV_RC(conv(h, s, h->w1_res[li], acts, W, F, hb, F, ores));
      GSV_RC(mask_tail(hb));
      ConvOpt osk; osk.cout = W; osk.w_row0 = W; osk.accumulate = li > 0;
      GSV_RC(conv(h, s, h->w1_res[li], acts, W, F, wout, F, osk));
    } else {
      ConvOpt osk; osk.accumulate = NL > 1;
      GSV_RC(conv(h, s, h->w1_res[li], acts, W, F, wout, F, osk));
    }
  }
  GSV_RC(mask_tail(wout));
  GSV_RC(conv(h, s, h->w1_proj, wout, W, F, st, F, o));
  GSV_RC(mask_tail(st));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(cl_to_cf_kernel<_Float16>, dim3(nblk((long long)F * W)), dim3(256), 0, s, (const _Float16*)st, F, W, 0, W, fea),
    hipLaunchKernelGGL(cl_to_cf_kernel<float>, dim3(nblk((long long)F * W)), dim3(256), 0, s, (const float*)st, F, W, 0, W, fea));
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_vits_last_timing(gsv_vits_t* h, float* total_ms, float* generator_ms) {
  GSV_REQUIRE(h && h->finalized && h->lastF > 0, "vits_last_timing: no decode yet");
  GSV_HIP(hipEventSynchronize(h->ev[2]));
  float a = 0.f, b = 0.f;
  GSV_HIP(hipEventElapsedTime(&a, h->ev[0], h->ev[2]));
  GSV_HIP(hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
  if (total_ms) *total_ms = a;
  if (generator_ms) *generator_ms = b;
  return GSV_OK;
}

int gsv_vits_extract_latent(gsv_vits_t* h, const float* ssl, int T50, int32_t* codes, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized && ssl && codes, "vits_extract_latent: bad argument");
  GSV_REQUIRE(T50 >= 2, "vits_extract_latent: need at least 2 ssl frames (got %d)", T50);
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const int SSL = c.ssl_dim, T25 = (T50 - 2) / 2 + 1;
  void* x; float *p, *dots;
  GSV_RC(need(h, "xl_x", (size_t)T50 * SSL * esz(h), &x));
  GSV_RC(need(h, "xl_p", (size_t)T25 * SSL * 4, (void**)&p));
  GSV_RC(need(h, "xl_d", (size_t)T25 * c.n_bins * 4, (void**)&dots));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(cf_to_cl_kernel<_Float16>, dim3(cdiv(T50, 32), cdiv(SSL, 32)), dim3(256), 0, s, ssl, T50, SSL, (_Float16*)x),
    hipLaunchKernelGGL(cf_to_cl_kernel<float>, dim3(cdiv(T50, 32), cdiv(SSL, 32)), dim3(256), 0, s, ssl, T50, SSL, (float*)x));
  ConvOpt o; o.stride = 2; o.pad = 0; o.out_f32 = 1;
  GSV_RC(conv(h, s, h->top_ssl_proj, x, SSL, T50, p, T25, o));
  // x . E^T in the engine dtype operands (fp32 engine: exact-f32 MFMA)
  void* pt;
  GSV_RC(need(h, "xl_pt", (size_t)T25 * SSL * esz(h), &pt));
  GSV_RC(launch_convert(p, pt, h->dtype, (long long)T25 * SSL, s));
  ConvArgs a;
  a.x = pt; a.w = h->codebook_t; a.y = dots; a.out_f32 = 1;
  a.T_in = T25; a.T_out = T25; a.T_virt = T25; a.Cin = SSL; a.Cout = c.n_bins; a.ldx = SSL; a.ldw = SSL; a.ldy = c.n_bins;
  GSV_RC(launch_conv_gemm(h->dtype, a, s));
  hipLaunchKernelGGL(argmax_code_kernel, dim3(T25), dim3(64), 0, s, dots, p, h->code_ee, SSL, c.n_bins, codes);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_vits_debug_tensor(gsv_vits_t* h, const char* name, float* out, int64_t cap, int64_t* numel, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized && name && out && numel, "vits_debug_tensor: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const std::string n(name);
  const int F = h->lastF, IC = c.inter_channels;
  if (n == "ge") {
    GSV_REQUIRE(cap >= c.gin_channels, "vits_debug_tensor: buffer too small");
    GSV_HIP(hipMemcpyAsync(out, h->ge, (size_t)c.gin_channels * 4, hipMemcpyDeviceToDevice, s));
    *numel = c.gin_channels;
    return GSV_OK;
  }
  GSV_REQUIRE(F > 0, "vits_debug_tensor: no decode yet");
  GSV_REQUIRE(cap >= (int64_t)F * IC, "vits_debug_tensor: buffer too small");
  *numel = (int64_t)F * IC;
  if (n == "m_p" || n == "logs_p") {
    const float* st = (const float*)h->bufs["stats"].p;
    hipLaunchKernelGGL(cl_to_cf_kernel<float>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, st, F, 2 * IC, n == "m_p" ? 0 : IC, IC, out);
  } else if (n == "z") {
    const void* z = h->bufs["z_keep"].p;
    GSV_DISPATCH(h,
      hipLaunchKernelGGL(cl_to_cf_kernel<_Float16>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, (const _Float16*)z, F, IC, 0, IC, out),
      hipLaunchKernelGGL(cl_to_cf_kernel<float>, dim3(nblk((long long)F * IC)), dim3(256), 0, s, (const float*)z, F, IC, 0, IC, out));
  } else {
    set_error("vits_debug_tensor: unknown tensor '%s' (ge, m_p, logs_p, z)", name);
    return GSV_ERR_ARG;
  }
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

}  // extern "C"

// =======================================================================================
// Vocoders of the v3/v4 path: v4 = the HiFi-GAN `Generator` used as a mel vocoder (H16, reference
// TTS_infer_pack/TTS.py:631-648, module/models.py:407-471), v3 = BigVGAN-v2 (H15, reference
// BigVGAN/bigvgan.py:226-355 with AMPBlock1 :31-131 and anti-aliased SnakeBeta).  Same conv kernels
// and channels-last layout as the v2 generator above; mel channels are zero-padded to a multiple of 8.
// =======================================================================================
struct VocAct { float *alpha = nullptr, *beta = nullptr; };

struct gsv_vocoder {
  gsv_vits ctx;               // reused as the allocation / staging / workspace context of the helpers above
  gsv_vocoder_config cfg;
  int cin_pad = 0;
  Conv conv_pre, conv_post;
  std::vector<Conv> ups, rb1, rb2;
  std::vector<VocAct> acts;   // BigVGAN: [stage][block][6] + final
  float *up12 = nullptr, *dn12 = nullptr;
  bool finalized = false;
};

namespace gsveng {

// Conv1d weight [cout][cin][k] with the input channels zero-padded to cin_pad
int make_conv_padded(gsv_vits* h, const std::string& name, int cout, int cin, int cin_pad, int k, bool bias, Conv* c) {
  std::vector<float> w, b;
  if (!fetch(h, name + ".weight", (size_t)cout * cin * k, cout, w)) return GSV_ERR_ARG;
  std::vector<float> p((size_t)cout * k * cin_pad, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i)
      for (int j = 0; j < k; ++j) p[((size_t)o * k + j) * cin_pad + i] = w[((size_t)o * cin + i) * k + j];
  GSV_RC(up_t(h, p, &c->w));
  if (bias) {
    if (!fetch(h, name + ".bias", cout, cout, b)) return GSV_ERR_ARG;
    GSV_RC(up_f32(h, b.data(), b.size(), &c->b));
  }
  c->cin = cin_pad; c->cout = cout; c->taps = k;
  return GSV_OK;
}

// Kaiser-windowed sinc low-pass of BigVGAN's Activation1d (filter.py:30-60), cutoff 0.25, half-width 0.3, 12 taps
void kaiser_sinc12(float* out) {
  const int K = 12, half = 6;
  const double cutoff = 0.25, hw = 0.3;
  const double A = 2.285 * (half - 1) * M_PI * 4 * hw + 7.95;
  const double beta = A > 50.0 ? 0.1102 * (A - 8.7) : (A >= 21.0 ? 0.5842 * pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0) : 0.0);
  auto i0 = [](double x) { double s = 1.0, t = 1.0; for (int k = 1; k < 60; ++k) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; } return s; };
  double f[12], sum = 0.0;
  for (int n = 0; n < K; ++n) {
    const double r = 2.0 * n / (K - 1) - 1.0;
    const double win = i0(beta * sqrt(1.0 - r * r)) / i0(beta);
    const double t = (n - half) + 0.5;
    const double xx = 2 * cutoff * t;
    const double sinc = xx == 0.0 ? 1.0 : sin(M_PI * xx) / (M_PI * xx);
    f[n] = 2 * cutoff * win * sinc;
    sum += f[n];
  }
  for (int n = 0; n < K; ++n) out[n] = (float)(f[n] / sum);
}

int voc_act(gsv_vocoder* v, hipStream_t s, const VocAct& a, const void* x, void* y, int Tn, int C) {
  gsv_vits* h = &v->ctx;
  dim3 grid(cdiv(Tn, 64), cdiv(C, 64));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(aa_act_cl_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)x, (_Float16*)y, Tn, C, C, a.alpha, a.beta,
                       v->cfg.snake_logscale, v->up12, v->dn12),
    hipLaunchKernelGGL(aa_act_cl_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)y, Tn, C, C, a.alpha, a.beta,
                       v->cfg.snake_logscale, v->up12, v->dn12));
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

}  // namespace gsveng

extern "C" {

int gsv_vocoder_create(const gsv_vocoder_config* cfg, int dtype, gsv_vocoder_t** out) {
  GSV_REQUIRE(cfg && out, "vocoder_create: null argument");
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "vocoder_create: bad dtype");
  GSV_REQUIRE(cfg->n_ups >= 1 && cfg->n_ups <= 8 && cfg->n_resblocks >= 1 && cfg->n_resblocks <= 4, "vocoder_create: bad shape");
  GSV_REQUIRE(cfg->kind == 0 || cfg->kind == 1, "vocoder_create: kind must be 0 (HiFi-GAN) or 1 (BigVGAN)");
  GSV_REQUIRE((cfg->upsample_initial_channel >> cfg->n_ups) % 8 == 0, "vocoder_create: final channel count must be a multiple of 8");
  int n = 0;
  GSV_HIP(hipGetDeviceCount(&n));
  gsv_vocoder* v = new gsv_vocoder();
  v->cfg = *cfg;
  v->ctx.dtype = dtype;
  v->cin_pad = (cfg->in_channels + 7) / 8 * 8;
  *out = v;
  return GSV_OK;
}

void gsv_vocoder_destroy(gsv_vocoder_t* v) {
  if (!v) return;
  for (void* p : v->ctx.allocs) (void)hipFree(p);
  for (auto& b : v->ctx.bufs) if (b.second.p) (void)hipFree(b.second.p);
  delete v;
}

int gsv_vocoder_load_tensor(gsv_vocoder_t* v, const char* name, const float* data, int64_t numel) {
  GSV_REQUIRE(v && name && data && numel > 0, "vocoder_load_tensor: bad argument");
  GSV_REQUIRE(!v->finalized, "vocoder_load_tensor: handle already finalized");
  v->ctx.staged[name].assign(data, data + numel);
  return GSV_OK;
}

int gsv_vocoder_finalize(gsv_vocoder_t* v) {
  GSV_REQUIRE(v && !v->finalized, "vocoder_finalize: bad handle");
  gsv_vits* h = &v->ctx;
  const auto& c = v->cfg;
  const int UIC = c.upsample_initial_channel;
  const bool big = c.kind == 1;
  GSV_RC(make_conv_padded(h, "conv_pre", UIC, c.in_channels, v->cin_pad, 7, true, &v->conv_pre));
  v->ups.resize(c.n_ups);
  int ch = UIC;
  auto load_act = [&](const std::string& prefix, int C, VocAct* a) -> int {
    GSV_RC(make_vec(h, prefix + ".alpha", C, &a->alpha));
    if (h->staged.count(prefix + ".beta")) { GSV_RC(make_vec(h, prefix + ".beta", C, &a->beta)); }
    else a->beta = a->alpha;   // Snake: one parameter for both (activation1d.py:58-61)
    return GSV_OK;
  };
  for (int i = 0; i < c.n_ups; ++i) {
    const int cin = UIC >> i, cout = UIC >> (i + 1);
    const std::string un = big ? "ups." + std::to_string(i) + ".0" : "ups." + std::to_string(i);
    GSV_RC(make_ups(h, un, cin, cout, c.up_kernels[i], c.up_rates[i], &v->ups[i]));
    ch = cout;
    for (int j = 0; j < c.n_resblocks; ++j) {
      const std::string r = "resblocks." + std::to_string(i * c.n_resblocks + j);
      for (int k = 0; k < 3; ++k) {
        Conv c1, c2;
        GSV_RC(make_conv(h, r + ".convs1." + std::to_string(k), ch, ch, c.rb_kernels[j], true, &c1));
        GSV_RC(make_conv(h, r + ".convs2." + std::to_string(k), ch, ch, c.rb_kernels[j], true, &c2));
        v->rb1.push_back(c1);
        v->rb2.push_back(c2);
      }
      if (big)
        for (int k = 0; k < 6; ++k) {
          VocAct a;
          GSV_RC(load_act(r + ".activations." + std::to_string(k) + ".act", ch, &a));
          v->acts.push_back(a);
        }
    }
  }
  if (big) {
    VocAct a;
    GSV_RC(load_act("activation_post.act", ch, &a));
    v->acts.push_back(a);
    float f[12];
    kaiser_sinc12(f);
    GSV_RC(up_f32(h, f, 12, &v->up12));
    GSV_RC(up_f32(h, f, 12, &v->dn12));
  }
  GSV_RC(make_conv(h, "conv_post", 1, ch, 7, c.bias_at_final != 0, &v->conv_post));
  h->staged.clear();
  h->finalized = true;
  v->finalized = true;
  return GSV_OK;
}

int gsv_vocoder_forward(gsv_vocoder_t* v, const float* mel, int F, float* wav, gsv_stream_t stream) {
  GSV_REQUIRE(v && v->finalized, "vocoder_forward: handle not finalized");
  GSV_REQUIRE(mel && wav && F >= 1, "vocoder_forward: empty input");
  hipStream_t s = (hipStream_t)stream;
  gsv_vits* h = &v->ctx;
  const auto& c = v->cfg;
  const size_t es = esz(h);
  const bool big = c.kind == 1;
  const int UIC = c.upsample_initial_channel;
  size_t maxel = (size_t)F * UIC;
  {
    long long Tn = F; int ch = UIC;
    for (int i = 0; i < c.n_ups; ++i) { Tn *= c.up_rates[i]; ch >>= 1; maxel = std::max(maxel, (size_t)Tn * ch); }
  }
  void* xin;
  GSV_RC(need(h, "voc_in", (size_t)F * v->cin_pad * es, &xin));
  GSV_HIP(hipMemsetAsync(xin, 0, (size_t)F * v->cin_pad * es, s));
  GSV_DISPATCH(h,
    hipLaunchKernelGGL(cf_to_cl_kernel<_Float16>, dim3(cdiv(F, 32), cdiv(c.in_channels, 32)), dim3(256), 0, s, mel, F, c.in_channels, (_Float16*)xin, v->cin_pad),
    hipLaunchKernelGGL(cf_to_cl_kernel<float>, dim3(cdiv(F, 32), cdiv(c.in_channels, 32)), dim3(256), 0, s, mel, F, c.in_channels, (float*)xin, v->cin_pad));
  void* gb[6];
  const char* gnames[6] = {"v0", "v1", "v2", "v3", "v4", "v5"};
  for (int i = 0; i < 6; ++i) GSV_RC(need(h, gnames[i], maxel * es, &gb[i]));
  void* cur = gb[3];
  { ConvOpt o; GSV_RC(conv(h, s, v->conv_pre, xin, v->cin_pad, F, cur, F, o)); }
  int Tn = F, ch = UIC, ai = 0;
  for (int i = 0; i < c.n_ups; ++i) {
    const int Tout = Tn * c.up_rates[i];
    ch >>= 1;
    void* xup = gb[0]; void* xt = gb[1]; void* R = gb[2]; void* xa = gb[5]; void* xs = (cur == gb[3]) ? gb[4] : gb[3];
    { ConvOpt ou; if (!big) { ou.pre_act = ACT_LRELU; ou.pre_slope = 0.1f; }
      GSV_RC(conv(h, s, v->ups[i], cur, ch * 2, Tn, xup, Tout, ou)); }
    for (int j = 0; j < c.n_resblocks; ++j) {
      const void* xr = xup;
      for (int k = 0; k < 3; ++k) {
        const Conv& c1 = v->rb1[(i * c.n_resblocks + j) * 3 + k];
        const Conv& c2 = v->rb2[(i * c.n_resblocks + j) * 3 + k];
        if (!big && c1.b && c2.b && c1.taps == c2.taps && conv_pair_eligible(h->dtype, ch, c1.taps, c.rb_dilations[j][k], Tout)) {
          ConvPairArgs pa;                                // narrow stages of the v4 HiFi-GAN vocoder: same fused pair as gsv_vits_decode
          pa.x = (const _Float16*)xr; pa.w1 = (const _Float16*)c1.w; pa.b1 = c1.b; pa.w2 = (const _Float16*)c2.w; pa.b2 = c2.b;
          pa.T = Tout; pa.C = ch; pa.taps = c1.taps; pa.dil = c.rb_dilations[j][k]; pa.ldx = ch; pa.ldy = ch;
          if (k < 2) { pa.y = (_Float16*)R; }
          else { pa.y = (_Float16*)xs; pa.scale = 1.f / (float)c.n_resblocks; pa.accumulate = j > 0; }
          if ((const void*)pa.y == xr) { pa.y = (_Float16*)xt; }
          GSV_RC(launch_conv_pair(pa, s));
          if (k < 2) { if (pa.y == (_Float16*)xt) { std::swap(xt, R); } xr = R; }
          continue;
        }
        ConvOpt o1; o1.dil = c.rb_dilations[j][k];
        ConvOpt o2; o2.res = xr; o2.ldr = ch;
        const void* in1 = xr;
        if (big) { GSV_RC(voc_act(v, s, v->acts[ai + 2 * k], xr, xa, Tout, ch)); in1 = xa; }
        else { o1.pre_act = ACT_LRELU; o1.pre_slope = 0.1f; o2.pre_act = ACT_LRELU; o2.pre_slope = 0.1f; }
        GSV_RC(conv(h, s, c1, in1, ch, Tout, xt, Tout, o1));
        const void* in2 = xt;
        if (big) { GSV_RC(voc_act(v, s, v->acts[ai + 2 * k + 1], xt, xa, Tout, ch)); in2 = xa; }
        if (k < 2) {
          GSV_RC(conv(h, s, c2, in2, ch, Tout, R, Tout, o2));
          xr = R;
        } else {
          o2.scale = 1.f / (float)c.n_resblocks; o2.accumulate = j > 0;
          GSV_RC(conv(h, s, c2, in2, ch, Tout, xs, Tout, o2));
        }
      }
      if (big) ai += 6;
    }
    cur = xs; Tn = Tout;
  }
  ConvOpt op; op.out_f32 = 1;
  const void* pin = cur;
  if (big) { GSV_RC(voc_act(v, s, v->acts[ai], cur, gb[5], Tn, ch)); pin = gb[5]; }
  else { op.pre_act = ACT_LRELU; op.pre_slope = 0.01f; }
  op.post_act = c.tanh_at_final ? ACT_TANH : ACT_CLAMP1;
  GSV_RC(conv(h, s, v->conv_post, pin, ch, Tn, wav, Tn, op));
  return GSV_OK;
}

}  // extern "C"

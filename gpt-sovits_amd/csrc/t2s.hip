// AR semantic-token decoder engine (H1-H5) for gfx950.
//
// State lives in HBM behind an opaque handle: weights in the engine dtype, a head-major KV
// arena [layer][k|v][row][head][pos][head_dim] (so one (row, head) stream is contiguous and a
// wave reads it as 1 KiB wave-instructions), per-row lengths/flags, and the token history.
// The reference re-concatenates the cache every step (t2s_model.py:186-187) and
// index_selects finished rows away on the host (:727-745); here rows are appended in place and
// finished rows are flagged on the device, so a decode step has no host synchronisation and
// is replayed as one hipGraph.
//
// Decode step = per layer 5 kernels (QKV+append, attention, out-proj, FFN1, FFN2), LayerNorm
// fused into the consumer's prologue, split-K across the waves of a workgroup with an LDS
// combine (no global partials), then logits + a one-wave-per-row sampling kernel.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "t2s_sample.h"
#include "t2s_mega.h"

namespace gsv {

typedef float f4v __attribute__((ext_vector_type(4)));

// =======================================================================================
// kernels
// =======================================================================================

// x[row] = E_text[id] + bert_proj(bert)[row] + alpha_t * pe[pos]     (H2; t2s_model.py:612-617)
// or       E_audio[tok] + alpha_a * pe[pos]                            (t2s_model.py:636-640)
// rows are packed per utterance: [x_0 .. x_{X-1}, y_0 .. y_{P-1}]
template <typename T>
__global__ void embed_prefill_kernel(const int* __restrict__ phones, const int* __restrict__ prompts,
                                     const int* __restrict__ row_off, const int* __restrict__ ph_off,
                                     const int* __restrict__ x_len, const float* __restrict__ e_text,
                                     const float* __restrict__ e_audio, const float* __restrict__ bertp,  // [sumX][d] or null
                                     const float* __restrict__ bert_bias, const float* __restrict__ pe, float alpha_t,
                                     float alpha_a, int P, int d, T* __restrict__ x) {
  const int b = blockIdx.y;
  const int i = blockIdx.x;  // position within the row's sequence
  const int X = x_len[b];
  if (i >= X + P) return;
  T* out = x + (long long)(row_off[b] + i) * d;
  if (i < X) {
    const int id = phones[ph_off[b] + i];
    const float* e = e_text + (long long)id * d;
    const float* bp = bertp ? bertp + (long long)(ph_off[b] + i) * d : nullptr;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
      float v = e[c] + (bp ? bp[c] : bert_bias[c]);
      out[c] = (T)(v + alpha_t * pe[(long long)i * d + c]);
    }
  } else {
    const int tok = prompts[b * P + (i - X)];
    const float* e = e_audio + (long long)tok * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[c] = (T)(e[c] + alpha_a * pe[(long long)(i - X) * d + c]);
  }
}

// scatter the prefill K/V (columns d..3d of qkv) into the head-major cache
template <typename T>
__global__ void kv_scatter_kernel(const T* __restrict__ qkv, const int* __restrict__ row_off, const int* __restrict__ x_len,
                                  int P, int d, int H, int smax, T* __restrict__ kc, T* __restrict__ vc) {
  const int b = blockIdx.y, i = blockIdx.x;
  if (i >= x_len[b] + P) return;
  const int hd = d / H;
  const T* src = qkv + (long long)(row_off[b] + i) * 3 * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const int h = c / hd, e = c - h * hd;
    const long long o = (((long long)b * H + h) * smax + i) * hd + e;
    kc[o] = src[d + c];
    vc[o] = src[2 * d + c];
  }
}

// Prefill attention (H3): one thread per query, keys streamed with a block-uniform address.
// Mask (t2s_model.py:655-683): text rows see the text keys; audio rows see all text + causal audio.
template <typename T, int HD>
__global__ void prefill_attn_kernel(const T* __restrict__ qkv, const T* __restrict__ kc, const T* __restrict__ vc,
                                    const int* __restrict__ row_off, const int* __restrict__ x_len, int P, int d, int H,
                                    int smax, T* __restrict__ out) {
  const int b = blockIdx.z, h = blockIdx.y;
  const int X = x_len[b], S = X + P;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int q0 = blockIdx.x * blockDim.x;
  if (q0 >= S) return;
  const bool valid = i < S;
  const int nk = valid ? (i < X ? X : i + 1) : 0;
  // block-uniform upper bound of the key loop
  const int qlast = min(q0 + (int)blockDim.x, S) - 1;
  const int nk_max = (qlast < X) ? X : qlast + 1;
  float q[HD], acc[HD];
  const T* qp = qkv + (long long)(row_off[b] + (valid ? i : 0)) * 3 * d + h * HD;
  const float scale = rsqrtf((float)HD);
#pragma unroll
  for (int e = 0; e < HD; ++e) { q[e] = to_f(qp[e]) * scale; acc[e] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const T* kb = kc + ((long long)b * H + h) * smax * HD;
  const T* vb = vc + ((long long)b * H + h) * smax * HD;
  for (int j = 0; j < nk_max; ++j) {
    const T* kr = kb + (long long)j * HD;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < HD; ++e) s += q[e] * to_f(kr[e]);
    if (j < nk) {
      const float mn = fmaxf(m, s);
      const float corr = expf(m - mn);
      const float p = expf(s - mn);
      const T* vr = vb + (long long)j * HD;
      l = l * corr + p;
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[e] = acc[e] * corr + p * to_f(vr[e]);
      m = mn;
    }
  }
  if (valid) {
    T* o = out + (long long)(row_off[b] + i) * d + h * HD;
    const float inv = 1.f / l;
#pragma unroll
    for (int e = 0; e < HD; ++e) o[e] = (T)(acc[e] * inv);
  }
}

// ---------------------------------------------------------------------------------------
// Prefill attention on MFMA (fp16, head dim 32), same construction as the DiT kernel (attn.hip): transposed scores
// S^T = K Q^T (one 16x16x32 MFMA per 16 keys x 16 queries: k = head dim), the probabilities a lane holds are the B
// operand of O^T = V^T P^T, keys dealt in 32-key chunks to the 4 waves with online softmax, LDS combine.  K comes
// straight from the head-major cache (a 16-key fragment is 1 KB contiguous), V^T from a per-prefill scratch.
// Mask (t2s_model.py:655-683): text queries see the text keys; audio queries see all text + causal audio.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prefill_vt_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ row_off,
                                                         const int* __restrict__ x_len, int P, int d, int H, int spad,
                                                         _Float16* __restrict__ vt) {
  __shared__ _Float16 tile[32][34];
  const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * 32;
  const int S = x_len[b] + P;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int j = j0 + i;
    tile[i][tx] = j < S ? qkv[(long long)(row_off[b] + j) * 3 * d + 2 * d + h * 32 + tx] : (_Float16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) vt[(((long long)b * H + h) * 32 + i) * spad + j0 + tx] = tile[tx][i];
}

// The same V^T tiles plus the head-major K/V cache rows of those 32 positions: one launch per layer instead of
// kv_scatter_kernel + prefill_vt_kernel (a 32 x 32 tile of one head is 2 KB contiguous in either cache).
__global__ __launch_bounds__(256) void prefill_kvt_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ row_off,
                                                          const int* __restrict__ x_len, int P, int d, int H, int smax, int spad,
                                                          _Float16* __restrict__ kc, _Float16* __restrict__ vc,
                                                          _Float16* __restrict__ vt) {
  __shared__ _Float16 tile[32][34];
  const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * 32;
  const int S = x_len[b] + P;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int j = j0 + i;
    _Float16 v = (_Float16)0.f;
    if (j < S) {
      const _Float16* src = qkv + (long long)(row_off[b] + j) * 3 * d + h * 32 + tx;
      const long long o = (((long long)b * H + h) * smax + j) * 32 + tx;
      v = src[2 * d];
      kc[o] = src[d];
      vc[o] = v;
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) vt[(((long long)b * H + h) * 32 + i) * spad + j0 + tx] = tile[tx][i];
}

template <int QT>
__global__ __launch_bounds__(256) void prefill_flash32_f16_kernel(const _Float16* __restrict__ qkv, const _Float16* __restrict__ kc,
                                                                   const _Float16* __restrict__ vt, const int* __restrict__ row_off,
                                                                   const int* __restrict__ x_len, int P, int d, int H, int smax,
                                                                   int spad, _Float16* __restrict__ out) {
  constexpr int BQ = 16 * QT, LDO = 36;
  __shared__ float Os[4][BQ][LDO];
  __shared__ float Ms[4][BQ], Ls[4][BQ];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;
  const int X = x_len[b], S = X + P;
  if (q0 >= S) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const float scale = rsqrtf(32.f);
  h8 qf[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t)
    qf[t] = *(const h8*)(qkv + (long long)(row_off[b] + min(q0 + 16 * t + r, S - 1)) * 3 * d + h * 32 + g * 8);
  const _Float16* kb = kc + ((long long)b * H + h) * smax * 32 + g * 8;
  const _Float16* vb = vt + (((long long)b * H + h) * 32 + r) * spad + 8 * g;
  const int kra = 8 * (r >> 2) + (r & 3);           // permuted K rows: the lane's 8 scores are 8 consecutive keys (attn.hip)
  f4 o[QT][2];
  float m[QT], l[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) { m[t] = -INFINITY; l[t] = 0.f; o[t][0] = (f4){0.f, 0.f, 0.f, 0.f}; o[t][1] = o[t][0]; }
  const int qlast = min(q0 + BQ, S) - 1;
  const int nk = qlast < X ? X : qlast + 1;          // workgroup-uniform bound of the key range
  const int nchunks = (nk + 31) >> 5, lastc = nchunks - 1;
  struct KV { h8 ka, kb2; h8 v[2]; };
  auto fetch = [&](KV& f, int c) {
    const int key0 = c << 5;
    f.ka = *(const h8*)(kb + (long long)min(key0 + kra, S - 1) * 32);
    f.kb2 = *(const h8*)(kb + (long long)min(key0 + kra + 4, S - 1) * 32);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) f.v[dt] = *(const h8*)(vb + (long long)(dt * 16) * spad + key0);
  };
  auto process = [&](const KV& f, int c, bool valid) {
    const int key0 = c << 5;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int qi = q0 + 16 * t + r;
      const int lim = qi < X ? X : qi + 1;           // keys [0, lim) are visible to query qi
      f4 sa = (f4){0.f, 0.f, 0.f, 0.f}, sb = sa;
      sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ka, qf[t], sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.kb2, qf[t], sb, 0, 0, 0);
      float p[8];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = (valid && key0 + 8 * g + i < lim) ? sa[i] * scale : -INFINITY;
        p[4 + i] = (valid && key0 + 8 * g + 4 + i < lim) ? sb[i] * scale : -INFINITY;
        mx = fmaxf(mx, fmaxf(p[i], p[4 + i]));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m[t], mx);
      const float ms = mnew == -INFINITY ? 0.f : mnew;   // a causal query may see none of this wave's keys yet
      const float alpha = __expf(m[t] - ms);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - ms); ps += p[i]; }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l[t] = l[t] * alpha + ps;
      m[t] = mnew;
      const h8 pf = (h8){(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3], (_Float16)p[4], (_Float16)p[5], (_Float16)p[6], (_Float16)p[7]};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        o[t][dt] *= alpha;
        o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.v[dt], pf, o[t][dt], 0, 0, 0);
      }
    }
  };
#define GSV_PIN2() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  KV fA, fB;
  fetch(fA, min(wave, lastc));
  for (int c = wave; c < nchunks; c += 8) {
    fetch(fB, min(c + 4, lastc));
    GSV_PIN2();
    process(fA, c, true);
    GSV_PIN2();
    fetch(fA, min(c + 8, lastc));
    GSV_PIN2();
    process(fB, c + 4, c + 4 < nchunks);
    GSV_PIN2();
  }
#undef GSV_PIN2
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) *(f4*)&Os[wave][16 * t + r][dt * 16 + 4 * g] = o[t][dt];
    if (g == 0) { Ms[wave][16 * t + r] = m[t]; Ls[wave][16 * t + r] = l[t]; }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < BQ * 8; it += 256) {
    const int qq = it >> 3, d4 = (it & 7) * 4;
    if (q0 + qq >= S) continue;
    const float mt = fmaxf(fmaxf(Ms[0][qq], Ms[1][qq]), fmaxf(Ms[2][qq], Ms[3][qq]));   // finite: key 0 is visible to every query
    float den = 0.f;
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(Ms[w][qq] - mt);
      den += e * Ls[w][qq];
      acc += *(const f4*)&Os[w][qq][d4] * e;
    }
    const float inv = 1.f / den;
    *(h4*)(out + (long long)(row_off[b] + q0 + qq) * d + h * 32 + d4) =
        (h4){(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
  }
}

// gather each row's last prefill position of the fp32 pre-LN2 stream into the decode buffer
__global__ void gather_last_kernel(const float* __restrict__ y2, const int* __restrict__ row_off,
                                   const int* __restrict__ x_len, int P, int d, float* __restrict__ ybuf) {
  const int b = blockIdx.x;
  const float* src = y2 + (long long)(row_off[b] + x_len[b] + P - 1) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) ybuf[(long long)b * d + c] = src[c];
}

// ---------------------------------------------------------------------------------------
// Decode-step skinny GEMM:  Y[b][n] = sum_k X[b][k] W[n][k]   for b < B (B <= 16*CB)
// Workgroup = one 16-row tile of W (n0..n0+15), NW waves each owning K/NW of the contraction,
// partial 16x16 accumulators combined through LDS.  MFMA 16x16x32 f16 / 16x16x4 f32.
// ---------------------------------------------------------------------------------------
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_RELU = 2, EPI_LOGITS = 3 };

struct DecGemmArgs {
  // X source (exactly one of yin / xin)
  const float* yin;     // fp32 pre-LN stream [B][K] -> LN(gamma,beta) (or plain convert if gamma==null) -> LDS
  const float* gamma;
  const float* beta;
  float* xres_out;      // if non-null, workgroup 0 writes the normalised fp32 rows here [B][K]
  const void* xin;      // T activations [B][K] read straight from HBM/L2
  const void* w;        // T [N][K]
  const float* bias;    // [N] or null
  int B, K, N;
  int epi;
  // epilogue targets
  void* out_t;          // EPI_RELU: T [B][N]; EPI_QKV: q buffer T [B][d]
  float* out_f;         // EPI_RESID / EPI_LOGITS: fp32 [B][N]
  const float* xres;    // EPI_RESID: fp32 residual [B][N]
  void* kc; void* vc;   // EPI_QKV: cache bases for this layer
  const int* kv_len;    // EPI_QKV
  const int* active;
  int d, H, smax;
};

template <typename T> struct Frag16;
template <> struct Frag16<_Float16> { typedef h8 type; static constexpr int KS = 32; };
template <> struct Frag16<float> { typedef f4 type; static constexpr int KS = 16; };

__device__ __forceinline__ void mma16(f4v& acc, const h8& a, const h8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f4v& acc, const f4& a, const f4& b) {
  // lane group g = lane>>4 holds k = k0 + 4g + i; MFMA i contracts {k0+i, k0+4+i, k0+8+i, k0+12+i}
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
}

template <typename T, int CB, int NW, int KSL, int KV4>
__global__ __launch_bounds__(NW * 64) void dec_gemm_kernel(DecGemmArgs a) {
  // KSL = K / NW (k-slice per wave, compile time so every load is issued up front);
  // KV4 > 0: LayerNorm prologue with K = 32*KV4, 8 threads per row, the row held in registers.
  typedef typename Frag16<T>::type F;
  constexpr int G = DT<T>::G;
  constexpr int KS = Frag16<T>::KS;  // k per MFMA group step
  constexpr int NKS = KSL / KS;
  constexpr bool LNPRO = KV4 > 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * CB * 16;   // batch-row block of this workgroup (grid.y > 1 only for narrow-N kernels)
  const int K = a.K;
  const int ldx = K + G;  // padded LDS row (elements): breaks the power-of-two row stride
  T* xs = (T*)smem;
  float* red = (float*)smem;   // the split-K combine buffer reuses the X image (barrier below)

  const int rowl = lane & 15, kg = lane >> 4;
  const int kbeg = wave * KSL;
  const bool wok = (n0 + rowl) < a.N;
  const T* wrow = (const T*)a.w + (long long)(wok ? n0 + rowl : 0) * K + kbeg + G * kg;
  // weight stream first: its HBM latency overlaps the LayerNorm prologue
  F af[NKS];
#pragma unroll
  for (int i = 0; i < NKS; ++i) af[i] = *(const F*)(wrow + i * KS);   // unconditional: wrow is clamped to row 0 when n0 + rowl >= N,
                                                                       // and those output rows are never stored

  // epilogue operands (bias, residual, row state) are fetched now, not after the LDS combine, so the
  // kernel has one exposed memory round trip instead of two
  constexpr int ITEMS = (CB + NW - 1) / NW;
  f4 pre_bias[ITEMS], pre_res[ITEMS];
  int pre_act[ITEMS], pre_pos[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = tid + it * NW * 64;
    const int ln = item & 63;
    const int b = b0 + (item >> 6) * 16 + (ln & 15);
    const int n = n0 + 4 * (ln >> 4);
    const bool ok = item < CB * 64 && b < a.B && n < a.N;
    pre_bias[it] = (f4){0.f, 0.f, 0.f, 0.f};
    pre_res[it] = (f4){0.f, 0.f, 0.f, 0.f};
    pre_act[it] = 0; pre_pos[it] = 0;
    if (ok) {
      if (a.bias) {
        if (n + 3 < a.N) pre_bias[it] = *(const f4*)(a.bias + n);
        else for (int i = 0; i < 4; ++i) if (n + i < a.N) pre_bias[it][i] = a.bias[n + i];
      }
      if (a.epi == EPI_RESID) {
        if (n + 3 < a.N) pre_res[it] = *(const f4*)(a.xres + (long long)b * a.N + n);
        else for (int i = 0; i < 4; ++i) if (n + i < a.N) pre_res[it][i] = a.xres[(long long)b * a.N + n + i];
      } else if (a.epi == EPI_QKV) {
        pre_act[it] = a.active[b];
        pre_pos[it] = a.kv_len[b];
      }
    }
  }

  if (LNPRO) {
    constexpr int TPR = 8;                       // threads per row
    constexpr int RPP = NW * 64 / TPR;           // rows per pass
    constexpr int NV = KV4 > 0 ? KV4 : 1;
    const int sub = tid & (TPR - 1);
    f4 gm[NV], bt[NV];
    if (a.gamma) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        gm[j] = *(const f4*)(a.gamma + (j * TPR + sub) * 4);
        bt[j] = *(const f4*)(a.beta + (j * TPR + sub) * 4);
      }
    }
    for (int r0 = 0; r0 < CB * 16; r0 += RPP) {
      const int row = r0 + tid / TPR;
      if (row >= CB * 16) break;
      const bool live = b0 + row < a.B;
      f4 v[NV];
      const float* src = a.yin + (long long)(live ? b0 + row : 0) * K;
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = live ? *(const f4*)(src + (j * TPR + sub) * 4) : (f4){0.f, 0.f, 0.f, 0.f};
      if (a.gamma) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s / (float)K;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { float dl = v[j][e] - mean; q += dl * dl; }
        }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const float rstd = rsqrtf(q / (float)K + 1e-5f);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = live ? (v[j][e] - mean) * rstd * gm[j][e] + bt[j][e] : 0.f;
        }
      }
      T* dst = xs + (long long)row * ldx;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        typedef T T4 __attribute__((ext_vector_type(4)));
        *(T4*)(dst + (j * TPR + sub) * 4) = (T4){(T)v[j][0], (T)v[j][1], (T)v[j][2], (T)v[j][3]};
        if (a.xres_out && blockIdx.x == 0 && live) *(f4*)(a.xres_out + (long long)(b0 + row) * K + (j * TPR + sub) * 4) = v[j];
      }
    }
    __syncthreads();
  }

  f4v acc[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) acc[cb] = (f4v){0.f, 0.f, 0.f, 0.f};
  const T* xg = (const T*)a.xin;
  F bf[NKS][CB];
#pragma unroll
  for (int i = 0; i < NKS; ++i) {
    const int k = kbeg + i * KS + G * kg;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int brow = cb * 16 + rowl;
      if (LNPRO) bf[i][cb] = *(const F*)(xs + (long long)brow * ldx + k);
      else if (b0 + brow < a.B) bf[i][cb] = *(const F*)(xg + (long long)(b0 + brow) * K + k);
      else {
#pragma unroll
        for (int e = 0; e < G; ++e) bf[i][cb][e] = 0;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NKS; ++i)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) mma16(acc[cb], af[i], bf[i][cb]);
  // combine the NW partial tiles through LDS (fixed summation order: deterministic)
  if (LNPRO) __syncthreads();   // every wave has its B fragments in registers: the X image may be overwritten
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) *(f4v*)(red + ((wave * CB + cb) * 64 + lane) * 4) = acc[cb];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = tid + it * NW * 64;
    if (item >= CB * 64) break;
    const int cb = item >> 6, ln = item & 63;
    f4v v = (f4v){0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < NW; ++w) v += *(const f4v*)(red + ((w * CB + cb) * 64 + ln) * 4);
    // D layout of 16x16: col = ln & 15 (batch row), rows 4*(ln>>4) + i (output channel)
    const int b = b0 + cb * 16 + (ln & 15);
    const int n = n0 + 4 * (ln >> 4);
    if (b >= a.B || n >= a.N) continue;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = v[i] + pre_bias[it][i];
    const bool full = n + 3 < a.N;
    if (a.epi == EPI_RESID) {
      float* op = a.out_f + (long long)b * a.N + n;
      if (full) *(f4*)op = (f4){o[0] + pre_res[it][0], o[1] + pre_res[it][1], o[2] + pre_res[it][2], o[3] + pre_res[it][3]};
      else for (int i = 0; i < 4; ++i) if (n + i < a.N) op[i] = o[i] + pre_res[it][i];
    } else if (a.epi == EPI_RELU) {
      T* op = (T*)a.out_t + (long long)b * a.N + n;
      typedef T T4 __attribute__((ext_vector_type(4)));
      if (full) *(T4*)op = (T4){(T)fmaxf(o[0], 0.f), (T)fmaxf(o[1], 0.f), (T)fmaxf(o[2], 0.f), (T)fmaxf(o[3], 0.f)};
      else for (int i = 0; i < 4; ++i) if (n + i < a.N) op[i] = (T)fmaxf(o[i], 0.f);
    } else if (a.epi == EPI_LOGITS) {
      float* op = a.out_f + (long long)b * a.N + n;
      for (int i = 0; i < 4; ++i) if (n + i < a.N) op[i] = o[i];     // N = 1025: rows are not 16-byte aligned
    } else {  // EPI_QKV: n in [0,3d): q -> qbuf, k/v -> cache row kv_len[b]
      typedef T T4 __attribute__((ext_vector_type(4)));
      const int d = a.d, hd = d / a.H;
      const int which = n / d, c = n - which * d;
      const T4 ov = (T4){(T)o[0], (T)o[1], (T)o[2], (T)o[3]};
      if (which == 0) {
        *(T4*)((T*)a.out_t + (long long)b * d + c) = ov;
      } else if (pre_act[it]) {
        const int h = c / hd, e = c - h * hd;
        const int pos = pre_pos[it];
        if (pos < a.smax) {
          T* base = (T*)(which == 1 ? a.kc : a.vc);
          *(T4*)(base + (((long long)b * a.H + h) * a.smax + pos) * hd + e) = ov;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Decode attention (H4): one workgroup per (row, head); the KV stream of that pair is
// contiguous [pos][hd], read as 16-byte lane loads (1 KiB per wave-instruction), keys dealt
// round-robin to the 4 waves, per-lane online softmax, one LDS combine at the end.
// This is the HBM-bound kernel of the step: algorithmic bytes = 2*hd*sizeof(T) per cached key.
// ---------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, const int* __restrict__ kv_len,
                                                          const int* __restrict__ active, int H, int smax,
                                                          T* __restrict__ out) {
  constexpr int G = DT<T>::G;
  constexpr int LPK = HD / G;      // lanes per key
  constexpr int KPI = 64 / LPK;    // keys per wave-instruction
  typedef typename Frag16<T>::type F;
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane % LPK, slot = lane / LPK;
  const int d = H * HD;
  const T* kb = kc + ((long long)b * H + h) * smax * HD;
  const T* vb = vc + ((long long)b * H + h) * smax * HD;
  // q and the first SPEC key groups are fetched before the row state is known: the arena is
  // allocated to smax, so the addresses are valid and groups beyond kv_len are simply masked.
  // This takes the kv_len -> K/V dependency off the critical path (prompt + text is always longer
  // than SPEC*4*KPI keys in practice, so nothing extra is streamed).
  // The K/V stream (440 MB per step at the benchmark shape, read exactly once per step) is loaded NON-TEMPORAL so that it
  // does not evict the 152 MB of decoder weights from the 256 MB Infinity Cache between steps (GSV_KV_TEMPORAL=1 at build
  // time restores plain loads for A/B).
#ifdef GSV_KV_TEMPORAL
#define KVLOAD(p) (*(const F*)(p))
#else
#define KVLOAD(p) __builtin_nontemporal_load((const F*)(p))
#endif
  constexpr int SPEC = 3;
  const F qv = *(const F*)(q + (long long)b * d + h * HD + part * G);
  F ksp[SPEC], vsp[SPEC];
#pragma unroll
  // Every K/V load below is UNCONDITIONAL with a clamped key index (lanes past the end re-read the last valid row, which
  // costs no extra traffic, and are masked in consume()): per-lane conditional loads were compiled as one exec-masked
  // block per group with vmcnt(0) at each join, i.e. the NEXT groups were five dependent memory round trips.
  for (int i = 0; i < SPEC; ++i) {
    const int j = min((wave + 4 * i) * KPI + slot, smax - 1);
    ksp[i] = KVLOAD(kb + (long long)j * HD + part * G);
    vsp[i] = KVLOAD(vb + (long long)j * HD + part * G);
  }
  if (!active[b]) return;
  const int n = min(kv_len[b] + 1, smax);
  float qf[G];
  {
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int i = 0; i < G; ++i) qf[i] = to_f(qv[i]) * scale;
  }
  float m = -INFINITY, l = 0.f, acc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) acc[i] = 0.f;
  auto consume = [&](const F& kv, const F& vv, bool ok) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) s += qf[i] * to_f(kv[i]);
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
    if (ok) {
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn);
      const float p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int i = 0; i < G; ++i) acc[i] = acc[i] * corr + p * to_f(vv[i]);
      m = mn;
    }
  };
  // second batch: as soon as kv_len is known, the next NEXT groups are requested BEFORE the speculative
  // ones are consumed, so both batches share one memory round trip (up to (SPEC+NEXT)*4*KPI = 512 fp16 keys)
  constexpr int NEXT = 5;
  F kn[NEXT], vn[NEXT];
#pragma unroll
  for (int i = 0; i < NEXT; ++i) {
    const int j = min((wave + 4 * (SPEC + i)) * KPI + slot, n - 1);
    kn[i] = KVLOAD(kb + (long long)j * HD + part * G);
    vn[i] = KVLOAD(vb + (long long)j * HD + part * G);
  }
#pragma unroll
  for (int i = 0; i < SPEC; ++i) consume(ksp[i], vsp[i], (wave + 4 * i) * KPI + slot < n);
#pragma unroll
  for (int i = 0; i < NEXT; ++i) consume(kn[i], vn[i], (wave + 4 * (SPEC + i)) * KPI + slot < n);
  // keys past the first (SPEC+NEXT) groups (long sequences): batches of TB groups, two register sets alternated so the
  // next batch is in flight while the current one is consumed (all loads unconditional with a clamped key index).
  {
    constexpr int TB = 4, STR = 4 * KPI;
    int j0 = (wave + 4 * (SPEC + NEXT)) * KPI;
    if (j0 < n) {
      F ka[TB], va[TB], kq[TB], vq[TB];
      auto loadb = [&](F* kk, F* vv, int base) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int jc = min(base + i * STR + slot, n - 1);
          kk[i] = KVLOAD(kb + (long long)jc * HD + part * G);
          vv[i] = KVLOAD(vb + (long long)jc * HD + part * G);
        }
      };
      auto consb = [&](const F* kk, const F* vv, int base) {
#pragma unroll
        for (int i = 0; i < TB; ++i) consume(kk[i], vv[i], base + i * STR + slot < n);
      };
      loadb(ka, va, j0);
      for (; j0 < n; j0 += 2 * TB * STR) {
        loadb(kq, vq, j0 + TB * STR);
        consb(ka, va, j0);
        loadb(ka, va, j0 + 2 * TB * STR);
        consb(kq, vq, j0 + TB * STR);
      }
    }
  }
  // combine: global max, rescale, sum over key slots (lanes with equal `part`) and waves
  __shared__ float s_m[4];
  __shared__ float s_acc[4][HD + 1];
  float wm = wave_max(m);
  if (lane == 0) s_m[wave] = wm;
  __syncthreads();
  const float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  const float f = (m == -INFINITY) ? 0.f : __expf(m - M);
  l *= f;
#pragma unroll
  for (int i = 0; i < G; ++i) acc[i] *= f;
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int i = 0; i < G; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
  }
  if (lane < LPK) {
#pragma unroll
    for (int i = 0; i < G; ++i) s_acc[wave][lane * G + i] = acc[i];
    if (lane == 0) s_acc[wave][HD] = l;
  }
  __syncthreads();
  if (threadIdx.x < HD) {
    const int e = threadIdx.x;
    const float L = s_acc[0][HD] + s_acc[1][HD] + s_acc[2][HD] + s_acc[3][HD];
    const float v = s_acc[0][e] + s_acc[1][e] + s_acc[2][e] + s_acc[3][e];
    out[(long long)b * d + h * HD + e] = (T)(v / L);
  }
#undef KVLOAD
}

// ---------------------------------------------------------------------------------------
// Fused decode kernel A = LayerNorm prologue + QKV projection of ONE head + in-place KV append +
// attention over the cached keys, for RPG = 2 batch rows per workgroup (grid = heads x rows/2 = 256
// workgroups at B = 32: one per CU).  It replaces the separate QKV and attention launches: the
// K/V stream of the (row, head) pairs is requested at kernel entry (speculatively, before kv_len is
// known) so the HBM latency of the cache overlaps the weight fetch, the LayerNorm and the MFMAs.
//   waves 0,1 -> row 0, waves 2,3 -> row 1 for the attention; all 4 waves split K for the projection.
// ---------------------------------------------------------------------------------------
struct QkvAttnArgs {
  const float* yin; const float* gamma; const float* beta; float* xres_out;
  const void* w; const float* bias;           // [3d][K] and [3d]
  void* kc; void* vc; const int* kv_len; const int* active;
  void* out;                                  // T [B][d]
  int B, d, H, smax;
};

template <typename T, int HD, int KD>
__global__ __launch_bounds__(256) void dec_qkv_attn_kernel(QkvAttnArgs a) {
  typedef typename Frag16<T>::type F;
  constexpr int G = DT<T>::G;
  constexpr int KS = Frag16<T>::KS;
  constexpr int RPG = 2;
  constexpr int LPK = HD / G, KPI = 64 / LPK;
  constexpr int EPL = KD / 64;                 // LayerNorm elements per lane
  constexpr int LDXS = KD + G;
  constexpr int NKS_ALL = KD / KS;             // k-steps of the projection
  constexpr int NKS_W = NKS_ALL / 4;           // per wave
  constexpr int KCH = NKS_W > 4 ? 4 : NKS_W;   // k-steps in flight per wave
  static_assert(NKS_ALL % 4 == 0 && NKS_W % KCH == 0, "K split");
  __shared__ __attribute__((aligned(16))) T xs[(RPG + 1) * LDXS];
  __shared__ __attribute__((aligned(16))) float red[4 * 6 * 64 * 4];
  __shared__ float qkv_s[RPG][3 * HD];
  __shared__ float s_m[4];
  __shared__ float s_acc[4][HD + 1];

  const int h = blockIdx.x, rg = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, H = a.H, smax = a.smax;
  const int part = lane % LPK, slot = lane / LPK;
  const int arow = wave >> 1, half = wave & 1;          // attention: row of this wave, key-group parity
  const int b_att = rg * RPG + arow;
  const bool row_ok = b_att < a.B;
  const int bsafe = row_ok ? b_att : 0;
  const T* kb = (const T*)a.kc + ((long long)bsafe * H + h) * smax * HD;
  const T* vb = (const T*)a.vc + ((long long)bsafe * H + h) * smax * HD;

  // ---- (1) speculative K/V groups of this wave: group g = half + 2*i holds keys [g*KPI, (g+1)*KPI)
  constexpr int SPEC = 4, NEXT = 6;
  F ksp[SPEC], vsp[SPEC];
#pragma unroll
  for (int i = 0; i < SPEC; ++i) {
    const int j = (half + 2 * i) * KPI + slot;
    if (j < smax) { ksp[i] = *(const F*)(kb + (long long)j * HD + part * G); vsp[i] = *(const F*)(vb + (long long)j * HD + part * G); }
    else {
#pragma unroll
      for (int e = 0; e < G; ++e) { ksp[i][e] = 0; vsp[i][e] = 0; }
    }
  }
  const int act = row_ok ? a.active[bsafe] : 0;
  const int nold = row_ok ? min(a.kv_len[bsafe], smax - 1) : 0;   // cached keys; the new key goes to position nold

  // ---- (2) LayerNorm prologue: wave w < RPG normalises row w (fp32, two-pass), writes T copy to LDS
  if (wave < RPG) {
    const int b = rg * RPG + wave;
    float v[EPL];
    const bool live = b < a.B;
#pragma unroll
    for (int i = 0; i < EPL; ++i) v[i] = live ? a.yin[(long long)b * KD + lane + 64 * i] : 0.f;
    if (a.gamma) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) sum += v[i];
      const float mean = wave_sum(sum) / (float)KD;
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) { float dl = v[i] - mean; q2 += dl * dl; }
      const float rstd = rsqrtf(wave_sum(q2) / (float)KD + 1e-5f);
#pragma unroll
      for (int i = 0; i < EPL; ++i) v[i] = live ? (v[i] - mean) * rstd * a.gamma[lane + 64 * i] + a.beta[lane + 64 * i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      xs[wave * LDXS + lane + 64 * i] = (T)v[i];
      if (a.xres_out && h == 0 && live) a.xres_out[(long long)b * KD + lane + 64 * i] = v[i];
    }
  } else if (wave == RPG) {
    for (int c = lane; c < LDXS; c += 64) xs[RPG * LDXS + c] = (T)0.f;     // zero row for the unused MFMA columns
  }

  // ---- (3) second K/V batch now that kv_len is known (same round trip as the speculative one)
  F kn[NEXT], vn[NEXT];
#pragma unroll
  for (int i = 0; i < NEXT; ++i) {
    const int j = (half + 2 * (SPEC + i)) * KPI + slot;
    if (j < nold) { kn[i] = *(const F*)(kb + (long long)j * HD + part * G); vn[i] = *(const F*)(vb + (long long)j * HD + part * G); }
    else {
#pragma unroll
      for (int e = 0; e < G; ++e) { kn[i][e] = 0; vn[i][e] = 0; }
    }
  }

  // ---- (4) QKV projection of head h: 6 tiles of 16 weight rows (q|k|v x 2), K split over the 4 waves
  f4v acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j] = (f4v){0.f, 0.f, 0.f, 0.f};
  const int rowl = lane & 15, kg = lane >> 4;
  const T* wbase = (const T*)a.w;
  __syncthreads();                                                           // xs ready
  for (int kc0 = 0; kc0 < NKS_W; kc0 += KCH) {
    F af[KCH][6];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int k = (wave * NKS_W + kc0 + i) * KS + G * kg;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int wrow = (j >> 1) * d + h * HD + (j & 1) * 16 + rowl;          // q rows, k rows, v rows of head h
        af[i][j] = *(const F*)(wbase + (long long)wrow * KD + k);
      }
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int k = (wave * NKS_W + kc0 + i) * KS + G * kg;
      const F bf = *(const F*)(xs + (rowl < RPG ? rowl : RPG) * LDXS + k);
#pragma unroll
      for (int j = 0; j < 6; ++j) mma16(acc[j], af[i][j], bf);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) *(f4v*)(red + ((wave * 6 + j) * 64 + lane) * 4) = acc[j];
  __syncthreads();
  if (tid < RPG * 3 * HD) {
    const int r = tid / (3 * HD), nl = tid - r * (3 * HD);     // nl: 0..31 q, 32..63 k, 64..95 v
    const int j = nl >> 4, rr = nl & 15;
    const int ln = (rr >> 2) * 16 + r, i = rr & 3;             // D layout: col = batch row, row = 4*(ln>>4)+i
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((w * 6 + j) * 64 + ln) * 4 + i];
    const int which = nl / HD, e = nl - which * HD;
    v += a.bias[which * d + h * HD + e];
    qkv_s[r][nl] = v;
    const int b = rg * RPG + r;
    if (which > 0 && b < a.B && a.active[b]) {                  // append the new key / value row in place
      const int pos = min(a.kv_len[b], smax - 1);
      T* base = (T*)(which == 1 ? a.kc : a.vc);
      base[(((long long)b * H + h) * smax + pos) * HD + e] = (T)v;
    }
  }
  __syncthreads();

  // ---- (5) attention of (b_att, h): this wave's key groups + (half == 0) the new key from LDS
  float qf[G];
  {
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int i = 0; i < G; ++i) qf[i] = to_f((T)qkv_s[arow][part * G + i]) * scale;
  }
  float m = -INFINITY, l = 0.f, oacc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) oacc[i] = 0.f;
  auto consume = [&](const float* kf, const float* vf, bool ok) {
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) sc += qf[i] * kf[i];
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) sc += __shfl_xor(sc, o, 64);
    if (ok) {
      const float mn = fmaxf(m, sc);
      const float corr = __expf(m - mn);
      const float p = __expf(sc - mn);
      l = l * corr + p;
#pragma unroll
      for (int i = 0; i < G; ++i) oacc[i] = oacc[i] * corr + p * vf[i];
      m = mn;
    }
  };
  auto consume_frag = [&](const F& kv, const F& vv, bool ok) {
    float kf[G], vf[G];
#pragma unroll
    for (int i = 0; i < G; ++i) { kf[i] = to_f(kv[i]); vf[i] = to_f(vv[i]); }
    consume(kf, vf, ok);
  };
  const bool live_row = row_ok && act;
#pragma unroll
  for (int i = 0; i < SPEC; ++i) consume_frag(ksp[i], vsp[i], live_row && (half + 2 * i) * KPI + slot < nold);
#pragma unroll
  for (int i = 0; i < NEXT; ++i) consume_frag(kn[i], vn[i], live_row && (half + 2 * (SPEC + i)) * KPI + slot < nold);
  if (live_row) {
    for (int j0 = (half + 2 * (SPEC + NEXT)) * KPI; j0 < nold; j0 += 2 * KPI) {
      const int j = j0 + slot;
      const bool ok = j < nold;
      F kv, vv;
      if (ok) { kv = *(const F*)(kb + (long long)j * HD + part * G); vv = *(const F*)(vb + (long long)j * HD + part * G); }
      else {
#pragma unroll
        for (int e = 0; e < G; ++e) { kv[e] = 0; vv[e] = 0; }
      }
      consume_frag(kv, vv, ok);
    }
  }
  {
    // the key/value just produced (position nold), rounded to the cache dtype exactly as a later step reads it
    float kf[G], vf[G];
#pragma unroll
    for (int i = 0; i < G; ++i) { kf[i] = to_f((T)qkv_s[arow][HD + part * G + i]); vf[i] = to_f((T)qkv_s[arow][2 * HD + part * G + i]); }
    consume(kf, vf, live_row && half == 0 && slot == 0);
  }
  // combine the two waves of the row (and all key slots)
  float wm = wave_max(m);
  if (lane == 0) s_m[wave] = wm;
  __syncthreads();
  const float M = fmaxf(s_m[arow * 2], s_m[arow * 2 + 1]);
  const float f = (m == -INFINITY) ? 0.f : __expf(m - M);
  l *= f;
#pragma unroll
  for (int i = 0; i < G; ++i) oacc[i] *= f;
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int i = 0; i < G; ++i) oacc[i] += __shfl_xor(oacc[i], o, 64);
  }
  if (lane < LPK) {
#pragma unroll
    for (int i = 0; i < G; ++i) s_acc[wave][lane * G + i] = oacc[i];
    if (lane == 0) s_acc[wave][HD] = l;
  }
  __syncthreads();
  if (tid < RPG * HD) {
    const int r = tid / HD, e = tid - r * HD;
    const int b = rg * RPG + r;
    if (b < a.B && a.active[b]) {
      const float L = s_acc[2 * r][HD] + s_acc[2 * r + 1][HD];
      const float v = s_acc[2 * r][e] + s_acc[2 * r + 1][e];
      ((T*)a.out)[(long long)b * d + h * HD + e] = (T)(v / L);
    }
  }
}

// step tail: sample every row, update row state, emit the next step's input embedding
// (t2s_model.py:714-769).  grid = B, block = 64.
template <int NPL>
__global__ __launch_bounds__(64) void sample_step_kernel(const float* __restrict__ logits, int V, int EOS,
                                                         const StepParams* __restrict__ spp, int* __restrict__ ytok,
                                                         int ycap, int* __restrict__ kv_len, int* __restrict__ active,
                                                         int* __restrict__ step_ctr, int* __restrict__ n_active,
                                                         const float* __restrict__ e_audio, const float* __restrict__ pe,
                                                         float alpha_a, int d, float* __restrict__ ybuf) {
  extern __shared__ unsigned char seen[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const StepParams sp = *spp;
  const int step = step_ctr[b];
  if (!active[b]) return;
  const int Veff = (step < sp.eos_mask_steps) ? V - 1 : V;
  const int prev_len = sp.P + step;
  int* yrow = ytok + (long long)b * ycap;
  const float* nrow = nullptr;
  if (sp.noise)
    nrow = sp.noise + ((long long)step * sp.noise_rows + (sp.noise_rows > 1 ? b : 0)) * V;
  int smp, amx;
  sample_row<NPL>(logits + (long long)b * V, V, Veff, yrow, prev_len, sp.top_k, sp.top_p, sp.temperature,
                  sp.rep_penalty, nrow, sp.seed, b, step, seen, &smp, &amx);
  if (sp.dump)
    for (int v = lane; v < V; v += 64) sp.dump[((long long)step * gridDim.x + b) * V + v] = logits[(long long)b * V + v];
  if (sp.drawn && lane == 0) { int* dr = sp.drawn + ((long long)step * gridDim.x + b) * 2; dr[0] = smp; dr[1] = amx; }
  if (sp.force) { smp = sp.force[(long long)b * sp.max_steps + step]; amx = smp; }    // teacher forcing (parity hook)
  const bool fin = (smp == EOS) || (amx == EOS);
  const bool early = (sp.early_stop_num != -1 && (step + 1) > sp.early_stop_num) || (step >= sp.max_steps - 1);
  if (lane == 0) {
    if (prev_len < ycap) yrow[prev_len] = smp;
    if (fin || early) {
      active[b] = 0;
      sp.out_len[b] = step;
      atomicSub(n_active, 1);
    } else {
      sp.out_tokens[(long long)b * sp.max_steps + step] = smp;
      if (step > 0) kv_len[b] += 1;
    }
    step_ctr[b] = step + 1;
  }
  if (!(fin || early)) {
    const int tok = min(max(smp, 0), V - 1);
    const float* e = e_audio + (long long)tok * d;
    const float* p = pe + (long long)(sp.P + step) * d;
    for (int c = lane; c < d; c += 64) ybuf[(long long)b * d + c] = e[c] + alpha_a * p[c];
  }
}

template <int NPL>
__global__ __launch_bounds__(64) void sample_only_kernel(const float* __restrict__ logits, int V, int Veff,
                                                         const int* __restrict__ prev, int prev_len, int top_k, float top_p,
                                                         float temperature, float rp, const float* __restrict__ noise,
                                                         unsigned long long seed, int step, int* __restrict__ sampled,
                                                         int* __restrict__ argmax_tok) {
  extern __shared__ unsigned char seen[];
  const int b = blockIdx.x;
  int smp, amx;
  sample_row<NPL>(logits + (long long)b * V, V, Veff, prev + (long long)b * prev_len, prev_len, top_k, top_p, temperature,
                  rp, noise ? noise + (long long)b * V : nullptr, seed, b, step, seen, &smp, &amx);
  if (threadIdx.x == 0) { sampled[b] = smp; argmax_tok[b] = amx; }
}

}  // namespace gsv

// =======================================================================================
// engine
// =======================================================================================
using namespace gsv;

struct LayerW {
  void *qkv_w = nullptr, *out_w = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *qkv_b = nullptr, *out_b = nullptr, *b1 = nullptr, *b2 = nullptr;
  float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
};

struct gsv_t2s {
  gsv_t2s_config cfg;
  int dtype, max_batch, max_seq;
  bool finalized = false;
  std::map<std::string, std::vector<float>> staged;
  std::vector<LayerW> layers;
  void* bert_w = nullptr; float* bert_b = nullptr;
  float *e_text = nullptr, *e_audio = nullptr, *pe = nullptr;
  void* pred_w = nullptr;
  float alpha_t = 1.f, alpha_a = 1.f;
  int pe_rows = 0;
  // KV arena
  void* kv = nullptr; size_t kv_layer_stride = 0;  // elements per (layer, k|v)
  // row state
  int *d_x_len = nullptr, *d_row_off = nullptr, *d_ph_off = nullptr, *d_kv_len = nullptr, *d_active = nullptr,
      *d_step = nullptr, *d_n_active = nullptr, *d_ytok = nullptr;
  int ycap = 0;
  int* h_pinned = nullptr;
  StepParams* d_sp = nullptr;
  // decode buffers
  float *ybuf = nullptr, *xres = nullptr, *logits = nullptr;
  void *qbuf = nullptr, *abuf = nullptr, *hbuf = nullptr;
  // prefill workspace (grown on demand)
  size_t pf_rows = 0;
  void *pf_x = nullptr, *pf_qkv = nullptr, *pf_attn = nullptr, *pf_h = nullptr;
  void* pf_vt = nullptr; size_t pf_vt_cap = 0;   // V^T scratch of the MFMA prefill attention: [B][H][32][ceil32(maxS)] halfs
  float *pf_y = nullptr, *pf_bert = nullptr;
  void* pf_bert_t = nullptr;
  // current batch
  int B = 0, P = 0;
  int max_kv0 = 0;          // longest row's cached positions after prefill (host copy: bounds the decode budget)
  // persistent decode engine (t2s_mega.hip): fp16, v1/v2 shape, B <= 128; the launch-per-phase step stays as the
  // fp32 / other-shape path and behind GSV_T2S_NO_MEGA=1 for A/B
  MegaState mega;
  hipEvent_t mega_ev[2] = {nullptr, nullptr};
  float last_decode_ms = 0.f; int last_decode_steps = 0; int last_decode_mode = 0;
  bool mega_on = true;      // gsv_t2s_set_mega (A/B inside one process); GSV_T2S_NO_MEGA=1 never builds the engine
  const int* dbg_force = nullptr; float* dbg_dump = nullptr; int* dbg_drawn = nullptr; int dbg_stall = 0;   // gsv_t2s_set_debug / gsv_t2s_debug_stall: apply to the NEXT decode call only
  std::map<int, hipGraphExec_t> graphs;
  std::vector<void*> allocs;
};

namespace {

size_t esz(const gsv_t2s* h) { return dt_size(h->dtype); }

int dev_alloc(gsv_t2s* h, void** p, size_t bytes) {
  GSV_HIP(hipMalloc(p, bytes ? bytes : 16));
  h->allocs.push_back(*p);
  return GSV_OK;
}

int upload_f32(gsv_t2s* h, const std::vector<float>& v, float** out) {
  int rc = dev_alloc(h, (void**)out, v.size() * 4);
  if (rc) return rc;
  GSV_HIP(hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return GSV_OK;
}

int upload_t(gsv_t2s* h, const std::vector<float>& v, void** out) {
  if (h->dtype == GSV_F32) return upload_f32(h, v, (float**)out);
  std::vector<_Float16> tmp(v.size());
  for (size_t i = 0; i < v.size(); ++i) tmp[i] = (_Float16)v[i];
  int rc = dev_alloc(h, out, tmp.size() * 2);
  if (rc) return rc;
  GSV_HIP(hipMemcpy(*out, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  return GSV_OK;
}

const std::vector<float>* find(gsv_t2s* h, const std::string& k, size_t n) {
  auto it = h->staged.find(k);
  if (it == h->staged.end()) { set_error("t2s: missing tensor '%s'", k.c_str()); return nullptr; }
  if (it->second.size() != n) {
    set_error("t2s: tensor '%s' has %zu elements, expected %zu", k.c_str(), it->second.size(), n);
    return nullptr;
  }
  return &it->second;
}

template <typename T, int CB, int NW, int KSL, int KV4>
int launch_dec_gemm_inst(const DecGemmArgs& a, hipStream_t s) {
  constexpr int G = DT<T>::G;
  size_t lds = (size_t)NW * CB * 64 * 16;
  if (KV4 > 0) lds = std::max(lds, (size_t)CB * 16 * (a.K + G) * sizeof(T));
  auto kern = dec_gemm_kernel<T, CB, NW, KSL, KV4>;
  if (lds > 64 * 1024) GSV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(cdiv(a.N, 16), cdiv(a.B, CB * 16)), dim3(NW * 64), lds, s, a);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

// K = NW * KSL with KSL = 128 (64 for the 64-wide test model); LayerNorm prologue needs K = 32*KV4
template <typename T, int CB>
int launch_dec_gemm_cb(const DecGemmArgs& a, bool lnpro, hipStream_t s) {
  const int K = a.K;
#define GSV_DG_CASE(NWV, KSLV, KV4V) return launch_dec_gemm_inst<T, CB, NWV, KSLV, KV4V>(a, s)
  if (lnpro) {
    if (K == 512) GSV_DG_CASE(4, 128, 16);
    if (K == 256) GSV_DG_CASE(2, 128, 8);
    if (K == 128) GSV_DG_CASE(1, 128, 4);
    if (K == 64) GSV_DG_CASE(1, 64, 2);
    if (K == 1024) GSV_DG_CASE(8, 128, 32);
  } else {
    if (K == 2048) GSV_DG_CASE(16, 128, 0);
    if (K == 1024) GSV_DG_CASE(8, 128, 0);
    if (K == 512) GSV_DG_CASE(4, 128, 0);
    if (K == 256) GSV_DG_CASE(2, 128, 0);
    if (K == 128) GSV_DG_CASE(1, 128, 0);
    if (K == 64) GSV_DG_CASE(1, 64, 0);
    if (K == 4096) GSV_DG_CASE(16, 256, 0);
  }
#undef GSV_DG_CASE
  set_error("dec_gemm: unsupported contraction length K=%d (lnpro=%d)", K, (int)lnpro);
  return GSV_ERR_ARG;
}

template <typename T>
int launch_dec_gemm(const DecGemmArgs& a, bool lnpro, hipStream_t s) {
  // narrow outputs (N/16 < 64 workgroups, i.e. the N = 512 projections) are bound by what ONE CU can pull
  // (its weight tile plus the whole X operand): give every 16-row batch block its own workgroup so twice
  // as many CUs share the X traffic (the weight tile is then read once per block, from L2)
  if (!lnpro && a.N / 16 < 64 && a.B > 16) return launch_dec_gemm_cb<T, 1>(a, lnpro, s);
  const int cb = cdiv(a.B, 16);
  if (cb <= 1) return launch_dec_gemm_cb<T, 1>(a, lnpro, s);
  if (cb <= 2) return launch_dec_gemm_cb<T, 2>(a, lnpro, s);
  if (cb <= 4) return launch_dec_gemm_cb<T, 4>(a, lnpro, s);
  if (cb <= 8) return launch_dec_gemm_cb<T, 8>(a, lnpro, s);
  set_error("dec_gemm: batch %d too large", a.B);
  return GSV_ERR_ARG;
}

int sample_npl(int V) { return V <= 128 ? 2 : (V <= 1088 ? 17 : 32); }

}  // namespace

extern "C" {

int gsv_t2s_create(const gsv_t2s_config* cfg, int dtype, int max_batch, int max_seq, gsv_t2s_t** out) {
  GSV_REQUIRE(cfg && out, "t2s_create: null argument");
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "t2s_create: bad dtype");
  GSV_REQUIRE((cfg->dim == 64 || cfg->dim == 128 || cfg->dim == 256 || cfg->dim == 512 || cfg->dim == 1024) && cfg->ffn_dim == 4 * cfg->dim,
              "t2s_create: dim must be 64/128/256/512/1024 with ffn_dim = 4*dim (got %d, %d)", cfg->dim, cfg->ffn_dim);
  GSV_REQUIRE(cfg->dim / cfg->n_head == 32, "t2s_create: head_dim must be 32 (got %d)", cfg->dim / cfg->n_head);
  GSV_REQUIRE(cfg->vocab <= 2048, "t2s_create: vocab %d > 2048", cfg->vocab);
  GSV_REQUIRE(max_batch >= 1 && max_batch <= (dtype == GSV_F16 ? 128 : 64), "t2s_create: max_batch %d out of range", max_batch);
  int n = 0;
  GSV_HIP(hipGetDeviceCount(&n));
  gsv_t2s* h = new gsv_t2s();
  h->cfg = *cfg;
  h->dtype = dtype;
  h->max_batch = max_batch;
  h->max_seq = max_seq;
  h->layers.resize(cfg->n_layer);
  *out = h;
  return GSV_OK;
}

void gsv_t2s_destroy(gsv_t2s_t* h) {
  if (!h) return;
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.second);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->mega.h_err) (void)hipHostFree(h->mega.h_err);
  for (auto& e : h->mega_ev) if (e) (void)hipEventDestroy(e);
  delete h;
}

int gsv_t2s_load_tensor(gsv_t2s_t* h, const char* name, const float* data, int64_t numel) {
  GSV_REQUIRE(h && name && data && numel > 0, "t2s_load_tensor: bad argument");
  GSV_REQUIRE(!h->finalized, "t2s_load_tensor: handle already finalized");
  std::string k(name);
  if (k.rfind("model.", 0) == 0) k = k.substr(6);
  h->staged[k].assign(data, data + numel);
  return GSV_OK;
}

int gsv_t2s_finalize(gsv_t2s_t* h) {
  GSV_REQUIRE(h && !h->finalized, "t2s_finalize: bad handle");
  const auto& c = h->cfg;
  const size_t d = c.dim, ff = c.ffn_dim, V = c.vocab, PV = c.phoneme_vocab, BD = c.bert_dim;
  const std::vector<float>* t;
#define GSV_GET(key, n) if (!(t = find(h, key, n))) return GSV_ERR_ARG
  GSV_GET("bert_proj.weight", d * BD); GSV_RC(upload_t(h, *t, &h->bert_w));
  GSV_GET("bert_proj.bias", d); GSV_RC(upload_f32(h, *t, &h->bert_b));
  GSV_GET("ar_text_embedding.word_embeddings.weight", PV * d); GSV_RC(upload_f32(h, *t, &h->e_text));
  GSV_GET("ar_audio_embedding.word_embeddings.weight", V * d); GSV_RC(upload_f32(h, *t, &h->e_audio));
  GSV_GET("ar_text_position.alpha", 1); h->alpha_t = (*t)[0];
  GSV_GET("ar_audio_position.alpha", 1); h->alpha_a = (*t)[0];
  {
    auto it = h->staged.find("pe");
    GSV_REQUIRE(it != h->staged.end() && it->second.size() % d == 0, "t2s: missing sinusoid table 'pe' [n_pos][dim]");
    h->pe_rows = (int)(it->second.size() / d);
    GSV_REQUIRE(h->max_seq <= h->pe_rows, "t2s: max_seq %d exceeds the position table (%d rows)", h->max_seq, h->pe_rows);
    GSV_RC(upload_f32(h, it->second, &h->pe));
  }
  GSV_GET("ar_predict_layer.weight", V * d); GSV_RC(upload_t(h, *t, &h->pred_w));
  for (int i = 0; i < c.n_layer; ++i) {
    std::string p = "h.layers." + std::to_string(i) + ".";
    LayerW& L = h->layers[i];
    GSV_GET(p + "self_attn.in_proj_weight", 3 * d * d); GSV_RC(upload_t(h, *t, &L.qkv_w));
    GSV_GET(p + "self_attn.in_proj_bias", 3 * d); GSV_RC(upload_f32(h, *t, &L.qkv_b));
    GSV_GET(p + "self_attn.out_proj.weight", d * d); GSV_RC(upload_t(h, *t, &L.out_w));
    GSV_GET(p + "self_attn.out_proj.bias", d); GSV_RC(upload_f32(h, *t, &L.out_b));
    GSV_GET(p + "linear1.weight", ff * d); GSV_RC(upload_t(h, *t, &L.w1));
    GSV_GET(p + "linear1.bias", ff); GSV_RC(upload_f32(h, *t, &L.b1));
    GSV_GET(p + "linear2.weight", d * ff); GSV_RC(upload_t(h, *t, &L.w2));
    GSV_GET(p + "linear2.bias", d); GSV_RC(upload_f32(h, *t, &L.b2));
    GSV_GET(p + "norm1.weight", d); GSV_RC(upload_f32(h, *t, &L.n1w));
    GSV_GET(p + "norm1.bias", d); GSV_RC(upload_f32(h, *t, &L.n1b));
    GSV_GET(p + "norm2.weight", d); GSV_RC(upload_f32(h, *t, &L.n2w));
    GSV_GET(p + "norm2.bias", d); GSV_RC(upload_f32(h, *t, &L.n2b));
  }
#undef GSV_GET
  if (h->dtype == GSV_F16 && mega_shape_ok(c.dim, c.n_head, c.ffn_dim, c.vocab) && !getenv("GSV_T2S_NO_MEGA")) {
    // second copy of the decoder weights in the persistent engine's load order (every wave load = 1 KiB contiguous)
    MegaState& m = h->mega;
    const size_t lh = mega_layer_pack_halfs(), gh = mega_logits_pack_halfs();
    GSV_RC(dev_alloc(h, &m.wpack, (size_t)c.n_layer * lh * 2));
    GSV_RC(dev_alloc(h, &m.lpack, gh * 2));
    std::vector<_Float16> tmp(std::max(lh, gh));
    for (int i = 0; i < c.n_layer; ++i) {
      std::string p = "h.layers." + std::to_string(i) + ".";
      mega_pack_layer(h->staged[p + "self_attn.in_proj_weight"].data(), h->staged[p + "self_attn.out_proj.weight"].data(),
                      h->staged[p + "linear1.weight"].data(), h->staged[p + "linear2.weight"].data(), tmp.data());
      GSV_HIP(hipMemcpy((char*)m.wpack + (size_t)i * lh * 2, tmp.data(), lh * 2, hipMemcpyHostToDevice));
    }
    mega_pack_logits(h->staged["ar_predict_layer.weight"].data(), c.vocab, tmp.data());
    GSV_HIP(hipMemcpy(m.lpack, tmp.data(), gh * 2, hipMemcpyHostToDevice));
    // fp32 parameters of a layer side by side (no pointer chasing inside the kernel)
    GSV_RC(dev_alloc(h, (void**)&m.fpack, (size_t)c.n_layer * MEGA_FP_LAYER * 4));
    for (int i = 0; i < c.n_layer; ++i) {
      const LayerW& L = h->layers[i];
      float* dst = m.fpack + (size_t)i * MEGA_FP_LAYER;
      const float* srcs[8] = {L.qkv_b, L.out_b, L.b1, L.b2, L.n1w, L.n1b, L.n2w, L.n2b};
      const size_t ns[8] = {3 * d, d, ff, d, d, d, d, d};
      for (int k = 0; k < 8; ++k) {
        GSV_HIP(hipMemcpy(dst, srcs[k], ns[k] * 4, hipMemcpyDeviceToDevice));
        dst += ns[k];
      }
    }
    m.ring = 1;                                            // one hop buffer set (t2s_mega.hip hop_slot)
    m.hop_bytes = mega_hop_bytes(m.ring);
    GSV_RC(dev_alloc(h, (void**)&m.hop, m.hop_bytes));
    GSV_RC(dev_alloc(h, (void**)&m.err, 64));
    GSV_RC(dev_alloc(h, (void**)&m.snap, ((size_t)4 * h->max_batch + 4) * 4));
    GSV_HIP(hipHostMalloc((void**)&m.h_err, 64));
    GSV_HIP(hipEventCreate(&h->mega_ev[0]));
    GSV_HIP(hipEventCreate(&h->mega_ev[1]));
    m.ready = true;
  }
  h->staged.clear();
  const size_t B = h->max_batch, es = esz(h);
  h->kv_layer_stride = B * d * (size_t)h->max_seq;
  GSV_RC(dev_alloc(h, &h->kv, (size_t)c.n_layer * 2 * h->kv_layer_stride * es));
  GSV_RC(dev_alloc(h, (void**)&h->d_x_len, B * 4));
  GSV_RC(dev_alloc(h, (void**)&h->d_row_off, B * 4));
  GSV_RC(dev_alloc(h, (void**)&h->d_ph_off, B * 4));
  // one block [kv_len | active | step | n_active]: the persistent engine's fallback saves and restores it with one copy
  GSV_RC(dev_alloc(h, (void**)&h->d_kv_len, (3 * B + 4) * 4));
  h->d_active = h->d_kv_len + B; h->d_step = h->d_kv_len + 2 * B; h->d_n_active = h->d_kv_len + 3 * B;
  h->ycap = h->max_seq + 8;
  GSV_RC(dev_alloc(h, (void**)&h->d_ytok, B * h->ycap * 4));
  GSV_RC(dev_alloc(h, (void**)&h->d_sp, sizeof(StepParams)));
  GSV_HIP(hipHostMalloc((void**)&h->h_pinned, 64));
  GSV_RC(dev_alloc(h, (void**)&h->ybuf, B * d * 4));
  GSV_RC(dev_alloc(h, (void**)&h->xres, B * d * 4));
  GSV_RC(dev_alloc(h, (void**)&h->logits, B * V * 4));
  GSV_RC(dev_alloc(h, &h->qbuf, B * d * es));
  GSV_RC(dev_alloc(h, &h->abuf, B * d * es));
  GSV_RC(dev_alloc(h, &h->hbuf, B * ff * es));
  GSV_HIP(hipMemset(h->logits, 0, B * V * 4));
  h->finalized = true;
  return GSV_OK;
}

}  // extern "C"

static void* kv_ptr(gsv_t2s* h, int layer, int which) {
  return (char*)h->kv + ((size_t)(layer * 2 + which) * h->kv_layer_stride) * esz(h);
}

// logits (LN2 prologue of the last layer) + sampling/state update
static int launch_tail(gsv_t2s* h, hipStream_t s) {
  const auto& c = h->cfg;
  DecGemmArgs a;
  memset(&a, 0, sizeof(a));
  const LayerW& L = h->layers[c.n_layer - 1];
  a.yin = h->ybuf; a.gamma = L.n2w; a.beta = L.n2b; a.w = h->pred_w; a.bias = nullptr;
  a.B = h->B; a.K = c.dim; a.N = c.vocab; a.epi = EPI_LOGITS; a.out_f = h->logits;
  int rc = h->dtype == GSV_F16 ? launch_dec_gemm<_Float16>(a, true, s) : launch_dec_gemm<float>(a, true, s);
  if (rc) return rc;
  const int V = c.vocab, EOS = c.vocab - 1;
  const int npl = sample_npl(V);
#define GSV_SAMPLE(N)                                                                                          \
  hipLaunchKernelGGL(sample_step_kernel<N>, dim3(h->B), dim3(64), (size_t)((V + 15) & ~15), s, h->logits, V, EOS, h->d_sp, \
                     h->d_ytok, h->ycap, h->d_kv_len, h->d_active, h->d_step, h->d_n_active, h->e_audio, h->pe,  \
                     h->alpha_a, c.dim, h->ybuf)
  if (npl == 2) GSV_SAMPLE(2);
  else if (npl == 17) GSV_SAMPLE(17);
  else GSV_SAMPLE(32);
#undef GSV_SAMPLE
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

template <typename T, int KD>
static bool launch_qkv_attn_kd(const QkvAttnArgs& q, hipStream_t s) {
  if constexpr ((KD / Frag16<T>::KS) % 4 == 0) {
    hipLaunchKernelGGL((dec_qkv_attn_kernel<T, 32, KD>), dim3(q.H, cdiv(q.B, 2)), dim3(256), 0, s, q);
    return true;
  } else {
    return false;   // contraction too short to split over 4 waves: the caller uses the unfused kernels
  }
}

template <typename T>
static bool launch_qkv_attn(const QkvAttnArgs& q, hipStream_t s) {
  switch (q.d) {
    case 64: return launch_qkv_attn_kd<T, 64>(q, s);
    case 128: return launch_qkv_attn_kd<T, 128>(q, s);
    case 256: return launch_qkv_attn_kd<T, 256>(q, s);
    case 512: return launch_qkv_attn_kd<T, 512>(q, s);
    case 1024: return launch_qkv_attn_kd<T, 1024>(q, s);
    default: return false;
  }
}

template <typename T>
static int launch_decode_layers(gsv_t2s* h, hipStream_t s, int only_attn, hipEvent_t* attn_ev = nullptr) {
  const auto& c = h->cfg;
  const int d = c.dim, H = c.n_head;
  for (int li = 0; li < c.n_layer; ++li) {
    const LayerW& L = h->layers[li];
    // The fused LN+QKV+append+attention kernel is OPT-IN: measured on MI355X at B=32 it is slower than the
    // two separate launches (153.3 vs 148.1 ms per bench step): one workgroup per CU has to pull 96 KB of
    // weights plus 72 KB of K/V through a single CU's memory path and runs LN -> MFMA -> reduce -> softmax
    // as one serial chain, whereas the split kernels spread the same bytes over 96 + 512 workgroups.
    static const bool use_fuse = getenv("GSV_FUSED_QKV_ATTN") != nullptr;
    bool fused = false;
    if (!only_attn && use_fuse) {
      QkvAttnArgs q;
      memset(&q, 0, sizeof(q));
      q.yin = h->ybuf;
      if (li > 0) { q.gamma = h->layers[li - 1].n2w; q.beta = h->layers[li - 1].n2b; }
      q.xres_out = h->xres; q.w = L.qkv_w; q.bias = L.qkv_b;
      q.kc = kv_ptr(h, li, 0); q.vc = kv_ptr(h, li, 1); q.kv_len = h->d_kv_len; q.active = h->d_active;
      q.out = h->abuf; q.B = h->B; q.d = d; q.H = H; q.smax = h->max_seq;
      fused = launch_qkv_attn<T>(q, s);
    }
    if (fused) {
    } else {
    if (!only_attn) {
      DecGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.yin = h->ybuf;
      if (li > 0) { a.gamma = h->layers[li - 1].n2w; a.beta = h->layers[li - 1].n2b; }
      a.xres_out = h->xres;
      a.w = L.qkv_w; a.bias = L.qkv_b; a.B = h->B; a.K = d; a.N = 3 * d; a.epi = EPI_QKV;
      a.out_t = h->qbuf; a.kc = kv_ptr(h, li, 0); a.vc = kv_ptr(h, li, 1); a.kv_len = h->d_kv_len; a.active = h->d_active;
      a.d = d; a.H = H; a.smax = h->max_seq;
      GSV_RC(launch_dec_gemm<T>(a, true, s));
    }
    if (attn_ev) GSV_HIP(hipEventRecord(attn_ev[2 * li], s));
    hipLaunchKernelGGL((decode_attn_kernel<T, 32>), dim3(H, h->B), dim3(256), 0, s, (const T*)h->qbuf,
                       (const T*)kv_ptr(h, li, 0), (const T*)kv_ptr(h, li, 1), h->d_kv_len, h->d_active, H, h->max_seq,
                       (T*)h->abuf);
    if (attn_ev) GSV_HIP(hipEventRecord(attn_ev[2 * li + 1], s));
    }
    if (only_attn) continue;
    {
      DecGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.xin = h->abuf; a.w = L.out_w; a.bias = L.out_b; a.B = h->B; a.K = d; a.N = d; a.epi = EPI_RESID;
      a.out_f = h->ybuf; a.xres = h->xres;
      GSV_RC(launch_dec_gemm<T>(a, false, s));
    }
    {
      DecGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.yin = h->ybuf; a.gamma = L.n1w; a.beta = L.n1b; a.xres_out = h->xres;
      a.w = L.w1; a.bias = L.b1; a.B = h->B; a.K = d; a.N = c.ffn_dim; a.epi = EPI_RELU; a.out_t = h->hbuf;
      GSV_RC(launch_dec_gemm<T>(a, true, s));
    }
    {
      DecGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.xin = h->hbuf; a.w = L.w2; a.bias = L.b2; a.B = h->B; a.K = c.ffn_dim; a.N = d; a.epi = EPI_RESID;
      a.out_f = h->ybuf; a.xres = h->xres;
      GSV_RC(launch_dec_gemm<T>(a, false, s));
    }
  }
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

static int launch_step(gsv_t2s* h, hipStream_t s) {
  int rc = h->dtype == GSV_F16 ? launch_decode_layers<_Float16>(h, s, 0) : launch_decode_layers<float>(h, s, 0);
  if (rc) return rc;
  return launch_tail(h, s);
}

static int grow_prefill(gsv_t2s* h, size_t rows, size_t xrows) {
  if (rows <= h->pf_rows) return GSV_OK;
  const size_t d = h->cfg.dim, ff = h->cfg.ffn_dim, es = esz(h);
  rows = (rows + 255) & ~(size_t)255;
  // old buffers stay registered in allocs and are released at destroy; growth is rare
  GSV_RC(dev_alloc(h, &h->pf_x, rows * d * es));
  GSV_RC(dev_alloc(h, &h->pf_qkv, rows * 3 * d * es));
  GSV_RC(dev_alloc(h, &h->pf_attn, rows * d * es));
  GSV_RC(dev_alloc(h, &h->pf_h, rows * ff * es));
  GSV_RC(dev_alloc(h, (void**)&h->pf_y, rows * d * 4));
  GSV_RC(dev_alloc(h, (void**)&h->pf_bert, rows * d * 4));
  GSV_RC(dev_alloc(h, &h->pf_bert_t, rows * (size_t)h->cfg.bert_dim * es));
  h->pf_rows = rows;
  (void)xrows;
  return GSV_OK;
}

extern "C" {

int gsv_t2s_prefill(gsv_t2s_t* h, const int32_t* phones, const int32_t* phone_lens, int B, const float* bert,
                    const int32_t* prompts, int P, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized, "t2s_prefill: handle not finalized");
  GSV_REQUIRE(phones && phone_lens && (prompts || P == 0), "t2s_prefill: null argument");
  GSV_REQUIRE(B >= 1 && B <= h->max_batch, "t2s_prefill: batch %d exceeds max_batch %d", B, h->max_batch);
  GSV_REQUIRE(P >= 0, "t2s_prefill: negative prompt length %d", P);   // P == 0: prompt-free decode (t2s_model.py:849-856)
  hipStream_t s = (hipStream_t)stream;
  const auto& c = h->cfg;
  const int d = c.dim, H = c.n_head;
  std::vector<int> row_off(B), ph_off(B), kvl(B), ones(B, 1), zeros(B, 0);
  int M = 0, SX = 0, maxS = 0;
  for (int b = 0; b < B; ++b) {
    GSV_REQUIRE(phone_lens[b] >= 1, "t2s_prefill: empty phoneme sequence in row %d", b);
    row_off[b] = M; ph_off[b] = SX;
    const int S = phone_lens[b] + P;
    GSV_REQUIRE(S + 2 <= h->max_seq, "t2s_prefill: row %d needs %d positions, max_seq is %d", b, S + 2, h->max_seq);
    GSV_REQUIRE(phone_lens[b] <= h->pe_rows && P <= h->pe_rows, "t2s_prefill: sequence exceeds the position table");
    kvl[b] = S; M += S; SX += phone_lens[b];
    maxS = S > maxS ? S : maxS;
  }
  GSV_RC(grow_prefill(h, M, SX));
  if (h->dtype == GSV_F16) {
    const size_t need_vt = (size_t)B * H * 32 * ((maxS + 31) / 32 * 32) * 2;
    if (need_vt > h->pf_vt_cap) { GSV_RC(dev_alloc(h, &h->pf_vt, need_vt + need_vt / 4)); h->pf_vt_cap = need_vt + need_vt / 4; }
  }
  h->B = B; h->P = P; h->max_kv0 = maxS;
  GSV_HIP(hipMemcpyAsync(h->d_x_len, phone_lens, B * 4, hipMemcpyHostToDevice, s));
  GSV_HIP(hipMemcpyAsync(h->d_row_off, row_off.data(), B * 4, hipMemcpyHostToDevice, s));
  GSV_HIP(hipMemcpyAsync(h->d_ph_off, ph_off.data(), B * 4, hipMemcpyHostToDevice, s));
  {
    // row state [kv_len | active | step | n_active] is one block: one upload
    const size_t mb = (size_t)h->max_batch;
    std::vector<int> st(3 * mb + 4, 0);
    for (int b = 0; b < B; ++b) { st[b] = kvl[b]; st[mb + b] = 1; }
    st[3 * mb] = B;
    GSV_HIP(hipMemcpyAsync(h->d_kv_len, st.data(), st.size() * 4, hipMemcpyHostToDevice, s));
    GSV_HIP(hipStreamSynchronize(s));
  }
  if (P > 0)
    GSV_HIP(hipMemcpy2DAsync(h->d_ytok, (size_t)h->ycap * 4, prompts, (size_t)P * 4, (size_t)P * 4, B, hipMemcpyDeviceToDevice, s));
  GSV_HIP(hipStreamSynchronize(s));  // host vectors above go out of scope

  const float* bertp = nullptr;
  if (bert) {
    GSV_RC(launch_convert(bert, h->pf_bert_t, h->dtype, (long long)SX * c.bert_dim, s));
    ConvArgs g;
    g.x = h->pf_bert_t; g.w = h->bert_w; g.bias = h->bert_b; g.y = h->pf_bert; g.out_f32 = 1;
    g.T_in = SX; g.T_out = SX; g.T_virt = SX; g.Cin = c.bert_dim; g.Cout = d; g.ldx = c.bert_dim; g.ldw = c.bert_dim; g.ldy = d;
    GSV_RC(launch_conv_gemm(h->dtype, g, s));
    bertp = h->pf_bert;
  }
#define GSV_EMBED(T)                                                                                              \
  hipLaunchKernelGGL(embed_prefill_kernel<T>, dim3(maxS, B), dim3(128), 0, s, phones, prompts, h->d_row_off, h->d_ph_off, \
                     h->d_x_len, h->e_text, h->e_audio, bertp, h->bert_b, h->pe, h->alpha_t, h->alpha_a, P, d, (T*)h->pf_x)
  if (h->dtype == GSV_F16) GSV_EMBED(_Float16); else GSV_EMBED(float);
#undef GSV_EMBED
  GSV_HIP(hipGetLastError());

  static const bool scalar_pf = getenv("GSV_SCALAR_PREFILL_ATTN") != nullptr;   // A/B switch: thread-per-query VALU kernel
  static const bool split_scatter = getenv("GSV_PREFILL_SPLIT_SCATTER") != nullptr;   // A/B switch: K/V scatter and V^T as two launches
  for (int li = 0; li < c.n_layer; ++li) {
    const LayerW& L = h->layers[li];
    ConvArgs g;
    g.x = h->pf_x; g.w = L.qkv_w; g.bias = L.qkv_b; g.y = h->pf_qkv;
    g.T_in = M; g.T_out = M; g.T_virt = M; g.Cin = d; g.Cout = 3 * d; g.ldx = d; g.ldw = d; g.ldy = 3 * d;
    GSV_RC(launch_conv_gemm(h->dtype, g, s));
    if (h->dtype == GSV_F16) {
      const bool fused_kvt = d / H == 32 && !scalar_pf && !split_scatter;
      if (!fused_kvt)
        hipLaunchKernelGGL(kv_scatter_kernel<_Float16>, dim3(maxS, B), dim3(128), 0, s, (const _Float16*)h->pf_qkv,
                           h->d_row_off, h->d_x_len, P, d, H, h->max_seq, (_Float16*)kv_ptr(h, li, 0), (_Float16*)kv_ptr(h, li, 1));
      if (d / H == 32 && !scalar_pf) {
        const int spad = (maxS + 31) / 32 * 32;
        if (fused_kvt)
          hipLaunchKernelGGL(prefill_kvt_kernel, dim3(spad / 32, H, B), dim3(256), 0, s, (const _Float16*)h->pf_qkv, h->d_row_off,
                             h->d_x_len, P, d, H, h->max_seq, spad, (_Float16*)kv_ptr(h, li, 0), (_Float16*)kv_ptr(h, li, 1),
                             (_Float16*)h->pf_vt);
        else
        hipLaunchKernelGGL(prefill_vt_kernel, dim3(spad / 32, H, B), dim3(256), 0, s, (const _Float16*)h->pf_qkv, h->d_row_off,
                           h->d_x_len, P, d, H, spad, (_Float16*)h->pf_vt);
        hipLaunchKernelGGL(prefill_flash32_f16_kernel<4>, dim3(cdiv(maxS, 64), H, B), dim3(256), 0, s, (const _Float16*)h->pf_qkv,
                           (const _Float16*)kv_ptr(h, li, 0), (const _Float16*)h->pf_vt, h->d_row_off, h->d_x_len, P, d, H,
                           h->max_seq, spad, (_Float16*)h->pf_attn);
      } else
      hipLaunchKernelGGL((prefill_attn_kernel<_Float16, 32>), dim3(cdiv(maxS, 64), H, B), dim3(64), 0, s,
                         (const _Float16*)h->pf_qkv, (const _Float16*)kv_ptr(h, li, 0), (const _Float16*)kv_ptr(h, li, 1),
                         h->d_row_off, h->d_x_len, P, d, H, h->max_seq, (_Float16*)h->pf_attn);
    } else {
      hipLaunchKernelGGL(kv_scatter_kernel<float>, dim3(maxS, B), dim3(128), 0, s, (const float*)h->pf_qkv, h->d_row_off,
                         h->d_x_len, P, d, H, h->max_seq, (float*)kv_ptr(h, li, 0), (float*)kv_ptr(h, li, 1));
      hipLaunchKernelGGL((prefill_attn_kernel<float, 32>), dim3(cdiv(maxS, 64), H, B), dim3(64), 0, s,
                         (const float*)h->pf_qkv, (const float*)kv_ptr(h, li, 0), (const float*)kv_ptr(h, li, 1),
                         h->d_row_off, h->d_x_len, P, d, H, h->max_seq, (float*)h->pf_attn);
    }
    GSV_HIP(hipGetLastError());
    // y1 = attn Wo^T + bo + x  (fp32) ; x1 = LN1(y1)
    ConvArgs o;
    o.x = h->pf_attn; o.w = L.out_w; o.bias = L.out_b; o.y = h->pf_y; o.out_f32 = 1; o.res = h->pf_x; o.res_f32 = 0;
    o.T_in = M; o.T_out = M; o.T_virt = M; o.Cin = d; o.Cout = d; o.ldx = d; o.ldw = d; o.ldy = d; o.ldr = d;
    GSV_RC(launch_conv_gemm(h->dtype, o, s));
    GSV_RC(launch_layernorm(h->dtype, h->pf_y, 1, nullptr, 0, L.n1w, L.n1b, h->pf_x, 0, M, d, 1e-5f, s));
    ConvArgs f1;
    f1.x = h->pf_x; f1.w = L.w1; f1.bias = L.b1; f1.y = h->pf_h; f1.post_act = ACT_RELU;
    f1.T_in = M; f1.T_out = M; f1.T_virt = M; f1.Cin = d; f1.Cout = c.ffn_dim; f1.ldx = d; f1.ldw = d; f1.ldy = c.ffn_dim;
    GSV_RC(launch_conv_gemm(h->dtype, f1, s));
    ConvArgs f2;
    f2.x = h->pf_h; f2.w = L.w2; f2.bias = L.b2; f2.y = h->pf_y; f2.out_f32 = 1; f2.res = h->pf_x; f2.res_f32 = 0;
    f2.T_in = M; f2.T_out = M; f2.T_virt = M; f2.Cin = c.ffn_dim; f2.Cout = d; f2.ldx = c.ffn_dim; f2.ldw = c.ffn_dim;
    f2.ldy = d; f2.ldr = d;
    GSV_RC(launch_conv_gemm(h->dtype, f2, s));
    if (li + 1 < c.n_layer)
      GSV_RC(launch_layernorm(h->dtype, h->pf_y, 1, nullptr, 0, L.n2w, L.n2b, h->pf_x, 0, M, d, 1e-5f, s));
  }
  hipLaunchKernelGGL(gather_last_kernel, dim3(B), dim3(128), 0, s, h->pf_y, h->d_row_off, h->d_x_len, P, d, h->ybuf);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_t2s_decode(gsv_t2s_t* h, const gsv_sampling_params* sp, const float* noise, int noise_rows,
                   int32_t* out_tokens, int32_t* out_len, int* steps_run, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized && h->B > 0, "t2s_decode: call gsv_t2s_prefill first");
  GSV_REQUIRE(sp && out_tokens && out_len, "t2s_decode: null argument");
  GSV_REQUIRE(sp->max_steps >= 1, "t2s_decode: max_steps must be >= 1");
  GSV_REQUIRE(noise == nullptr || noise_rows == 1 || noise_rows == h->B, "t2s_decode: noise_rows must be 1 or B");
  hipStream_t s = (hipStream_t)stream;
  StepParams p;
  p.top_k = sp->top_k; p.top_p = sp->top_p; p.temperature = sp->temperature; p.rep_penalty = sp->repetition_penalty;
  p.early_stop_num = sp->early_stop_num; p.eos_mask_steps = sp->eos_mask_steps; p.max_steps = sp->max_steps;
  p.noise_rows = noise ? noise_rows : 0; p.seed = sp->seed; p.noise = noise; p.out_tokens = out_tokens; p.out_len = out_len;
  p.P = h->P;
  p.force = h->dbg_force; p.dump = h->dbg_dump; p.drawn = h->dbg_drawn;
  h->dbg_force = nullptr; h->dbg_dump = nullptr; h->dbg_drawn = nullptr;
  GSV_HIP(hipMemcpyAsync(h->d_sp, &p, sizeof(p), hipMemcpyHostToDevice, s));
  GSV_HIP(hipStreamSynchronize(s));
  // budget: step 0 samples from the prefill's last position, every later step appends one K/V position, so the
  // longest row ends at max_kv0 + budget - 1 cached positions; the sampling tail reads pe[P + step] and writes
  // token history [P + step].  A request that does not fit is refused here: the kernels clamp out-of-range
  // appends, which would otherwise yield silently wrong tokens with rc 0.
  int budget = sp->max_steps;
  if (sp->early_stop_num >= 0 && sp->early_stop_num + 1 < budget) budget = sp->early_stop_num + 1;
  GSV_REQUIRE(h->max_kv0 + budget <= h->max_seq,
              "t2s_decode: %d cached positions + %d steps exceed the K/V arena (max_seq %d); lower max_steps / early_stop_num "
              "or create the engine with a larger max_seq", h->max_kv0, budget, h->max_seq);
  GSV_REQUIRE(h->P + budget <= h->pe_rows, "t2s_decode: prompt %d + %d steps exceed the position table (%d rows)", h->P, budget,
              h->pe_rows);
  // step 0: logits of the last prefill position, sample, emit first embedding
  GSV_RC(launch_tail(h, s));
  int steps = 1;
  h->last_decode_mode = 0; h->last_decode_ms = 0.f; h->last_decode_steps = 0;
  if (h->mega.ready && h->mega_on && h->B <= MEGA_MAX_B && budget > 1) {
    MegaState& m = h->mega;
    if (m.census < 0) {
      // once per handle: are the engine's 256 workgroups co-resident on this device?  If not, a hand-off could wait
      // for a workgroup that never starts: the launch-per-phase step is used instead (gsv_t2s_decode_info reports it)
      const int rc = mega_census(s, m.err, m.h_err);
      if (rc < 0) return rc;
      m.census = rc;
    }
    if (m.census == 1) {
      static const bool map_local = getenv("GSV_MEGA_GROUP_XCD") != nullptr;
      GSV_HIP(hipMemsetAsync(m.hop, 0, m.hop_bytes, s));      // no tag survives a call (epochs are unique per launch as well: ep_base)
      GSV_HIP(hipMemsetAsync(m.err, 0, 64, s));
      // the row state as step 0 left it: if the launch ends in a hand-off timeout the batch is re-run from here on the
      // launch-per-phase path (the engine only appends K/V behind kv_len and token history behind P + step: restoring the
      // counters makes both invisible again; ybuf, the first step's input, is read-only for the engine)
      const size_t mb = (size_t)h->max_batch;
      GSV_HIP(hipMemcpyAsync(m.snap, h->d_kv_len, (3 * mb + 4) * 4, hipMemcpyDeviceToDevice, s));
      GSV_HIP(hipMemcpyAsync(m.snap + 3 * mb + 4, out_len, h->B * 4, hipMemcpyDeviceToDevice, s));
      MegaArgs a;
      memset(&a, 0, sizeof(a));
      a.wpack = (const h8*)m.wpack; a.lpack = (const h8*)m.lpack; a.fpack = m.fpack;
      a.kv = (_Float16*)h->kv; a.kv_layer_stride = h->kv_layer_stride; a.smax = h->max_seq;
      a.kv_len = h->d_kv_len; a.active = h->d_active; a.step_ctr = h->d_step; a.n_active = h->d_n_active;
      a.ytok = h->d_ytok; a.ycap = h->ycap; a.sp = h->d_sp; a.e_audio = h->e_audio; a.pe = h->pe; a.alpha_a = h->alpha_a;
      a.ybuf = h->ybuf; a.logits_out = h->logits; a.hop = m.hop; a.err = m.err; a.B = h->B; a.L = h->cfg.n_layer; a.V = h->cfg.vocab;
      static const int map_mode = getenv("GSV_MEGA_MAP") ? atoi(getenv("GSV_MEGA_MAP")) : 2;   // 2: roles by the XCD a workgroup runs on, 1: by blockIdx % 8
      a.nsteps = budget - 1; a.map_shared = map_local ? 0 : map_mode;
      // bits 0-3: hops (A, B, C, D) that poll one granule per line first; bits 8-12: 16ths of the lines that may still be
      // missing when the full passes start
      // bits 0-3: hops A-D poll ONE hint line before the full pass (hop B, 2 KB per row, is faster polled in full: 288 -> 284 us
      // per step); bit 7: every member polls ANOTHER publisher's line instead of all 32 polling the row's last line (297 -> 288
      // us); bits 8-12: miss threshold of sweep2's per-line hints (logits hop).  Measured and left off: bit 4 (sweep2: two polls
      // in flight, 3 % slower), bit 5 (payload through L2, needs GSV_MEGA_RING > 1: no gain).  Tried in sweep_wide and removed
      // again: several hint lines per row, polls in flight, one polling wave per workgroup, a slower pace, a wait before the
      // first poll (+1 to +9 %, profiles/r03_ab_hint_spread.txt) -- the knobs themselves cost 2 % in scalar registers
      static const int hint_mask = getenv("GSV_MEGA_HINT") ? atoi(getenv("GSV_MEGA_HINT")) : (13 | (1 << 7) | (2 << 8));
      a.hint_mask = hint_mask;
      a.ring = m.ring;
      m.launch_gen = (m.launch_gen + 1) & 2047;
      a.ep_base = m.launch_gen << 20;                       // 1500 steps x 98 hops < 2^20
      a.test_stall = h->dbg_stall; h->dbg_stall = 0;        // tests only (gsv_t2s_debug_stall): this launch loses one publish
      // measurement runs: GSV_MEGA_PROF=<file> dumps in-kernel shader-clock stamps of one (step, layer) for every wave
      const char* prof_path = getenv("GSV_MEGA_PROF");
      unsigned long long* d_prof = nullptr;
      const size_t prof_n = (size_t)256 * 8 * 32;
      if (prof_path && budget > 8) {
        GSV_HIP(hipMalloc((void**)&d_prof, prof_n * 8));
        GSV_HIP(hipMemsetAsync(d_prof, 0, prof_n * 8, s));
        a.prof = d_prof; a.prof_step = getenv("GSV_MEGA_PROF_STEP") ? atoi(getenv("GSV_MEGA_PROF_STEP")) : 5;
        a.prof_layer = getenv("GSV_MEGA_PROF_LAYER") ? atoi(getenv("GSV_MEGA_PROF_LAYER")) : 7;
        a.prof_quad = getenv("GSV_MEGA_PROF_QUAD") ? atoi(getenv("GSV_MEGA_PROF_QUAD")) : 0;
      }
      GSV_HIP(hipEventRecord(h->mega_ev[0], s));
      GSV_RC(launch_t2s_mega(a, s));
      GSV_HIP(hipEventRecord(h->mega_ev[1], s));
      GSV_HIP(hipMemcpyAsync(m.h_err, m.err, 16, hipMemcpyDeviceToHost, s));
      GSV_HIP(hipMemcpyAsync(h->h_pinned, h->d_step, 4, hipMemcpyDeviceToHost, s));
      GSV_HIP(hipStreamSynchronize(s));
      bool engine_ok = true;
      if (m.h_err[0] != 0u) {
        // A hand-off timed out (a member was not running: another kernel held its CU, or a fault).  The request is not
        // failed: the row state is restored to what step 0 left, the handle stops using the engine (census = 0: later calls
        // take the launch-per-phase step, gsv_t2s_engine_stats reports it) and THIS batch is re-run below on that path.
        // GSV_MEGA_STRICT=1 restores the old behaviour (GSV_ERR_STATE) for tests of the error path.
        set_error("t2s_decode: persistent engine hand-off timed out (epoch %u, workgroup %u, hop code 0x%x); the handle now uses "
                  "the launch-per-phase step (GSV_T2S_NO_MEGA=1 selects it from the start)", m.h_err[1], m.h_err[2], m.h_err[3]);
        m.fallbacks += 1;
        m.census = m.fallbacks >= 3 ? 0 : -1;     // a transient cause (a foreign kernel held a CU): census again at the next call; three strikes disable the engine
        m.last_err[0] = m.h_err[1]; m.last_err[1] = m.h_err[2]; m.last_err[2] = m.h_err[3];
        if (d_prof) (void)hipFree(d_prof);
        if (getenv("GSV_MEGA_STRICT")) return GSV_ERR_STATE;
        GSV_HIP(hipMemcpyAsync(h->d_kv_len, m.snap, (3 * mb + 4) * 4, hipMemcpyDeviceToDevice, s));
        GSV_HIP(hipMemcpyAsync(out_len, m.snap + 3 * mb + 4, h->B * 4, hipMemcpyDeviceToDevice, s));
        engine_ok = false;
      }
      if (engine_ok) {
      (void)hipEventElapsedTime(&h->last_decode_ms, h->mega_ev[0], h->mega_ev[1]);
      if (d_prof) {
        std::vector<unsigned long long> hp(prof_n);
        GSV_HIP(hipMemcpy(hp.data(), d_prof, prof_n * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_prof);
        if (FILE* f = fopen(prof_path, "w")) {
          fprintf(f, "# decode %.3f ms for %d steps; stamps of step %d layer %d: wg wave stamp0 then deltas to stamp0\n",
                  h->last_decode_ms, budget - 1, a.prof_step, a.prof_layer);
          for (int wg = 0; wg < 256; ++wg)
            for (int w = 0; w < 8; ++w) {
              const unsigned long long* p = &hp[((size_t)wg * 8 + w) * 32];
              if (!p[0]) continue;
              fprintf(f, "%d %d %llu", wg, w, p[0]);
              for (int i = 1; i < 24; ++i) fprintf(f, " %lld", p[i] ? (long long)(p[i] - p[0]) : -1ll);
              fprintf(f, "\n");
            }
          fclose(f);
        }
      }
      // every row ends by the budget's last step (early == step >= max_steps - 1); row 0's step counter tells how
      // far the longest-running group got only for its own group, so report the budget like the launch loop does
      h->last_decode_mode = 1; h->last_decode_steps = budget - 1;
      if (steps_run) *steps_run = budget;
      return GSV_OK;
      }   // engine_ok
    }
  }
  // steps >= 1 : captured once per batch size, replayed
  hipGraphExec_t exec = nullptr;
  auto it = h->graphs.find(h->B);
  if (it != h->graphs.end()) exec = it->second;
  else if (s != nullptr && !getenv("GSV_T2S_NO_GRAPH") && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
    hipGraph_t graph;
    int rc = launch_step(h, s);
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc) return rc;
    GSV_HIP(e);
    GSV_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    h->graphs[h->B] = exec;
  } else {
    (void)hipGetLastError();  // legacy default stream cannot capture: eager launches instead
  }
  const int check_every = 8;
  while (steps < budget) {
    if (exec) GSV_HIP(hipGraphLaunch(exec, s));
    else GSV_RC(launch_step(h, s));
    ++steps;
    if (steps % check_every == 0 || steps == budget) {
      GSV_HIP(hipMemcpyAsync(h->h_pinned, h->d_n_active, 4, hipMemcpyDeviceToHost, s));
      GSV_HIP(hipStreamSynchronize(s));
      if (h->h_pinned[0] <= 0) break;
    }
  }
  GSV_HIP(hipStreamSynchronize(s));
  if (steps_run) *steps_run = steps;
  return GSV_OK;
}

int gsv_t2s_set_mega(gsv_t2s_t* h, int on) {
  GSV_REQUIRE(h, "t2s_set_mega: null handle");
  h->mega_on = on != 0;
  return GSV_OK;
}

int gsv_t2s_set_debug(gsv_t2s_t* h, const int32_t* force_tokens, float* logits_dump, int32_t* drawn_dump) {
  GSV_REQUIRE(h && h->finalized, "t2s_set_debug: handle not finalized");
  h->dbg_force = force_tokens; h->dbg_dump = logits_dump; h->dbg_drawn = drawn_dump;
  return GSV_OK;
}

int gsv_t2s_debug_stall(gsv_t2s_t* h, int member) {
  GSV_REQUIRE(h && h->finalized && member >= 0 && member < 32, "t2s_debug_stall: bad argument");
  h->dbg_stall = member + 1;
  return GSV_OK;
}

int gsv_t2s_engine_stats(gsv_t2s_t* h, int* engine_available, int* fallbacks, unsigned* last_error3) {
  GSV_REQUIRE(h && h->finalized, "t2s_engine_stats: handle not finalized");
  if (engine_available) *engine_available = h->mega.ready && h->mega.census != 0 ? 1 : 0;
  if (fallbacks) *fallbacks = h->mega.fallbacks;
  if (last_error3) { last_error3[0] = h->mega.last_err[0]; last_error3[1] = h->mega.last_err[1]; last_error3[2] = h->mega.last_err[2]; }
  return GSV_OK;
}

int gsv_t2s_decode_info(gsv_t2s_t* h, int* mode, float* device_ms, int* steps) {
  GSV_REQUIRE(h && h->finalized, "t2s_decode_info: handle not finalized");
  if (mode) *mode = h->last_decode_mode;
  if (device_ms) *device_ms = h->last_decode_ms;
  if (steps) *steps = h->last_decode_steps;
  return GSV_OK;
}

int gsv_t2s_debug_logits(gsv_t2s_t* h, float* out, gsv_stream_t stream) {
  GSV_REQUIRE(h && h->finalized && out && h->B > 0, "t2s_debug_logits: bad state");
  GSV_HIP(hipMemcpyAsync(out, h->logits, (size_t)h->B * h->cfg.vocab * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GSV_OK;
}

int64_t gsv_t2s_step_bytes(gsv_t2s_t* h, int64_t* attn_bytes) {
  if (!h || !h->finalized) return 0;
  const auto& c = h->cfg;
  const int64_t es = (int64_t)esz(h), d = c.dim, ff = c.ffn_dim;
  std::vector<int> kvl(h->B > 0 ? h->B : 1, 0), act(h->B > 0 ? h->B : 1, 0);
  int64_t keys = 0;
  if (h->B > 0) {
    (void)hipMemcpy(kvl.data(), h->d_kv_len, h->B * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < h->B; ++b) keys += kvl[b] + 1;
  }
  // per cached key and layer: K and V rows of d elements
  const int64_t attn = keys * 2 * d * es;              // one layer's launch
  const int64_t weights = ((int64_t)c.n_layer * (3 * d * d + d * d + 2 * d * ff) + (int64_t)c.vocab * d) * es;
  if (attn_bytes) *attn_bytes = attn;
  return weights + attn * c.n_layer + (int64_t)h->B * 2 * d * es * c.n_layer;
}

int gsv_t2s_time_step(gsv_t2s_t* h, int iters, float* step_ms, float* attn_ms, gsv_stream_t stream) {
  // In-situ timing of the decode step at the current cache state: the full per-layer kernel
  // sequence (QKV+append, attention, out-proj, FFN1, FFN2) is launched eagerly on `stream` with a
  // HIP event pair around every decode-attention launch, so each attention launch runs behind its
  // producer and in front of its consumer exactly as in the replayed graph.  Rows are forced
  // active for the measurement (finished rows skip attention) and restored afterwards; no row
  // state advances because the sampling tail is not launched.
  GSV_REQUIRE(h && h->finalized && h->B > 0 && iters > 0, "t2s_time_step: bad state");
  hipStream_t s = (hipStream_t)stream;
  const int L = h->cfg.n_layer, B = h->B;
  std::vector<int> saved(B), ones(B, 1);
  GSV_HIP(hipMemcpy(saved.data(), h->d_active, B * 4, hipMemcpyDeviceToHost));
  GSV_HIP(hipMemcpy(h->d_active, ones.data(), B * 4, hipMemcpyHostToDevice));
  std::vector<hipEvent_t> ev((size_t)iters * 2 * L + 2);
  for (auto& e : ev) GSV_HIP(hipEventCreate(&e));
  int rc = GSV_OK;
  for (int w = 0; w < 2 && !rc; ++w)
    rc = h->dtype == GSV_F16 ? launch_decode_layers<_Float16>(h, s, 0) : launch_decode_layers<float>(h, s, 0);
  GSV_HIP(hipEventRecord(ev[(size_t)iters * 2 * L], s));
  for (int i = 0; i < iters && !rc; ++i)
    rc = h->dtype == GSV_F16 ? launch_decode_layers<_Float16>(h, s, 0) : launch_decode_layers<float>(h, s, 0);
  GSV_HIP(hipEventRecord(ev[(size_t)iters * 2 * L + 1], s));
  GSV_HIP(hipStreamSynchronize(s));
  // (b) the attention kernel alone: iters x L launches back to back between ONE event pair (an event pair
  // per launch adds ~3 us of its own); the L layers' arenas are distinct memory (L x bytes > Infinity Cache
  // at the benchmark shape), so every launch streams its K/V from HBM like it does inside a step
  float attn_total = 0.f;
  if (!rc) {
    GSV_HIP(hipEventRecord(ev[0], s));
    for (int i = 0; i < iters && !rc; ++i)
      rc = h->dtype == GSV_F16 ? launch_decode_layers<_Float16>(h, s, 1) : launch_decode_layers<float>(h, s, 1);
    GSV_HIP(hipEventRecord(ev[1], s));
    GSV_HIP(hipStreamSynchronize(s));
    if (!rc) GSV_HIP(hipEventElapsedTime(&attn_total, ev[0], ev[1]));
  }
  GSV_HIP(hipMemcpy(h->d_active, saved.data(), B * 4, hipMemcpyHostToDevice));
  if (!rc) {
    if (attn_ms) *attn_ms = attn_total / (float)(iters * L);
    float ms = 0.f;
    GSV_HIP(hipEventElapsedTime(&ms, ev[(size_t)iters * 2 * L], ev[(size_t)iters * 2 * L + 1]));
    if (step_ms) *step_ms = ms / iters;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int gsv_t2s_debug_set_state(gsv_t2s_t* h, int B, int kv_len) {
  // measurement hook: pretend B rows hold kv_len cached positions each (cache contents are whatever the
  // arena holds); used by tools/attn_sweep.py to time the decode-attention kernel at other (B, S) points
  GSV_REQUIRE(h && h->finalized, "t2s_debug_set_state: handle not finalized");
  GSV_REQUIRE(B >= 1 && B <= h->max_batch && kv_len >= 1 && kv_len + 2 <= h->max_seq, "t2s_debug_set_state: out of range");
  std::vector<int> kv(B, kv_len), zero(B, 0);
  GSV_HIP(hipMemcpy(h->d_kv_len, kv.data(), B * 4, hipMemcpyHostToDevice));
  GSV_HIP(hipMemcpy(h->d_active, zero.data(), B * 4, hipMemcpyHostToDevice));
  GSV_HIP(hipMemset(h->ybuf, 0, (size_t)B * h->cfg.dim * 4));
  GSV_HIP(hipMemset(h->kv, 0, (size_t)h->cfg.n_layer * 2 * h->kv_layer_stride * esz(h)));
  h->B = B;
  return GSV_OK;
}

int gsv_op_decode_attn(const void* q, const void* kc, const void* vc, const int32_t* kv_len, const int32_t* active, int B, int H,
                       int smax, int dtype, void* out, gsv_stream_t stream) {
  GSV_REQUIRE(q && kc && vc && kv_len && active && out, "op_decode_attn: null pointer");
  GSV_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && smax >= 1, "op_decode_attn: bad shape B=%d H=%d smax=%d", B, H, smax);
  GSV_REQUIRE(dtype == GSV_F16 || dtype == GSV_F32, "op_decode_attn: bad dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GSV_F16)
    hipLaunchKernelGGL((decode_attn_kernel<_Float16, 32>), dim3(H, B), dim3(256), 0, s, (const _Float16*)q, (const _Float16*)kc,
                       (const _Float16*)vc, kv_len, active, H, smax, (_Float16*)out);
  else
    hipLaunchKernelGGL((decode_attn_kernel<float, 32>), dim3(H, B), dim3(256), 0, s, (const float*)q, (const float*)kc,
                       (const float*)vc, kv_len, active, H, smax, (float*)out);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

int gsv_op_sample(const float* logits, int B, int vocab, int vocab_eff, const int32_t* prev, int prev_len,
                  const gsv_sampling_params* sp, const float* noise, int step, int32_t* sampled, int32_t* argmax_tok,
                  gsv_stream_t stream) {
  GSV_REQUIRE(logits && sp && sampled && argmax_tok && B > 0, "op_sample: null argument");
  GSV_REQUIRE(vocab <= 2048 && vocab_eff <= vocab, "op_sample: vocab too large");
  hipStream_t s = (hipStream_t)stream;
  const int npl = sample_npl(vocab);
  const size_t lds = (size_t)((vocab + 15) & ~15);
#define GSV_SO(N)                                                                                                   \
  hipLaunchKernelGGL(sample_only_kernel<N>, dim3(B), dim3(64), lds, s, logits, vocab, vocab_eff, prev, prev_len, sp->top_k, \
                     sp->top_p, sp->temperature, sp->repetition_penalty, noise, (unsigned long long)sp->seed, step, sampled, \
                     argmax_tok)
  if (npl == 2) GSV_SO(2);
  else if (npl == 17) GSV_SO(17);
  else GSV_SO(32);
#undef GSV_SO
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

}  // extern "C"

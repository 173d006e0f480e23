// Fused softmax attention for the DiT blocks (H14, reference f5_tts/model/modules.py:397-457, F.scaled_dot_product_attention
// without a mask), fp16, head dim 64, gfx950.
//
// The problem is small (T ~ 1000 frames, 16 heads: 3.6 GFLOP per block) and the materialised path (scores GEMM ->
// softmax -> V transpose -> PV GEMM) spends 4 launches and a 56 MB fp32 score round trip on it, so the work is cut
// for LATENCY, not for tile reuse: one workgroup per (16 query rows, head) = 944 workgroups at T = 934, the key range
// dealt in 32-key chunks to the workgroup's 4 waves (3.7 waves per SIMD hide each other's load latency), every wave
// running an online softmax over its chunks, and one LDS combine at the end.
//
// MFMA operand trick (no LDS transpose of P): scores are computed TRANSPOSED, S^T = K Q^T with mfma_f32_16x16x32_f16
// (A = 16 keys x 32 d, B = Q^T), so a lane holds, for its query q = lane & 15, MFMA rows {4g..4g+3} of two 16-row score
// tiles (g = lane >> 4).  The K rows are loaded PERMUTED (row 4g'+i of tile a = key 8g'+i, of tile b = key 8g'+4+i), so
// those 8 values are the 8 consecutive keys 8g..8g+7 of the 32-key chunk, and they ARE a valid B operand (k-slot (g, j))
// of the second product O^T = V^T P^T whose A operand is then ONE aligned 16-byte load per d-tile from the
// pre-transposed V (vt_kernel below): fragment loads are what these kernels are bound by.  The running max / sum / rescale factors of query q then live in the lanes that
// hold O^T[.][q]: no cross-lane traffic except two xor-shuffles per reduction.
#include <stdlib.h>

#include "common.h"

// order fence for the software pipelines: the empty asm stops IR-level sinking / hoisting of the loads across it, the
// sched_barrier stops the machine scheduler
#define GSV_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace gsv {

__device__ __forceinline__ f4 mma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Vt[head][d][key] = V[key][head*64 + d], keys zero padded to ldv (a multiple of 32).  The extra grid plane z == heads
// applies the DiT's rotary embedding in place to the first 2*rope_half channels of q and k (x_transformers'
// apply_rotary_pos_emb on the un-split projections, modules.py:420-427): same launch, one kernel less per block.
__global__ __launch_bounds__(256) void vt_kernel(const _Float16* __restrict__ v, int ld, int T, int ldv, _Float16* __restrict__ vt,
                                                 int heads, _Float16* __restrict__ q, int ldq, _Float16* __restrict__ k, int ldk,
                                                 const float* __restrict__ rope_cs, int rope_half) {
  if ((int)blockIdx.z == heads) {
    const int n = T * rope_half * 2;
    const int stride = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
      const int which = i / (T * rope_half);
      const int rr = i - which * T * rope_half;
      const int t = rr / rope_half, p = rr - t * rope_half;
      _Float16* x = (which ? k + (long long)t * ldk : q + (long long)t * ldq) + 2 * p;
      const float c = rope_cs[((long long)t * rope_half + p) * 2], sn = rope_cs[((long long)t * rope_half + p) * 2 + 1];
      const float a = (float)x[0], b = (float)x[1];
      x[0] = (_Float16)(a * c - b * sn);
      x[1] = (_Float16)(b * c + a * sn);
    }
    return;
  }
  __shared__ _Float16 tile[32][34];
  const int head = blockIdx.z, j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int j = j0 + i;
    tile[i][tx] = j < T ? v[(long long)j * ld + head * 64 + c0 + tx] : (_Float16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int j = j0 + tx;
    if (j < ldv) vt[((long long)head * 64 + c0 + i) * ldv + j] = tile[tx][i];
  }
}

// QT = 16-query tiles per workgroup.  Every K / V^T fragment a wave loads feeds QT score tiles and QT output tiles:
// fragment-shaped global loads cost ~120 clocks of the CU's texture-address path each (measured),
// and with QT = 1 those loads, not MFMA or latency, set the kernel's time.
template <int QT>
__global__ __launch_bounds__(256) void flash_attn64_f16_kernel(const _Float16* __restrict__ q, int ldq, const _Float16* __restrict__ k,
                                                               int ldk, const _Float16* __restrict__ vt, int ldv, int T, float scale,
                                                               _Float16* __restrict__ out, int ldo) {
  constexpr int LDO = 68, BQ = 16 * QT;
  extern __shared__ float smem[];
  float* Os = smem;                                  // [4][BQ][LDO]
  float* Ms = smem + 4 * BQ * LDO;                   // [4][BQ]
  float* Ls = Ms + 4 * BQ;                           // [4][BQ]
  // heads are dealt to XCDs in contiguous runs (2 heads per XCD at 16 heads): a head's K / V^T stay in one L2
  const int vid = xcd_virtual_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int head = vid / gridDim.x, q0 = (vid % gridDim.x) * BQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  h8 qf0[QT], qf1[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const _Float16* qp = q + (long long)min(q0 + 16 * t + r, T - 1) * ldq + head * 64 + g * 8;
    qf0[t] = *(const h8*)qp;
    qf1[t] = *(const h8*)(qp + 32);
  }
  const _Float16* kh = k + head * 64 + g * 8;
  const _Float16* vh = vt + (long long)head * 64 * ldv + (long long)r * ldv + 8 * g;
  const int kra = 8 * (r >> 2) + (r & 3);           // MFMA row r of score tile a <-> key offset (tile b: + 4): see header
  f4 o[QT][4];
  float m[QT], l[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[t][d] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  const int nchunks = (T + 31) >> 5;
  // Two operand sets, used alternately by a loop unrolled by two: with ONE set the prefetched registers have to be
  // copied into the loop-carried ones at the back edge, and that copy waits for vmcnt(0) -- the prefetch then hides
  // nothing (seen in the ISA: vmcnt(12) ... vmcnt(0) inside one trip).
  struct KV { h8 ka0, ka1, kb0, kb1; h8 v[4]; };
  const int lastc = nchunks - 1;
  auto fetch = [&](KV& f, int c) {
    const int key0 = c << 5;
    const _Float16* pa = kh + (long long)min(key0 + kra, T - 1) * ldk;
    const _Float16* pb = kh + (long long)min(key0 + kra + 4, T - 1) * ldk;
    f.ka0 = *(const h8*)pa; f.ka1 = *(const h8*)(pa + 32); f.kb0 = *(const h8*)pb; f.kb1 = *(const h8*)(pb + 32);
#pragma unroll
    for (int d = 0; d < 4; ++d) f.v[d] = *(const h8*)(vh + (long long)(d * 16) * ldv + key0);
  };
  auto process = [&](const KV& f, int c, bool valid) {
    const int key0 = c << 5;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f4 sa = (f4){0.f, 0.f, 0.f, 0.f}, sb = sa;
      sa = mma16(f.ka0, qf0[t], sa); sa = mma16(f.ka1, qf1[t], sa);
      sb = mma16(f.kb0, qf0[t], sb); sb = mma16(f.kb1, qf1[t], sb);
      float p[8];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) {                  // lane (q = r, g) holds the 8 CONSECUTIVE keys key0 + 8g .. + 7
        p[i] = (valid && key0 + 8 * g + i < T) ? sa[i] * scale : -INFINITY;
        p[4 + i] = (valid && key0 + 8 * g + 4 + i < T) ? sb[i] * scale : -INFINITY;
        mx = fmaxf(mx, fmaxf(p[i], p[4 + i]));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m[t], mx);           // finite: the first chunk of a trip is always valid and has a valid key
      const float alpha = __expf(m[t] - mnew);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - mnew); ps += p[i]; }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l[t] = l[t] * alpha + ps;
      m[t] = mnew;
      const h8 pf = (h8){(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3], (_Float16)p[4], (_Float16)p[5], (_Float16)p[6], (_Float16)p[7]};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[t][d] *= alpha;
        o[t][d] = mma16(f.v[d], pf, o[t][d]);       // O^T[d*16 + 4g + i][q = 16t + r]
      }
    }
  };
  KV fA, fB;
  fetch(fA, min(wave, lastc));                    // loads are unconditional (index clamped): see gemm_sk.hip
  for (int c = wave; c < nchunks; c += 8) {
    fetch(fB, min(c + 4, lastc));
    GSV_PIN();
    process(fA, c, true);
    GSV_PIN();
    fetch(fA, min(c + 8, lastc));
    GSV_PIN();
    process(fB, c + 4, c + 4 < nchunks);         // unconditional (fully masked past the end): a conditional block lets
                                                 // the compiler sink fB's loads into it, behind process(fA)
    GSV_PIN();
  }
  // ---- combine the 4 waves' partial (m, l, O)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int d = 0; d < 4; ++d) *(f4*)&Os[((size_t)wave * BQ + 16 * t + r) * LDO + d * 16 + 4 * g] = o[t][d];
    if (g == 0) { Ms[wave * BQ + 16 * t + r] = m[t]; Ls[wave * BQ + 16 * t + r] = l[t]; }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qq = 16 * t + (threadIdx.x >> 4), d4 = (threadIdx.x & 15) * 4;
    const float mt = fmaxf(fmaxf(Ms[qq], Ms[BQ + qq]), fmaxf(Ms[2 * BQ + qq], Ms[3 * BQ + qq]));
    float den = 0.f;
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(Ms[w * BQ + qq] - mt); // exp(-inf) = 0 for a wave that saw no chunk
      den += e * Ls[w * BQ + qq];
      acc += *(const f4*)&Os[((size_t)w * BQ + qq) * LDO + d4] * e;
    }
    if (q0 + qq < T) {
      const float inv = 1.f / den;
      *(h4*)(out + (long long)(q0 + qq) * ldo + head * 64 + d4) =
          (h4){(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
    }
  }
}

template <int QT>
static int launch_flash_qt(const void* q, int ldq, const void* k, int ldk, const void* vt_buf, int ldv, int T, int heads, float scale,
                           void* out, int ldo, hipStream_t s) {
  const size_t lds = ((size_t)4 * 16 * QT * 68 + 8 * 16 * QT) * 4;
  static bool attr = false;
  if (!attr) {
    GSV_HIP(hipFuncSetAttribute((const void*)flash_attn64_f16_kernel<QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  hipLaunchKernelGGL(flash_attn64_f16_kernel<QT>, dim3(cdiv(T, 16 * QT), heads), dim3(256), lds, s, (const _Float16*)q, ldq,
                     (const _Float16*)k, ldk, (const _Float16*)vt_buf, ldv, T, scale, (_Float16*)out, ldo);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

// ---------------------------------------------------------------------------------------
// enc_p self-attention with window-4 relative positions (H10, reference module/attentions.py:227-258), fp16, head dim 96:
// the same transposed-score construction; the relative-key bias b[i][r] = (q_i / sqrt(d)) . rel_k[r] is added to the
// scores of the 9 keys |j - i| <= 4 (only the 2-3 chunks that straddle the diagonal take that path), their biased LOGITS
// are parked in LDS (logits need no online rescaling), and the epilogue turns them into the band probabilities that
// weight rel_v (attentions.py:253-256).  Replaces scores GEMM -> softmax_rows -> V transpose -> PV GEMM -> relv_add and
// their 2 x T^2 x 6 bytes of HBM round trips (T = 6400 folded frames at the benchmark shape).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vt96_kernel(const _Float16* __restrict__ v, int ld, int T, int ldv, _Float16* __restrict__ vt) {
  __shared__ _Float16 tile[32][34];
  const int head = blockIdx.z, j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int j = j0 + i;
    tile[i][tx] = j < T ? v[(long long)j * ld + head * 96 + c0 + tx] : (_Float16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int j = j0 + tx;
    if (j < ldv) vt[((long long)head * 96 + c0 + i) * ldv + j] = tile[tx][i];
  }
}

template <int QT>
__global__ __launch_bounds__(256) void flash_rel96_f16_kernel(const _Float16* __restrict__ q, int ldq, const _Float16* __restrict__ k,
                                                              int ldk, const _Float16* __restrict__ vt, int ldv, int T, float scale,
                                                              const float* __restrict__ rel_k, const float* __restrict__ rel_v,
                                                              _Float16* __restrict__ out, int ldo) {
  constexpr int D = 96, KS = 3, DT = 6, W = 4, NB = 9, LDO = 100, BQ = 16 * QT;
  extern __shared__ float smem[];
  float* Os = smem;                                  // [4][BQ][LDO]
  float* Ms = Os + 4 * BQ * LDO;                     // [4][BQ]
  float* Ls = Ms + 4 * BQ;                           // [4][BQ]
  float* Bias = Ls + 4 * BQ;                         // [BQ][NB]   relative-key bias per query
  float* Sb = Bias + BQ * NB;                        // [BQ][NB]   biased logits of the band keys (-inf outside [0, T))
  const int vid = xcd_virtual_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int head = vid / gridDim.x, q0 = (vid % gridDim.x) * BQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  h8 qf[QT][KS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const _Float16* qp = q + (long long)min(q0 + 16 * t + r, T - 1) * ldq + head * D + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[t][ks] = *(const h8*)(qp + 32 * ks);
  }
  // relative-key bias: wave w computes the tiles t = w, w + 4, ...; each lane dots its 24 channels, the 4 lanes of a query sum
  for (int i = threadIdx.x; i < BQ * NB; i += 256) Sb[i] = -INFINITY;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if ((t & 3) != wave) continue;
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) {
      float sdot = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) sdot += (float)qf[t][ks][e] * rel_k[rr * D + ks * 32 + g * 8 + e];
      sdot += __shfl_xor(sdot, 16, 64);
      sdot += __shfl_xor(sdot, 32, 64);
      if (g == 0) Bias[(16 * t + r) * NB + rr] = sdot * scale;
    }
  }
  __syncthreads();
  const _Float16* kh = k + head * D + g * 8;
  const _Float16* vh = vt + (long long)head * D * ldv + (long long)r * ldv + 8 * g;
  const int kra = 8 * (r >> 2) + (r & 3);
  f4 o[QT][DT];
  float m[QT], l[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) o[t][d] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  const int nchunks = (T + 31) >> 5, lastc = nchunks - 1;
  struct KV { h8 ka[KS], kb[KS]; h8 v[DT]; };
  auto fetch = [&](KV& f, int c) {
    const int key0 = c << 5;
    const _Float16* pa = kh + (long long)min(key0 + kra, T - 1) * ldk;
    const _Float16* pb = kh + (long long)min(key0 + kra + 4, T - 1) * ldk;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { f.ka[ks] = *(const h8*)(pa + 32 * ks); f.kb[ks] = *(const h8*)(pb + 32 * ks); }
#pragma unroll
    for (int d = 0; d < DT; ++d) f.v[d] = *(const h8*)(vh + (long long)(d * 16) * ldv + key0);
  };
  auto process = [&](const KV& f, int c, bool valid) {
    const int key0 = c << 5;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f4 sa = (f4){0.f, 0.f, 0.f, 0.f}, sb = sa;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { sa = mma16(f.ka[ks], qf[t][ks], sa); sb = mma16(f.kb[ks], qf[t][ks], sb); }
      float p[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = (valid && key0 + 8 * g + i < T) ? sa[i] * scale : -INFINITY;
        p[4 + i] = (valid && key0 + 8 * g + 4 + i < T) ? sb[i] * scale : -INFINITY;
      }
      const int qt0 = q0 + 16 * t;
      if (valid && key0 + 31 >= qt0 - W && key0 <= qt0 + 15 + W) {      // wave-uniform: this chunk touches the diagonal band
        const int qi = qt0 + r;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = key0 + 8 * g + i;
          const int rp = j - qi + W;
          if (rp >= 0 && rp < NB && j < T && qi < T) {
            p[i] += Bias[(16 * t + r) * NB + rp];
            Sb[(16 * t + r) * NB + rp] = p[i];
          }
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, p[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m[t], mx);
      const float alpha = __expf(m[t] - mnew);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - mnew); ps += p[i]; }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l[t] = l[t] * alpha + ps;
      m[t] = mnew;
      const h8 pf = (h8){(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3], (_Float16)p[4], (_Float16)p[5], (_Float16)p[6], (_Float16)p[7]};
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        o[t][d] *= alpha;
        o[t][d] = mma16(f.v[d], pf, o[t][d]);
      }
    }
  };
  KV fA, fB;
  fetch(fA, min(wave, lastc));
  for (int c = wave; c < nchunks; c += 8) {
    fetch(fB, min(c + 4, lastc));
    GSV_PIN();
    process(fA, c, true);
    GSV_PIN();
    fetch(fA, min(c + 8, lastc));
    GSV_PIN();
    process(fB, c + 4, c + 4 < nchunks);
    GSV_PIN();
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int d = 0; d < DT; ++d) *(f4*)&Os[((size_t)wave * BQ + 16 * t + r) * LDO + d * 16 + 4 * g] = o[t][d];
    if (g == 0) { Ms[wave * BQ + 16 * t + r] = m[t]; Ls[wave * BQ + 16 * t + r] = l[t]; }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < BQ * (D / 4); it += 256) {
    const int qq = it / (D / 4), d4 = (it % (D / 4)) * 4;
    if (q0 + qq >= T) continue;
    const float mt = fmaxf(fmaxf(Ms[qq], Ms[BQ + qq]), fmaxf(Ms[2 * BQ + qq], Ms[3 * BQ + qq]));
    float den = 0.f;
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(Ms[w * BQ + qq] - mt);
      den += e * Ls[w * BQ + qq];
      acc += *(const f4*)&Os[((size_t)w * BQ + qq) * LDO + d4] * e;
    }
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) {
      const float pw = __expf(Sb[qq * NB + rr] - mt);                   // band probability * den; exp(-inf) = 0 outside [0, T)
      const f4 rv = *(const f4*)(rel_v + rr * D + d4);
      acc += rv * pw;
    }
    const float inv = 1.f / den;
    *(h4*)(out + (long long)(q0 + qq) * ldo + head * D + d4) =
        (h4){(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
  }
}

// self-attention with relative positions (window 4), fp16, head dim 96; vt_buf: heads * 96 * ceil32(T) halfs of scratch
int launch_flash_rel96_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldvv, void* vt_buf, int T, int heads,
                           float scale, const float* rel_k, const float* rel_v, void* out, int ldo, hipStream_t s) {
  GSV_REQUIRE(T >= 1 && heads >= 1 && rel_k && rel_v, "flash_rel96: bad argument");
  GSV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)out % 8) == 0 &&
              ((uintptr_t)rel_v % 16) == 0, "flash_rel96: operands must be 16-byte aligned with leading dims multiple of 8");
  const int ldv = (T + 31) / 32 * 32;
  hipLaunchKernelGGL(vt96_kernel, dim3(ldv / 32, 3, heads), dim3(256), 0, s, (const _Float16*)v, ldvv, T, ldv, (_Float16*)vt_buf);
  constexpr int QT = 2;
  const size_t lds = ((size_t)4 * 16 * QT * 100 + 8 * 16 * QT + 2 * 16 * QT * 9) * 4;
  static bool attr = false;
  if (!attr) {
    GSV_HIP(hipFuncSetAttribute((const void*)flash_rel96_f16_kernel<QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  hipLaunchKernelGGL(flash_rel96_f16_kernel<QT>, dim3(cdiv(T, 16 * QT), heads), dim3(256), lds, s, (const _Float16*)q, ldq,
                     (const _Float16*)k, ldk, (const _Float16*)vt_buf, ldv, T, scale, rel_k, rel_v, (_Float16*)out, ldo);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

// q / k: [T][ld] with head h at columns h*64..; v likewise; vt_buf: heads * 64 * ceil32(T) halfs of scratch
int launch_flash_attn64_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldvv, void* vt_buf, int T, int heads,
                            float scale, void* out, int ldo, hipStream_t s, const float* rope_cs, int rope_half, bool vt_ready) {
  GSV_REQUIRE(T >= 1 && heads >= 1, "flash_attn: empty problem");
  GSV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)out % 8) == 0,
              "flash_attn: operands must be 16-byte aligned with leading dims multiple of 8");
  const int ldv = (T + 31) / 32 * 32;
  if (!vt_ready)
    hipLaunchKernelGGL(vt_kernel, dim3(ldv / 32, 2, heads + (rope_cs ? 1 : 0)), dim3(256), 0, s, (const _Float16*)v, ldvv, T, ldv,
                       (_Float16*)vt_buf, heads, (_Float16*)const_cast<void*>(q), ldq, (_Float16*)const_cast<void*>(k), ldk, rope_cs, rope_half);
  static const int qt_env = getenv("GSV_FLASH_QT") ? atoi(getenv("GSV_FLASH_QT")) : 0;     // A/B switch
  // more query tiles per workgroup = fewer K / V fragment loads per query, but fewer workgroups: keep >= ~1 per CU
  int qt = qt_env ? qt_env : ((long long)cdiv(T, 64) * heads >= 200 ? 4 : ((long long)cdiv(T, 32) * heads >= 200 ? 2 : 1));
  if (qt >= 4) return launch_flash_qt<4>(q, ldq, k, ldk, vt_buf, ldv, T, heads, scale, out, ldo, s);
  if (qt >= 2) return launch_flash_qt<2>(q, ldq, k, ldk, vt_buf, ldv, T, heads, scale, out, ldo, s);
  return launch_flash_qt<1>(q, ldq, k, ldk, vt_buf, ldv, T, heads, scale, out, ldo, s);
}

}  // namespace gsv

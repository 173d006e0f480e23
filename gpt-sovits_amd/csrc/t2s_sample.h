// AR sampling (H5) shared by the launch-per-phase decode step (t2s.hip) and the persistent decode engine (t2s_mega.hip).
#pragma once
#include "common.h"

namespace gsv {

// ---------------------------------------------------------------------------------------
// device-side parameter block shared by the step kernels (lives in HBM so that the captured
// graph does not bake sampling parameters)
// ---------------------------------------------------------------------------------------
struct StepParams {
  int top_k;
  float top_p;
  float temperature;
  float rep_penalty;
  int early_stop_num;
  int eos_mask_steps;
  int max_steps;
  int noise_rows;          // 0: counter RNG, 1: shared noise, B: per-row noise
  unsigned long long seed;
  const float* noise;      // [max_steps][noise_rows][V] or null
  int* out_tokens;         // [B][max_steps]
  int* out_len;            // [B]
  int P;                   // prompt length (position offset of generated tokens)
  // parity hooks (gsv_t2s_set_debug; null in production): teacher forcing and a per-step dump of the raw logits
  const int* force;        // [B][max_steps]: the token taken at step s of row b instead of the sampled one
  float* dump;             // [max_steps][B][V]: logits of every executed step, before the repetition penalty
  int* drawn;              // [max_steps][B][2]: (token the sampler drew, argmax of the penalised logits) before any forcing
};


// ---------------------------------------------------------------------------------------
// Sampling (H5): one wave per row, the whole row in registers (NPL values per lane), all
// reductions are wavefront shuffles.  Semantics follow reference AR/models/utils.py:147-199:
// repetition penalty (in place, so the EOS argmax test of t2s_model.py:721 sees penalised
// logits) -> top-p on the un-tempered distribution -> /temperature -> top-k (ties kept) ->
// softmax -> argmax(p / Exp(1)).  No sort: tokens are extracted in descending order only as
// far as top-k / top-p need.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct ArgMax { float v; int i; };
// (largest value, lowest index on ties) over the wave.  DPP row operations + 4 v_readlane instead of six ds_bpermute
// butterfly steps on two registers: the extraction loop below calls this once per extracted token, and the shuffle version
// was most of its ~1.7 us per round (tools/sample_bench.py: top-k 15 cost 35 us per step).
template <int CTRL> __device__ __forceinline__ void argmax_dpp_step(float& v, int& i) {
  const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
  const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, true);
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ ArgMax wave_argmax(float v, int i) {
  argmax_dpp_step<0xB1>(v, i);      // quad_perm [1, 0, 3, 2]
  argmax_dpp_step<0x4E>(v, i);      // quad_perm [2, 3, 0, 1]
  argmax_dpp_step<0x141>(v, i);     // row_half_mirror
  argmax_dpp_step<0x140>(v, i);     // row_mirror: every lane of a 16-lane row now holds the row's result
  float bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
  for (int r = 16; r < 64; r += 16) {
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r));
    const int oi = __builtin_amdgcn_readlane(i, r);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  return {bv, bi};
}

// order-preserving map float -> unsigned (a > b <=> key(a) > key(b)); -0 and +0 share a key like they compare equal,
// NaN sorts below everything (the extraction loop never picks a NaN either)
__device__ __forceinline__ unsigned ordered_key(float x) {
  unsigned b = __float_as_uint(x);
  if (b == 0x80000000u) b = 0u;
  return x != x ? 0u : (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// everything after the repetition penalty: x[] holds the (penalised) logits of this lane's tokens v = lane + 64 i
template <int NPL>
__device__ __forceinline__ void sample_core(float (&x)[NPL], int Veff, int top_k, float top_p, float temperature,
                                            const float* __restrict__ noise_row, unsigned long long seed, int row, int step,
                                            int* out_sample, int* out_argmax) {
  const int lane = threadIdx.x & 63;
  // argmax of the penalised logits (first index on ties)
  float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int v = lane + 64 * i;
    if (v < Veff && (x[i] > bv)) { bv = x[i]; bi = v; }
  }
  ArgMax am = wave_argmax(bv, bi);
  *out_argmax = am.i;
  const float max0 = am.v;

  const bool use_p = top_p < 1.0f;
  const bool use_k = top_k > 0 && top_k < Veff;
  unsigned int keep = 0;  // bit i: x[i] survives the filters
  if (!use_p && !use_k) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (lane + 64 * i < Veff) keep |= 1u << i;
  } else if (!use_p && top_k == 1) {
    // greedy: the kept set is the maximum and its ties (what the extraction below ends with after two rounds, ~3 us)
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (lane + 64 * i < Veff && x[i] == max0) keep |= 1u << i;
    // a single maximum (no tie): it wins the race whatever the noise, so the softmax and the Exp(1) draw are skipped -- the
    // sampler sits between the logits hop and the next step's first hop, fully serialised with the step
    const unsigned long long owners = __ballot(keep != 0u);
    if (__builtin_popcountll(owners) == 1 &&
        __builtin_amdgcn_readlane(__builtin_popcount(keep), __builtin_ctzll(owners)) == 1) {
      *out_sample = am.i;
      return;
    }
  } else if (!use_p) {
    // top-k alone (the reference's CLI / API / web UI defaults leave top_p at 1): the k-th largest value by a radix select
    // on order-preserving integer keys -- 32 rounds of NPL compares whose wave-wide count is one s_bcnt1 on the compare
    // mask each, no cross-lane reduction, no dependence on k, and done as soon as a prefix separates exactly k values (the extraction loop below costs ~1.45 us per extracted
    // token: 32 us per step at top-k 15, 125 us at top-k 100; tools/sample_bench.py).  Keeps exactly the set the loop
    // keeps: values >= the k-th largest (ties kept, utils.py:171-174), never -inf.
    unsigned key[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) key[i] = lane + 64 * i < Veff ? ordered_key(x[i]) : 0u;
    unsigned thr = 0u;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned cand = thr | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < NPL; ++i) cnt += __builtin_popcountll(__ballot(key[i] >= cand));
      if (cnt >= top_k) thr = cand;
      if (cnt == top_k) break;      // exactly k values >= cand: that IS the kept set (a tie at the k-th value never counts k)
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (key[i] >= thr && x[i] > -INFINITY && lane + 64 * i < Veff) keep |= 1u << i;
  } else {
    // top-p (with or without top-k), reference utils.py:169-186: the kept set is a PREFIX of the descending order -- the tokens whose
    // inclusive cumulative probability stays <= top_p (rank 0 always) -- cut further to the values >= the k-th largest.  Both
    // prefixes are found as thresholds on the order-preserving integer keys: the mass of {key >= c} falls as c grows, so the
    // smallest c whose mass is <= top_p comes from a 32-round bit-by-bit search (17 compare-selects + one DPP wave sum per round,
    // ~0.3 us), independent of how many tokens survive.  The descending extraction this replaces cost 1.45 us per kept token:
    // 148 us per step at top-p 0.9 without top-k (profiles/r03_sampling.json).  Differences from a sequential cumulative sum
    // are confined to summation order (a boundary token within ~1e-6 of top_p) and to ties at the boundary (kept or dropped
    // together here; the reference's sort order on ties is unspecified): both "parity unpinned", DESIGN.md section 2.
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (lane + 64 * i < Veff) s += expf(x[i] - max0);
    const float S = wave_sum(s);
    unsigned key[NPL];
    float pr[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const bool ok = lane + 64 * i < Veff && x[i] > -INFINITY;
      key[i] = ok ? ordered_key(x[i]) : 0u;
      pr[i] = ok ? expf(x[i] - max0) / S : 0.f;
    }
    unsigned lo = 0u;                   // largest c whose mass is still > top_p (c = 0: everything, mass 1 > top_p)
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned cand = lo | (1u << bit);
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) m += key[i] >= cand ? pr[i] : 0.f;
      if (wave_sum(m) > top_p) lo = cand;
    }
    unsigned thr = lo + 1u;             // keys > lo: the largest prefix whose mass is <= top_p (lo = 0xffffffff cannot happen: NaN-free keys < it)
    const unsigned kmax = ordered_key(max0);
    if (thr > kmax) thr = kmax;         // rank 0 is always kept
    if (use_k) {
      unsigned tk = 0u;
#pragma unroll 1
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = tk | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) cnt += __builtin_popcountll(__ballot(key[i] >= cand));
        if (cnt >= top_k) tk = cand;
        if (cnt == top_k) break;
      }
      if (tk > thr) thr = tk;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (key[i] >= thr && x[i] > -INFINITY && lane + 64 * i < Veff) keep |= 1u << i;
  }
  // softmax over the kept set at temperature T, then the exponential race
  const float tdiv = fmaxf(temperature, 1e-5f);
  float lm = -INFINITY;
#pragma unroll
  for (int i = 0; i < NPL; ++i) if ((keep >> i) & 1u) lm = fmaxf(lm, x[i] / tdiv);
  lm = wave_max(lm);
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) if ((keep >> i) & 1u) z += expf(x[i] / tdiv - lm);
  z = wave_sum(z);
  float sv = -INFINITY; int si = 0x7fffffff;
  // filtered-out tokens have p = 0 and score 0 / q = 0: they can only win when no kept token has a positive score, which
  // cannot happen (rank 0 is always kept), so their Exp(1) draw is never needed.  Each lane walks ITS kept tokens (lowest
  // first): the wave runs max-over-lanes popcount(keep) bodies -- 1-2 for top-k 15 -- where a loop over the NPL slots ran
  // one body per slot that any lane kept (13 of 17 at top-k 15: 0.5 us each, tools/sample_bench.py).
  unsigned int todo = keep;
  while (todo) {
    const int i = __ffs(todo) - 1;
    todo &= todo - 1;
    float xi = x[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) xi = i == j ? x[j] : xi;
    const int v = lane + 64 * i;
    float p = expf(xi / tdiv - lm) / z;
    float qn;
    if (noise_row) qn = noise_row[v];
    else {
      unsigned long long hsh = splitmix64(seed ^ splitmix64(((unsigned long long)row << 40) ^ ((unsigned long long)step << 20) ^ (unsigned long long)v));
      float u = (float)(hsh >> 40) * (1.0f / 16777216.0f);
      qn = fmaxf(-log1pf(-u), 1e-20f);
    }
    float sc = p / qn;
    if (sc > sv || (sc == sv && v < si)) { sv = sc; si = v; }
  }
  ArgMax sm = wave_argmax(sv, si);
  *out_sample = sm.i;
}


template <int NPL>
__device__ void sample_row(const float* __restrict__ lg_row, int V, int Veff, const int* __restrict__ prev, int prev_len,
                           int top_k, float top_p, float temperature, float rp, const float* __restrict__ noise_row,
                           unsigned long long seed, int row, int step, unsigned char* seen /* LDS [V] */,
                           int* out_sample, int* out_argmax) {
  const int lane = threadIdx.x & 63;
  float x[NPL];
  // token v = lane + 64*i  (coalesced loads, low index first inside a lane)
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int v = lane + 64 * i;
    x[i] = (v < Veff) ? lg_row[v] : -INFINITY;
  }
  if (rp != 1.0f) {
    for (int v = lane; v < V; v += 64) seen[v] = 0;
    __syncthreads();
    for (int t = lane; t < prev_len; t += 64) {
      int tok = prev[t];
      if (tok >= 0 && tok < V) seen[tok] = 1;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int v = lane + 64 * i;
      if (v < Veff && seen[v]) x[i] = (x[i] < 0.f) ? x[i] * rp : x[i] / rp;
    }
  }
  sample_core<NPL>(x, Veff, top_k, top_p, temperature, noise_row, seed, row, step, out_sample, out_argmax);
}

}  // namespace gsv

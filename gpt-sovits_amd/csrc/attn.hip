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
// (A = 16 keys x 32 d, B = Q^T), so a lane holds, for its query q = lane & 15, the keys {4g..4g+3} of two 16-key tiles
// (g = lane >> 4).  Those 8 probabilities ARE a valid B operand (k-slot (g, j)) of the second product
// O^T = V^T P^T once V^T's A operand uses the same key <-> k-slot permutation, i.e. two 8-byte loads from the
// pre-transposed V (vt_kernel below).  The running max / sum / rescale factors of query q then live in the lanes that
// hold O^T[.][q]: no cross-lane traffic except two xor-shuffles per reduction.
#include "common.h"

namespace gsv {

__device__ __forceinline__ f4 mma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Vt[head][d][key] = V[key][head*64 + d], keys zero padded to ldv (a multiple of 32)
__global__ __launch_bounds__(256) void vt_kernel(const _Float16* __restrict__ v, int ld, int T, int ldv, _Float16* __restrict__ vt) {
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

__global__ __launch_bounds__(256) void flash_attn64_f16_kernel(const _Float16* __restrict__ q, int ldq, const _Float16* __restrict__ k,
                                                               int ldk, const _Float16* __restrict__ vt, int ldv, int T, float scale,
                                                               _Float16* __restrict__ out, int ldo) {
  constexpr int LDO = 68;
  __shared__ float Os[4][16][LDO];
  __shared__ float Ms[4][16], Ls[4][16];
  const int head = blockIdx.y, q0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int qrow = min(q0 + r, T - 1);
  const _Float16* qp = q + (long long)qrow * ldq + head * 64 + g * 8;
  const h8 qf0 = *(const h8*)qp, qf1 = *(const h8*)(qp + 32);
  const _Float16* kh = k + head * 64 + g * 8;
  const _Float16* vh = vt + (long long)head * 64 * ldv + (long long)r * ldv + 4 * g;
  f4 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (f4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int nchunks = (T + 31) >> 5;
  for (int c = wave; c < nchunks; c += 4) {
    const int key0 = c << 5;
    const int ka = min(key0 + r, T - 1), kb = min(key0 + 16 + r, T - 1);
    const _Float16* pa = kh + (long long)ka * ldk;
    const _Float16* pb = kh + (long long)kb * ldk;
    const h8 a0 = *(const h8*)pa, a1 = *(const h8*)(pa + 32), b0 = *(const h8*)pb, b1 = *(const h8*)(pb + 32);
    h4 va[4], vb[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const _Float16* pv = vh + (long long)(d * 16) * ldv + key0;
      va[d] = *(const h4*)pv;
      vb[d] = *(const h4*)(pv + 16);
    }
    f4 sa = (f4){0.f, 0.f, 0.f, 0.f}, sb = sa;
    sa = mma16(a0, qf0, sa); sa = mma16(a1, qf1, sa);
    sb = mma16(b0, qf0, sb); sb = mma16(b1, qf1, sb);
    float p[8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p[i] = (key0 + 4 * g + i < T) ? sa[i] * scale : -INFINITY;
      p[4 + i] = (key0 + 16 + 4 * g + i < T) ? sb[i] * scale : -INFINITY;
      mx = fmaxf(mx, fmaxf(p[i], p[4 + i]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(m, mx);              // finite: every chunk holds at least one valid key
    const float alpha = __expf(m - mnew);
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - mnew); ps += p[i]; }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
    m = mnew;
    const h8 pf = (h8){(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3], (_Float16)p[4], (_Float16)p[5], (_Float16)p[6], (_Float16)p[7]};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      o[d] *= alpha;
      const h8 av = (h8){va[d][0], va[d][1], va[d][2], va[d][3], vb[d][0], vb[d][1], vb[d][2], vb[d][3]};
      o[d] = mma16(av, pf, o[d]);                 // O^T[d*16 + 4g + i][q = r]
    }
  }
  // ---- combine the 4 waves' partial (m, l, O)
#pragma unroll
  for (int d = 0; d < 4; ++d) *(f4*)&Os[wave][r][d * 16 + 4 * g] = o[d];
  if (g == 0) { Ms[wave][r] = m; Ls[wave][r] = l; }
  __syncthreads();
  const int qq = threadIdx.x >> 4, d4 = (threadIdx.x & 15) * 4;
  float mt = fmaxf(fmaxf(Ms[0][qq], Ms[1][qq]), fmaxf(Ms[2][qq], Ms[3][qq]));
  float den = 0.f;
  f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float e = __expf(Ms[w][qq] - mt);       // exp(-inf) = 0 for a wave that saw no chunk
    den += e * Ls[w][qq];
    acc += *(const f4*)&Os[w][qq][d4] * e;
  }
  if (q0 + qq < T) {
    const float inv = 1.f / den;
    *(h4*)(out + (long long)(q0 + qq) * ldo + head * 64 + d4) =
        (h4){(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
  }
}

// q / k: [T][ld] with head h at columns h*64..; v likewise; vt_buf: heads * 64 * ceil32(T) halfs of scratch
int launch_flash_attn64_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldvv, void* vt_buf, int T, int heads,
                            float scale, void* out, int ldo, hipStream_t s) {
  GSV_REQUIRE(T >= 1 && heads >= 1, "flash_attn: empty problem");
  GSV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)out % 8) == 0,
              "flash_attn: operands must be 16-byte aligned with leading dims multiple of 8");
  const int ldv = (T + 31) / 32 * 32;
  hipLaunchKernelGGL(vt_kernel, dim3(ldv / 32, 2, heads), dim3(256), 0, s, (const _Float16*)v, ldvv, T, ldv, (_Float16*)vt_buf);
  hipLaunchKernelGGL(flash_attn64_f16_kernel, dim3(cdiv(T, 16), heads), dim3(256), 0, s, (const _Float16*)q, ldq, (const _Float16*)k, ldk,
                     (const _Float16*)vt_buf, ldv, T, scale, (_Float16*)out, ldo);
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

}  // namespace gsv

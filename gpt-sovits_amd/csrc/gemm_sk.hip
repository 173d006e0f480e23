// "Skinny" fp16 GEMM for gfx950: y[t][co] = epilogue(sum_k w[co][k] * x[t][k]) when the output has too few 128 x 128
// tiles to fill 256 CUs (the DiT's Linear layers at T ~ 1000 frames: 64-192 tiles; prefill / encoder 1x1 layers).
//
// There the LDS-tiled kernel (conv_lds.hip) is latency-bound, not MFMA-bound: one workgroup per CU walks 16-32
// dependent global -> LDS -> barrier -> MFMA steps of ~2 us each.  This kernel is cut the other way:
//   * 64 x 64 output tile per workgroup (4x as many workgroups), and the K range dealt in 64-wide chunks to the
//     workgroup's 4 waves (in-workgroup split-K): each wave streams its chunks straight from global memory into MFMA
//     operand registers -- no LDS, no barrier in the main loop -- with the next chunk's 16 loads in flight under the
//     current chunk's 16 MFMAs (64 KB in flight per CU);
//   * k-slot permutation: within a chunk lane (r, h) owns the 64 contiguous bytes k = 32h .. 32h+31 of row r and MFMA j
//     takes its j-th 16-byte piece from BOTH operands, so every lane load is one aligned 64-byte run and the two lanes
//     of a row cover one 128-byte line;
//   * the 4 partial tiles are summed through LDS in a fixed order (deterministic), then bias / gate / residual /
//     activation are applied and whole channels-last row segments (128 B per 4 threads) are stored.
#include "common.h"

// order fence for the software pipelines: the empty asm stops IR-level sinking / hoisting of the loads across it, the
// sched_barrier stops the machine scheduler
#define GSV_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace gsv {

namespace {

__device__ __forceinline__ void mma32(f16v& acc, h8 a, h8 b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); }

// bias / gate / residual / scale / activation on 16 consecutive output channels of row t, stored as two 16-byte pieces
__device__ __forceinline__ void row_epilogue(const ConvArgs& a, float (&v)[16], int t, int cbase_in) {
  const int cbase = cbase_in;
  const int nv = max(0, min(16, a.Cout - cbase));
  if (nv == 0) return;
  const long long yoff = (long long)t * a.ldy + a.y_col0 + cbase;
  const long long roff = (long long)t * a.ldr + cbase;
  const bool full = nv == 16;
  const bool vec_ok = full && ((a.ldy & 7) == 0) && ((a.y_col0 & 7) == 0) && ((a.ldr & 7) == 0);
  const bool has_act = a.post_act != ACT_NONE;
  float rr[16];
  if (a.res) {
    if (vec_ok && !a.res_f32) {
      const h8 r0 = *(const h8*)((const _Float16*)a.res + roff), r1 = *(const h8*)((const _Float16*)a.res + roff + 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { rr[e] = (float)r0[e]; rr[8 + e] = (float)r1[e]; }
    } else {
      for (int e = 0; e < nv; ++e) rr[e] = a.res_f32 ? ((const float*)a.res)[roff + e] : (float)((const _Float16*)a.res)[roff + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (e >= nv) break;
    float u = v[e];
    if (a.bias) u += a.bias[cbase + e];
    if (a.gate) u *= a.gate[cbase + e];
    if (a.res) u += rr[e];
    u *= a.scale;
    v[e] = has_act ? post_act_f(a.post_act, u) : u;
  }
  if (a.out_f32) {
    float* yp = (float*)a.y + yoff;
    if (vec_ok) {
#pragma unroll
      for (int e = 0; e < 16; e += 4) *(f4*)(yp + e) = (f4){v[e], v[e + 1], v[e + 2], v[e + 3]};
    } else {
      for (int e = 0; e < nv; ++e) yp[e] = v[e];
    }
  } else {
    _Float16* yp = (_Float16*)a.y + yoff;
    if (vec_ok) {
      *(h8*)yp = (h8){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
      *(h8*)(yp + 8) = (h8){(_Float16)v[8], (_Float16)v[9], (_Float16)v[10], (_Float16)v[11], (_Float16)v[12], (_Float16)v[13], (_Float16)v[14], (_Float16)v[15]};
    } else {
      for (int e = 0; e < nv; ++e) yp[e] = (_Float16)v[e];
    }
  }
}

struct Frag { h8 a[2][4], b[2][4]; };

__global__ __launch_bounds__(256) void gemm_sk_f16_kernel(ConvArgs a) {
  constexpr int LDO = 68;
  extern __shared__ float os[];                     // [4 waves][64 t][LDO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // each XCD takes a contiguous band of output-channel tiles: its L2 holds 1/8 of the weights plus the activations
  const int vid = xcd_virtual_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int t0 = (vid % gridDim.x) * 64, c0 = (vid / gridDim.x) * 64;
  const _Float16* __restrict__ x = (const _Float16*)a.x;
  const _Float16* __restrict__ w = (const _Float16*)a.w;
  const _Float16* wp[2];
  const _Float16* xp[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    wp[m] = w + (long long)min(c0 + 32 * m + r, a.Cout - 1) * a.ldw + 32 * h;
    xp[m] = x + (long long)min(t0 + 32 * m + r, a.T_in - 1) * a.ldx + 32 * h;
  }
  f16v acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  const int nchunks = a.Cin >> 6;
  auto load = [&](Frag& f, int c) {
    const int k0 = c << 6;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f.a[m][j] = *(const h8*)(wp[m] + k0 + 8 * j);
        f.b[m][j] = *(const h8*)(xp[m] + k0 + 8 * j);
      }
  };
  auto compute = [&](const Frag& f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) mma32(acc[m][n], f.a[m][j], f.b[n][j]);
  };
  // Prefetch rules learned from the ISA: (1) loads are UNCONDITIONAL (chunk index clamped, a
  // spare tail load is harmless) -- a load under a runtime condition makes the compiler branch around it and drain
  // vmcnt(0) at the join; (2) two register sets used alternately by a loop unrolled by two -- with one set the
  // back-edge copy waits for vmcnt(0); (3) sched_barrier pins each 16-MFMA block behind the OTHER set's 16 loads, so
  // the waits are counted (vmcnt(16)) and a whole chunk is in flight under the MFMAs.
  Frag f0, f1;
  const int last = nchunks - 1;
  load(f0, min(wave, last));
  for (int c = wave; c < nchunks; c += 8) {
    load(f1, min(c + 4, last));
    GSV_PIN();
    compute(f0);
    GSV_PIN();
    load(f0, min(c + 8, last));
    GSV_PIN();
    if (c + 4 < nchunks) compute(f1);
    GSV_PIN();
  }
  // ---- partial tiles -> LDS as [t][co] rows (lane holds column t = 32n + r, rows co = 32m + (i&3) + 8(i>>2) + 4h)
  float* mine = os + (size_t)wave * 64 * LDO;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f4*)(mine + (size_t)(32 * n + r) * LDO + 32 * m + 8 * q + 4 * h) =
            (f4){acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
  __syncthreads();
  // ---- fixed-order sum + epilogue: thread -> row tl = tid / 4, 16 channels at 16 * (tid % 4)
  const int tl = tid >> 2, cs = (tid & 3) * 16;
  const int t = t0 + tl;
  if (t >= a.T_virt || t >= a.T_out) return;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; e += 4) {
    f4 s4 = *(const f4*)(os + (size_t)tl * LDO + cs + e);
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) s4 += *(const f4*)(os + ((size_t)ww * 64 + tl) * LDO + cs + e);
    v[e] = s4[0]; v[e + 1] = s4[1]; v[e + 2] = s4[2]; v[e + 3] = s4[3];
  }
  row_epilogue(a, v, t, c0 + cs);
}


// ---- variant B: 64 x 64 tile, the 4 waves own 32 x 32 quadrants over the FULL K, operands shared through LDS.
// 8x fewer cache lines touched per flop than the streaming kernel above (whole 512-byte row segments, each fragment
// read by two waves from LDS), and the latency of a dependent global -> LDS -> MFMA step is amortised over a 256-wide K
// slab (64 KB per step, two register sets + two LDS buffers = the next TWO slabs in flight under the current one).
// SLAB = K halfs staged per pipeline stage: 256 (64 KB per buffer, 128 KB in all: ONE workgroup per CU) or 128 (32 KB per buffer,
// 64 KB in all: TWO workgroups per CU -- one's loads and barriers under the other's MFMAs, and 480-workgroup grids (the DiT's FF1
// at T = 934) run in one round instead of two)
template <int SLAB> struct SlabT { h8 a[SLAB / 32], b[SLAB / 32]; };

template <bool WNT, int SLAB>
__global__ __launch_bounds__(256) void gemm_t64_f16_kernel(ConvArgs a) {
  constexpr int LDO = 68;
  constexpr int RB = SLAB * 2;                      // bytes per staged row
  constexpr int PPR = RB / 16;                      // 16-byte pieces per row (32 / 16)
  constexpr int RPP = 256 / PPR;                    // rows per staging pass (8 / 16)
  constexpr int NP = 64 / RPP;                      // passes = vectors per thread per operand (8 / 4)
  constexpr int OPB = 64 * RB;                      // bytes per operand per buffer
  extern __shared__ float os[];                     // 2 x [A: 64 rows x RB][B: 64 rows x RB]; epilogue: [64 t][LDO] fp32
  char* lds = (char*)os;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int vid = xcd_virtual_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int t0 = (vid % gridDim.x) * 64, c0 = (vid / gridDim.x) * 64;
  const _Float16* __restrict__ x = (const _Float16*)a.x;
  const _Float16* __restrict__ w = (const _Float16*)a.w;
  // staging: thread -> 16-byte piece (tid % PPR) of rows (tid / PPR) + RPP i; LDS slot = piece ^ (row & 15) (conflict-free
  // fragment reads for 16 consecutive rows, conflict-free lane-contiguous writes)
  const int sp = tid % PPR, sr = tid / PPR;
  const _Float16* wp[NP];
  const _Float16* xp[NP];
  int woff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = RPP * i + sr;
    wp[i] = w + (long long)min(c0 + row, a.Cout - 1) * a.ldw + 8 * sp;
    xp[i] = x + (long long)min(t0 + row, a.T_in - 1) * a.ldx + 8 * sp;
    woff[i] = row * RB + ((sp ^ (row & 15)) << 4);
  }
  typedef SlabT<SLAB> Slab;
  auto load = [&](Slab& f, int c) {
    const int k0 = c * SLAB;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      f.a[i] = WNT ? __builtin_nontemporal_load((const h8*)(wp[i] + k0)) : *(const h8*)(wp[i] + k0);
      f.b[i] = *(const h8*)(xp[i] + k0);
    }
  };
  auto stage = [&](const Slab& f, int buf) {
    char* base = lds + buf * 2 * OPB;
#pragma unroll
    for (int i = 0; i < NP; ++i) { *(h8*)(base + woff[i]) = f.a[i]; *(h8*)(base + OPB + woff[i]) = f.b[i]; }
  };
  f16v acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const int rowA = wm * 32 + r, rowB = wn * 32 + r;
  auto compute = [&](int buf) {
    const char* pa = lds + buf * 2 * OPB + rowA * RB;
    const char* pb = lds + buf * 2 * OPB + OPB + rowB * RB;
#pragma unroll
    for (int j = 0; j < SLAB / 16; j += 2) {
      const h8 fa0 = *(const h8*)(pa + ((((2 * j + h)) ^ (rowA & 15)) << 4));
      const h8 fb0 = *(const h8*)(pb + ((((2 * j + h)) ^ (rowB & 15)) << 4));
      const h8 fa1 = *(const h8*)(pa + ((((2 * j + 2 + h)) ^ (rowA & 15)) << 4));
      const h8 fb1 = *(const h8*)(pb + ((((2 * j + 2 + h)) ^ (rowB & 15)) << 4));
      mma32(acc0, fa0, fb0);
      mma32(acc1, fa1, fb1);
    }
  };
  const int nslab = a.Cin / SLAB, last = nslab - 1;    // even (Cin % (2 SLAB) == 0, checked by the launcher)
  Slab s0, s1;
  load(s0, 0);
  load(s1, min(1, last));
  for (int c = 0; c < nslab; c += 2) {
    stage(s0, 0);
    GSV_PIN();
    load(s0, min(c + 2, last));
    GSV_PIN();
    __syncthreads();
    compute(0);
    GSV_PIN();
    stage(s1, 1);
    GSV_PIN();
    load(s1, min(c + 3, last));
    GSV_PIN();
    __syncthreads();
    compute(1);
    GSV_PIN();
  }
  __syncthreads();
  // ---- quadrant -> LDS as [t][co] rows, then the same row-segment epilogue as the streaming kernel
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *(f4*)(os + (size_t)(32 * wn + r) * LDO + 32 * wm + 8 * q + 4 * h) =
        (f4){acc0[4 * q] + acc1[4 * q], acc0[4 * q + 1] + acc1[4 * q + 1], acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]};
  __syncthreads();
  const int tl = tid >> 2, cs = (tid & 3) * 16;
  const int t = t0 + tl;
  if (t >= a.T_virt || t >= a.T_out) return;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; e += 4) {
    const f4 s4 = *(const f4*)(os + (size_t)tl * LDO + cs + e);
    v[e] = s4[0]; v[e + 1] = s4[1]; v[e + 2] = s4[2]; v[e + 3] = s4[3];
  }
  row_epilogue(a, v, t, c0 + cs);
}

}  // namespace

// 0 = launched, 1 = not eligible, < 0 error
int launch_gemm_sk(int dtype, const ConvArgs& a, hipStream_t s) {
  static const bool off = getenv("GSV_NO_GEMM_SK") != nullptr;     // A/B switch for profiling
  if (off || dtype != GSV_F16 || a.vt_out || a.rope_cs) return 1;     // fused QKV epilogues live in gemm_lds_kernel
  if (a.taps != 1 || a.stride != 1 || a.ups_u > 0 || a.accumulate || a.pad != 0 || a.Z != 1 || a.pre_act != ACT_NONE) return 1;
  if (a.Cin % 64 != 0 || a.Cin < 256 || a.Cout < 64) return 1;
  if (a.ldx % 8 != 0 || a.ldw % 8 != 0 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return 1;
  if (a.T_in < a.T_virt) return 1;
  // only where the 128 x 128 LDS-tiled kernel cannot fill the chip
  // and only for skinny M: at M = 5760 (AR prefill) the 128 x 128 kernel wins even at 0.7 workgroups per CU
  // (17.7 vs 25.9 us at N = K = 512, 41 vs 53 us at K = 2048; tools/gemm_probe.py)
  const long long tiles128 = (long long)cdiv(a.T_virt, 128) * cdiv(a.Cout, 128);
  // measured in the DiT step (tools/cfm_bench.py, GSV_SK_MAX_TILES): 384 -> 3.16, 160 -> 3.02, 100 -> 3.31 ms per step, i.e. the
  // QKV projection (192 tiles) is better off on the 128 x 128 kernel, FF1 (128 tiles) and the N = 1024 layers are not
  static const int max_tiles = getenv("GSV_SK_MAX_TILES") ? atoi(getenv("GSV_SK_MAX_TILES")) : 160;
  if (tiles128 >= max_tiles || a.T_virt > 2048) return 1;
  static const bool t64 = !(getenv("GSV_GEMM_T64") && getenv("GSV_GEMM_T64")[0] == '0');   // A/B switch
  if (t64 && a.Cin % 512 == 0) {
    // 128-half stages (two workgroups per CU) by default; GSV_T64_SLAB=256 restores round 2's single resident workgroup
    static const int slab = getenv("GSV_T64_SLAB") ? atoi(getenv("GSV_T64_SLAB")) : 128;
    static bool attr64 = false;
    if (!attr64) {
      GSV_HIP(hipFuncSetAttribute((const void*)gemm_t64_f16_kernel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      GSV_HIP(hipFuncSetAttribute((const void*)gemm_t64_f16_kernel<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      GSV_HIP(hipFuncSetAttribute((const void*)gemm_t64_f16_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      GSV_HIP(hipFuncSetAttribute((const void*)gemm_t64_f16_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      attr64 = true;
    }
    dim3 grid64(cdiv(a.T_virt, 64), cdiv(a.Cout, 64));
    if (slab == 128) {
      if (a.w_nt) hipLaunchKernelGGL((gemm_t64_f16_kernel<true, 128>), grid64, dim3(256), 65536, s, a);
      else hipLaunchKernelGGL((gemm_t64_f16_kernel<false, 128>), grid64, dim3(256), 65536, s, a);
    } else {
      if (a.w_nt) hipLaunchKernelGGL((gemm_t64_f16_kernel<true, 256>), grid64, dim3(256), 131072, s, a);
      else hipLaunchKernelGGL((gemm_t64_f16_kernel<false, 256>), grid64, dim3(256), 131072, s, a);
    }
    GSV_HIP(hipGetLastError());
    return 0;
  }
  static bool attr = false;
  const size_t lds = (size_t)4 * 64 * 68 * 4;
  if (!attr) {
    GSV_HIP(hipFuncSetAttribute((const void*)gemm_sk_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  dim3 grid(cdiv(a.T_virt, 64), cdiv(a.Cout, 64));
  hipLaunchKernelGGL(gemm_sk_f16_kernel, grid, dim3(256), lds, s, a);
  GSV_HIP(hipGetLastError());
  return 0;
}

}  // namespace gsv

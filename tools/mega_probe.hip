// Timing skeleton of a persistent decode step (measurement tool, not product code).
//
// Question it answers before t2s_mega.hip is written: what do the PRIMITIVES of a persistent AR decode step
// cost on gfx950 at the real sizes -- (a) an all-gather hop among the 32 workgroups of a group through 8-byte
// {tag, value} granules (sc1 stores, sc1 sweep loads), (b) the per-CU weight stream that runs ahead of the hops
// (one 1 KiB wave-instruction per 64 lanes, non-temporal), (c) the K/V stream by LDS-DMA, (d) the MFMAs.
//
// Geometry = the planned engine: 256 workgroups x 512 threads (one per CU), group = blockIdx % 8 (observed: one XCD),
// member = blockIdx / 8; waves 0-3 sweep the hop granules into LDS ("comm"), waves 4-7 prefetch weights into registers
// one phase ahead, multiply, and publish ("compute").  Per layer four phases with the sizes of
// QKV(+attention) / out-proj / FFN1 / FFN2 at R rows per group.
//
//   hipcc --offload-arch=gfx950 -O3 -o mega_probe tools/mega_probe.hip && ./mega_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// a bare s_barrier: __syncthreads() adds a workgroup-scope release fence = s_waitcnt vmcnt(0), which drains the weight
// prefetch at every barrier (probe 1 and 2 measured exactly that: hops and weight stream ADDED instead of overlapping)
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int GROUPS = 8, MEMBERS = 32, NWG = GROUPS * MEMBERS;
constexpr int NPH = 4;
// weight 16-byte loads per lane per phase (4 compute waves): 96 / 16 / 64 / 64 KB per CU
constexpr int WL[NPH] = {24, 4, 16, 16};
constexpr int WL_LAYER = 24 + 4 + 16 + 16;           // 60 KiB-instructions per wave per layer = 240 KB per CU

struct Args {
  const h8* w;          // [layer][member][wave][WL_LAYER][64] h8
  const h8* kv;         // [layer][wg][KVI][256 lanes] h8  (nt, LDS-DMA)
  gu64* hop;            // [group][4 hops][granules]
  gu32* tmo;            // timeout word
  float* sink;
  int layers, steps, R, kvi;   // kvi: K/V KiB-instructions per compute wave per layer
  int flags;            // 64: plain (not nt) weight loads, 128: __syncthreads instead of the bare barrier; 1: weights, 2: kv, 4: mfma, 8: hops, 16: member-major XCD mapping, 32: K/V issue spread over the phases
};

// granules per hop at R rows: A/C R*512 (fp32 y), B R*256 (fp16 pairs), D R*1024
__device__ __forceinline__ int hop_granules(int ph, int R) { return ph == 0 || ph == 2 ? R * 512 : (ph == 1 ? R * 256 : R * 1024); }

__global__ __launch_bounds__(512, 1) void probe_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // default: group = blockIdx % 8 (a group = one XCD under round-robin placement: every XCD streams ALL weights from
  // the Infinity Cache).  flag 16: the 8 workgroups that read the SAME weight slice (same member, one per group) share
  // an XCD, so a slice is filled into that L2 once and hit 7 times; the hops cross XCDs (they are sc1 anyway).
  int group = blockIdx.x % GROUPS, member = blockIdx.x / GROUPS;
  if (a.flags & 16) { const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8; member = xcd * 4 + (slot >> 3); group = slot & 7; }
  const int wgid = group * MEMBERS + member;
  const bool comm = wave < 4;
  const int cw = wave & 3;
  unsigned* xs = (unsigned*)smem;                      // hop payload image, up to 16 K words (64 KB)
  unsigned char* kvs = smem + 65536;                   // K/V image, up to 80 KB
  volatile int& s_abort = *(volatile int*)(smem + 65536 + 81920);   // all LDS is dynamic (a static would misalign the base)
  if (tid == 0) s_abort = 0;
  __syncthreads();
  const int hop_stride = 16 * 1024;                    // granules per hop buffer (R <= 16)
  gu64* hopg = a.hop + (size_t)group * NPH * hop_stride;
  unsigned epoch = 0;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  h8 wreg[2][24];
  float kvsum = 0.f;
  // prologue: prefetch phase 0 weights of layer 0
  const h8* wbase = a.w + ((size_t)member * 4 + cw) * WL_LAYER * 64 + lane;
  const size_t wlayer = (size_t)MEMBERS * 4 * WL_LAYER * 64;
  auto prefetch = [&](h8* dst, int layer, int ph) {
    int off = 0;
    for (int p = 0; p < ph; ++p) off += WL[p];
    const h8* src = wbase + (size_t)layer * wlayer + (size_t)off * 64;
#pragma unroll
    for (int i = 0; i < 24; ++i)
      if (i < WL[ph]) dst[i] = (a.flags & 64) ? src[(size_t)i * 64] : __builtin_nontemporal_load(src + (size_t)i * 64);
  };
  if (!comm && (a.flags & 1)) prefetch(wreg[0], 0, 0);
  if (!comm && cw == 0 && (a.flags & 8)) {             // first hop of the launch has no producer phase: publish it here
    const int npiece = hop_granules(0, a.R) / MEMBERS;
    for (int g = lane; g < npiece; g += 64)
      __hip_atomic_store(hopg + member * npiece + g, (1ull << 32), RLX_AGENT);
  }
  for (int step = 0; step < a.steps; ++step) {
    for (int layer = 0; layer < a.layers; ++layer) {
      // K/V stream of this layer (consumed in phase 0 of the NEXT layer in the real engine; here: same layer's phase 1 stub)
      if (!comm && (a.flags & 2) && !(a.flags & 32)) {
        const h8* src = a.kv + (((size_t)layer * NWG + wgid) * a.kvi) * 256 + cw * 64 + lane;
        for (int i = 0; i < a.kvi; ++i) {
          const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(kvs) + (unsigned)((i * 4 + cw) * 1024);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 256),
                                           (__attribute__((address_space(3))) void*)(size_t)__builtin_amdgcn_readfirstlane(dst), 16, 0, 2 /*nt*/);
        }
      }
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        ++epoch;
        const int cur = ph & 1, nxt = cur ^ 1;
        const int ng = hop_granules(ph, a.R);
        if (comm) {
          if (a.flags & 8) {
            // sweep: 4 waves x 64 lanes, granule g = it*256 + cw*64 + lane
            bool fail = false;
            for (int it = 0; it * 256 < ng; it += 8) {
              unsigned long long v[8];
              unsigned spins = 0;
              for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  const int g = (it + k) * 256 + cw * 64 + lane;
                  v[k] = g < ng ? __hip_atomic_load(hopg + ph * hop_stride + g, RLX_AGENT) : ((unsigned long long)epoch << 32);
                  ok &= (unsigned)(v[k] >> 32) == epoch;
                }
                if (__all(ok)) break;
                if (++spins > (1u << 20)) { fail = true; break; }
                __builtin_amdgcn_s_sleep(1);
              }
              if (fail) break;
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int g = (it + k) * 256 + cw * 64 + lane;
                if (g < ng) xs[g] = (unsigned)v[k];
              }
            }
            if (fail) { s_abort = 1; if (lane == 0) __hip_atomic_store(a.tmo, epoch, RLX_AGENT); }
          }
        } else {
          // prefetch the next phase's weights before waiting for the hop
          if (a.flags & 1) {
            const int nph = (ph + 1) % NPH;
            const int nlayer = ph == NPH - 1 ? (layer + 1) % a.layers : layer;
            prefetch(wreg[nxt], nlayer, nph);
          }
        }
        if (a.flags & 128) __syncthreads(); else BAR();     // barrier 1: hop payload is in LDS
        if (s_abort) return;
        if (!comm) {
          if (a.flags & 4) {
#pragma unroll
            for (int i = 0; i < 24; ++i)
              if (i < WL[ph]) {
                const h8 b = *(const h8*)(xs + ((i * 64 + lane) & 4095) * 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[cur][i], b, acc, 0, 0, 0);
              }
          } else if (a.flags & 1) {
#pragma unroll
            for (int i = 0; i < 24; ++i)
              if (i < WL[ph]) acc[0] += (float)wreg[cur][i][0];
          }
          if (ph == 0 && (a.flags & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of this layer has landed (drains the weight prefetch too)
          }
        }
        if (a.flags & 128) __syncthreads(); else BAR();     // barrier 2: split-K partials / K/V image complete
        if (!comm) {
          if (ph == 0 && (a.flags & 2)) {
            for (int i = 0; i < a.kvi; ++i) {
              const h8 kk = *(const h8*)(kvs + ((i * 4 + cw) * 64 + lane) * 16);
              kvsum += (float)kk[0] + (float)kk[7];
            }
          }
          // publish this member's piece of the NEXT hop: granules [member * npiece, +npiece)
          if ((a.flags & 8) && cw == 0) {
            const int nh = (ph + 1) % NPH;
            const int npiece = hop_granules(nh, a.R) / MEMBERS;
            const unsigned val = __float_as_uint(acc[0] + kvsum);
            for (int g = lane; g < npiece; g += 64)
              __hip_atomic_store(hopg + nh * hop_stride + member * npiece + g, ((unsigned long long)(epoch + 1) << 32) | val, RLX_AGENT);
          }
          // K/V of the NEXT layer, a quarter per phase (consumed at that layer's phase 0); issued AFTER this phase's publish
          if ((a.flags & 2) && (a.flags & 32)) {
            const int nl = (layer + 1) % a.layers;
            const h8* src = a.kv + (((size_t)nl * NWG + wgid) * a.kvi) * 256 + cw * 64 + lane;
            const int i0 = (a.kvi * ph) / 4, i1 = (a.kvi * (ph + 1)) / 4;
            for (int i = i0; i < i1; ++i) {
              const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(kvs) + (unsigned)((i * 4 + cw) * 1024);
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 256),
                                               (__attribute__((address_space(3))) void*)(size_t)__builtin_amdgcn_readfirstlane(dst), 16, 0, 2 /*nt*/);
            }
          }
        }
      }
    }
  }
  if (!comm) a.sink[blockIdx.x * 256 + cw * 64 + lane] = acc[0] + acc[1] + kvsum;
}

int main(int argc, char** argv) {
  const int layers = 24, steps = argc > 1 ? atoi(argv[1]) : 20;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const size_t wbytes = (size_t)layers * MEMBERS * 4 * WL_LAYER * 64 * 16;
  const int kvi_max = 20;
  const size_t kvbytes = (size_t)layers * NWG * kvi_max * 256 * 16;
  void *w, *kv, *hop, *tmo, *sink;
  CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&kv, kvbytes));
  CK(hipMemset(w, 0, wbytes)); CK(hipMemset(kv, 0, kvbytes));
  const size_t hopbytes = (size_t)GROUPS * NPH * 16 * 1024 * 8;
  CK(hipMalloc(&hop, hopbytes)); CK(hipMalloc(&tmo, 16)); CK(hipMalloc(&sink, NWG * 256 * 4));
  printf("weights %.1f MB, kv %.1f MB\n", wbytes / 1e6, kvbytes / 1e6);
  CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { const char* name; int flags, R, kvi; };
  const Cfg cfgs[] = {
      {"[shared] weights only", 1 | 16, 4, 0},
      {"[shared] weights only, plain loads", 1 | 16 | 64, 4, 0},
      {"[shared] hops only R=4", 8 | 16, 4, 0},
      {"[shared] hops+w+mfma R=4", 13 | 16, 4, 0},
      {"[shared] hops+w+mfma R=4 plain", 13 | 16 | 64, 4, 0},
      {"[shared] hops+w+mfma R=4 __syncthreads", 13 | 16 | 128, 4, 0},
      {"[shared] hops+w+mfma+kv72 R=4", 15 | 16, 4, 18},
      {"[shared] hops+w+mfma+kv72 spread R=4", 15 | 16 | 32, 4, 18},
      {"[shared] hops+w+mfma+kv72 spread R=4 plain", 15 | 16 | 32 | 64, 4, 18},
      {"[shared] hops+w+mfma+kv72 spread R=1", 15 | 16 | 32, 1, 18},
      {"[group=xcd] weights only", 1, 4, 0},
      {"[group=xcd] hops only R=4", 8, 4, 0},
      {"[group=xcd] hops+w+mfma R=4", 13, 4, 0},
      {"[group=xcd] hops+w+mfma+kv72 spread R=4", 15 | 32, 4, 18},
      {"[group=xcd] hops+w+mfma+kv72 spread R=4 plain", 15 | 32 | 64, 4, 18},
  };

  for (const Cfg& c : cfgs) {
    float best = 1e9f;
    unsigned tm = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(hop, 0, hopbytes)); CK(hipMemset(tmo, 0, 16));
      Args a{(const h8*)w, (const h8*)kv, (gu64*)hop, (gu32*)tmo, (float*)sink, layers, steps, c.R, c.kvi, c.flags};
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe_kernel, dim3(NWG), dim3(512), 150 * 1024, 0, a);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(&tm, tmo, 4, hipMemcpyDeviceToHost));
      if (tm) break;
    }
    printf("%-44s %8.3f ms/launch  %7.2f us/layer  %7.3f us/phase  %6.1f us/step%s\n", c.name, best, best * 1e3 / (steps * layers),
           best * 1e3 / (steps * layers * NPH), best * 1e3 / steps, tm ? "  TIMEOUT" : "");
    fflush(stdout);
  }
  return 0;
}

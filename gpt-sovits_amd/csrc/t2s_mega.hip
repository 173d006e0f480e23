// Persistent AR decode engine for gfx950: every step of a decode call (reference AR/models/t2s_model.py:694-769: the
// `for idx in range(1500)` loop of infer_panel_batch_infer, 24 x T2SBlock.decode_next_token :176-221, ar_predict_layer,
// sample :AR/models/utils.py:140-199, EOS / early-stop bookkeeping) runs inside ONE kernel launch.
//
// Why: as five launches per layer the step is a chain of 122 dependent kernels of 4.6-7 us that each move 0.5-18 MB
// (round 1: 0.67 ms per step at B = 32, 11 % of the HBM roofline); nothing in it is bandwidth-bound, every kernel spends
// its life ramping up and waiting for its first bytes.  Here the chain's hand-offs stay on the chip and the weights are
// already in registers when a hand-off completes.
//
// Decomposition (measured first with tools/mega_probe.hip, profiles/r02_mega_probe.txt):
//   * rows are independent, so the batch is split into 8 GROUPS of R = ceil(B/8) rows (row b -> group b % 8); a group
//     is served by 32 workgroups (one per CU, 512 threads) that never talk to another group: no grid-wide barrier.
//   * inside a group the layer is tensor-parallel over the 32 members: member j = (head j/2, row parity j%2) computes
//     q,k,v of its head for its rows, appends K/V, attends (all local), then the out-projection, FFN1 and FFN2 are
//     split over output columns (16 / 64 / 16 columns per member).  Between phases the members all-gather the small
//     activation vector (R x 512 fp32, R x 512 fp16, R x 2048 fp16): 4 hops per layer.
//   * a hop = 8-byte {tag = epoch, value} granules written with sc1 (write-through) stores and swept with sc1 loads
//     until every tag matches (MI355X_MICROARCH.md, visibility: R2 "the data is the flag"; no fence, no flag, placement
//     independent).  Every spin is bounded; a timeout sets an error word and every workgroup leaves.
//   * waves 0-3 of a workgroup ("comm") sweep hops into LDS and do the LayerNorms; waves 4-7 ("compute") hold the
//     member's weight slices in REGISTERS, prefetched one phase ahead with perfectly coalesced 1 KiB wave loads from a
//     copy of the weights packed in exactly that order, run the MFMAs (K split over the 4 waves in the same 128-wide
//     chunks and summation order as dec_gemm_kernel), reduce through LDS and publish.
//   * K/V of the member's (head, rows) is prefetched into LDS by LDS-DMA (non-temporal) one layer ahead.
//   * blockIdx -> (group, member): the 8 workgroups that read the SAME weight slice (one per group) are placed on the
//     same XCD (blockIdx % 8 under the observed round-robin placement), so a slice is filled into that L2 once and hit
//     seven times; the hops are sc1 and do not care.  Correctness never depends on the placement.
//   * logits are split over the members too; row r of a group is sampled by member r (one wave, sample_core of
//     t2s_sample.h), which also emits the next input embedding: the step's tail is two more hops.
//   * (round 3) more than 32 rows: a group serves up to four QUADS of rows, phase by phase one after the other with the weight
//     slices it already holds; the comm waves sweep quad i + 1 into a second copy of the operand image while the compute
//     waves work on quad i (comm_role_pipe / compute_role_pipe, MODE 2).  The step is latency-bound, so rows are added in
//     time, not in tile width: 128 rows cost 0.79 ms per step against 4 x 0.30.
//   * weight slices are loaded with the DEFAULT cache policy (the 8 same-slice workgroups of an XCD share one L2 fill), K/V
//     non-temporal (read once per step; keeps the weights in the Infinity Cache).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "t2s_mega.h"

namespace gsv {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// Workgroup barrier WITHOUT __syncthreads()'s release fence: that fence is an s_waitcnt vmcnt(0), which would drain the
// weight prefetch (issued one phase ahead on purpose) at every barrier -- tools/mega_probe.hip measured hops and weight
// stream adding up instead of overlapping with it.  LDS traffic of this wave is complete (lgkmcnt(0)) before the barrier;
// the "memory" clobber keeps the compiler from moving memory operations across it.
#define MG_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The (sticky) abort flag is tested ONCE per layer: read at the top of the layer, acted on behind the layer's first barrier.
// Round 2 tested it behind every barrier -- 14 LDS round trips of ~100 cycles per layer on the critical path; read right in
// front of the barrier instead, the round trip delays the barrier arrival (measured: -1 % at 32 rows, +1.7 % at 128).  A wave
// that has failed has set the error word and returned (a finished wave no longer counts at s_barrier); its workgroup follows
// within a layer, the other workgroups through their bounded hop waits, and the launch is re-run from the snapshot.
#define MG_BAR_AL(q) do { MG_BAR(); if (ab_l) return; } while (0)

namespace {

constexpr int MG_GROUPS = 8, MG_MEMBERS = 32, MG_NWG = MG_GROUPS * MG_MEMBERS, MG_THREADS = 512;
constexpr int RMAX = 4;                       // rows of a group served together: one QUAD (B <= 32: the group's only quad)
constexpr int QMAX = 4;                       // quads per group (B <= 128): phase by phase one quad after the other, see t2s_mega_kernel
constexpr int D = 512, NH = 16, HD = 32, FF = 2048;
constexpr int XS_LD = 528, HS_LD = 2064;      // LDS row strides in halfs (rows 32 B apart mod 256 B: conflict-free b128 reads)
constexpr int KV_CAP = 320;                   // cached positions per own row held in the LDS image (rest: global loads)
constexpr int VPAD = 1088;                    // logits row stride in the hop buffer (17 x 64)
constexpr unsigned SPIN_MAX = 1u << 18;

// packed weight geometry (KiB-instructions per compute wave)
constexpr int WI_P1 = 24, WI_P2 = 4, WI_P3 = 16, WI_P4 = 16, WI_LG = 12;
constexpr size_t P1_HALFS = (size_t)NH * 4 * WI_P1 * 512;                       // per layer, by head
constexpr size_t PM_HALFS = (size_t)MG_MEMBERS * 4 * (WI_P2 + WI_P3 + WI_P4) * 512;  // per layer, by member
constexpr size_t LAYER_HALFS = P1_HALFS + PM_HALFS;
constexpr size_t LOGIT_HALFS = (size_t)MG_MEMBERS * 4 * WI_LG * 512;

// hop buffers of one group, in granules
constexpr int HOP_A = 0;                            // [RMAX][512] fp32 y  + RMAX state granules
constexpr int HOP_ST = RMAX * 512;
constexpr int HOP_B = HOP_ST + 64;                  // [RMAX][256] half pairs (attention output)
constexpr int HOP_C = HOP_B + RMAX * 256;           // [RMAX][512] fp32 y1
constexpr int HOP_D = HOP_C + RMAX * 512;           // [RMAX][1024] half pairs (FFN hidden)
constexpr int HOP_E = HOP_D + RMAX * 1024;          // [RMAX][VPAD] fp32 logits
constexpr int HOP_GROUP = HOP_E + RMAX * VPAD;     // one quad's buffers
constexpr int HOP_GROUP_ALL = QMAX * HOP_GROUP;    // a group's buffers: quad after quad

// LDS carve (bytes); everything dynamic so the base stays 16-byte aligned
constexpr int L_XRES = 0;                                     // fp32 [RMAX][512]   LayerNorm output (residual operand); one quad per group
constexpr int L_RS = 0;                                       // f4 [2][QMAX][64]   the same 8 KB in multi-quad launches: residual operands kept per quad
constexpr int L_XS = L_XRES + RMAX * 512 * 4;                 // half [2][RMAX][XS_LD] LayerNorm output (MFMA operand); second copy: pipelined quads
constexpr int L_AT = L_XS + 2 * RMAX * XS_LD * 2;             // half [2][RMAX][XS_LD] attention output of all heads
constexpr int L_HS = L_AT + 2 * RMAX * XS_LD * 2;             // half [2][RMAX][HS_LD] FFN hidden
constexpr int L_RED = L_HS + 2 * RMAX * HS_LD * 2;            // f4 [24][16] split-K partials: the RMAX valid batch rows of each 16 x 16 tile
constexpr int L_QKV = L_RED + 24 * 16 * 16;                   // half [2][3][32] q,k,v of the own rows
constexpr int L_ATT = L_QKV + 2 * 3 * 32 * 2;                 // float [8] m + [8 waves][4 rows of 16 lanes][36] acc,l partials
constexpr int L_STAGE = L_ATT + 32 + 8 * 4 * 36 * 4;          // 4 x 1 KB: per-wave transposition buffers of the publishers
constexpr int L_ST = L_STAGE + 4 * 1024;                      // int: active[QMAX * RMAX], kvlen[..], step[..], abort, -, barrier counter, flag
constexpr int ST_N = QMAX * RMAX;                             // rows of a group
constexpr int L_SEEN = L_ST + (3 * ST_N + 16) * 4;            // bytes [VPAD]
constexpr int L_KV = (L_SEEN + VPAD + 63) & ~63;              // [2 rows][K|V][KV_CAP][64 B]
constexpr int L_TOTAL = L_KV + 2 * 2 * KV_CAP * 64;
static_assert(L_TOTAL <= 160 * 1024, "LDS budget");
static_assert(L_STAGE % 16 == 0 && L_XS % 16 == 0 && L_AT % 16 == 0 && L_HS % 16 == 0 && L_RED % 16 == 0 && L_QKV % 16 == 0 && L_KV % 16 == 0, "LDS alignment");

__device__ __forceinline__ void gstore(gu64* p, unsigned tag, unsigned val) {
  __hip_atomic_store(p, ((u64)tag << 32) | val, RLX_AGENT);           // global_store_dwordx2 ... sc1 (one untorn granule)
}
__device__ __forceinline__ u64 gload(gu64* p) { return __hip_atomic_load(p, RLX_AGENT); }   // global_load_dwordx2 ... sc1

// Cross-lane reductions on the VALU's data-parallel primitives (DPP) instead of __shfl_xor: hipcc lowers a shuffle to
// ds_bpermute_b32, an LDS-crossbar round trip of ~100 cycles, and the reductions here are dependent chains of 6 (wave sum /
// max) to 36 (attention) of them per layer.  quad_perm swaps inside quads, row_half_mirror / row_mirror fold 8 / 16 lanes
// (a sum or max does not care that a mirror is not an xor), the four 16-lane rows are folded through v_readlane.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
__device__ __forceinline__ float rl_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f<DPP_QUAD_XOR1>(v);
  v += dpp_f<DPP_QUAD_XOR2>(v);
  v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_f<DPP_ROW_MIRROR>(v);
  return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
  return fmaxf(fmaxf(rl_f(v, 0), rl_f(v, 16)), fmaxf(rl_f(v, 32), rl_f(v, 48)));
}

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const h2 v = (h2){(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}

// LDS-DMA of one KiB (16 B per lane, lane i -> lds_dst + 16 i), non-temporal; invisible to the compiler's vmcnt
// bookkeeping (cdna_hip_programming.md 5.7): the consumer waits with asm vmcnt(0) + a workgroup barrier
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// fp32 parameters of one layer in the packed buffer (MegaArgs::fpack): offsets in floats
constexpr int FP_QKVB = 0, FP_OUTB = 1536, FP_B1 = 2048, FP_B2 = 4096, FP_N1W = 4608, FP_N1B = 5120, FP_N2W = 5632, FP_N2B = 6144,
              FP_LAYER = 6656;

struct Ctx {
  unsigned char* smem;
  int lane, wave, cw, tid_c;
  bool comm;
  int group, member, head, half, R;   // R = rows of the CURRENT quad
  int qd, Rtot;                       // current quad; rows of the group (all quads)
  gu64* hop;                    // this group's hop buffers of the current (step, layer) ring slot
  gu64* hop_base;               // slot 0
  gu32* err;
  int hint_miss16;              // hint phase ends when at most this many 16ths of the polled lines are still missing
  int hint_pipe;                // two hint polls in flight (GSV_MEGA_HINT bit 4)
  int hint_spread;              // each member polls a different publisher's line
  unsigned long long* prof;     // this wave's 32 stamp slots or null
  bool prof_on;
};

// in-kernel stamp (measurement runs: GSV_MEGA_PROF=<file>): 100 MHz real-time counter at point i of the profiled layer
#define MG_STAMP(q, i) do { if ((q).prof_on && (q).lane == 0) (q).prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)

// The row state lives in LDS and is addressed through LDS-typed pointers: through a generic pointer the volatile abort flag
// became a FLAT load, and a flat load is waited for with vmcnt(0) -- which drained the whole weight prefetch after every
// barrier (seen in the ISA; the dynamic LDS segment starts at LDS address 0: there is no static LDS in this kernel).
typedef __attribute__((address_space(3))) int lds_int;
typedef volatile __attribute__((address_space(3))) int lds_vint;
// the accessors return the CURRENT quad's four entries (local row lr = 4 * quad + r)
__device__ __forceinline__ lds_int* st_active(const Ctx& c) { return (lds_int*)(unsigned)L_ST + RMAX * c.qd; }
__device__ __forceinline__ lds_int* st_kvlen(const Ctx& c) { return (lds_int*)(unsigned)L_ST + ST_N + RMAX * c.qd; }
__device__ __forceinline__ lds_int* st_step(const Ctx& c) { return (lds_int*)(unsigned)L_ST + 2 * ST_N + RMAX * c.qd; }
__device__ __forceinline__ lds_vint* st_abort(const Ctx&) { return (lds_vint*)(unsigned)L_ST + 3 * ST_N; }
__device__ __forceinline__ lds_int* st_cbar(const Ctx&) { return (lds_int*)(unsigned)L_ST + 3 * ST_N + 2; }

// Barrier among the 4 compute waves only (an s_barrier would also wait for the comm waves, which are busy issuing the next
// layer's K/V loads during P1): an arrival counter in LDS; LDS operations of a wave complete in order, so the add is behind
// the wave's earlier LDS writes.  `gen` counts arrivals expected so far (same in every compute wave).
__device__ __noinline__ void mega_fail(gu32* err, int lane, unsigned epoch, unsigned code);
__device__ __forceinline__ bool compute_barrier(const Ctx& q, int& gen) {
  gen += 4;
  if (q.lane == 0) __hip_atomic_fetch_add(st_cbar(q), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  int ab = *st_abort(q);           // returned to the caller: read in front of the spin, not behind it
  while (*(lds_vint*)st_cbar(q) < gen && !(ab = *st_abort(q)) && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(0);
  // a wave that gives up must not go on to reduce partial sums another wave has not written: like every other bounded spin
  // of the engine it sets the abort flag and the error word (code 0x40), and the caller returns at its abort test
  if (spins >= (1u << 24)) { mega_fail(q.err, q.lane, (unsigned)gen, 0x40u); ab = 1; }
  asm volatile("" ::: "memory");
  return ab != 0;
}

// by value: a `const Ctx&` parameter of a non-inlined function forces the whole Ctx into scratch memory, re-stored at the
// start of every phase (30 scratch instructions in the hot loops, seen in the ISA)
__device__ __noinline__ void mega_fail(gu32* err, int lane, unsigned epoch, unsigned code) {
  lds_vint* ab = (lds_vint*)(unsigned)L_ST + 3 * ST_N;                  // st_abort
  const int was = *ab;
  *ab = 1;
  // the FIRST failure of a workgroup is the one reported: its waves run on to the end of the layer (abort is tested once per
  // layer), and every bounded wait they still meet gives up at once
  if (lane == 0 && !was) {
    __hip_atomic_store(err + 1, epoch, RLX_AGENT);
    __hip_atomic_store(err + 2, (unsigned)blockIdx.x, RLX_AGENT);
    __hip_atomic_store(err + 3, code, RLX_AGENT);
    __hip_atomic_store(err, 1u, RLX_AGENT);
  }
}

// one wave re-reads N granules per lane (granule k*64 + lane) of ONE or TWO rows (g1 may be null) until every tag equals `epoch`
template <int N>
__device__ __forceinline__ bool sweep2(const Ctx& c, gu64* g0, gu64* g1, int nvalid, unsigned epoch, unsigned (&v0)[N],
                                       unsigned (&v1)[N], unsigned code, bool hint = true) {
  // Every producer writes whole 128-B lines (16 granules) with one store instruction, so the LAST granule of each line is
  // polled first: one load per lane and row covers 64 lines, a 16th of a full pass.  When most lines are there the rows are
  // read in full -- and still checked tag by tag (the hint is an optimisation: the data remains its own flag).  Full-row
  // polling by 4 waves x 256 CUs was ~8 TB/s of fabric traffic by itself.
  if (hint) {
    const int nlines = (nvalid + 15) >> 4;
    constexpr int NP = (N + 15) / 16;                     // hint loads per lane and row (64 lines each)
    // Two polls are kept in flight, half a round trip apart: a poll costs ~1 us of sc1 latency, so one at a time notices an
    // arrival on average half a microsecond late; c.hint_pipe = 0 restores the single poll (A/B switch GSV_MEGA_HINT bit 4).
    u64 pa[2 * NP], pb[2 * NP];
    auto issue = [&](u64 (&h)[2 * NP]) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int ln = j * 64 + c.lane;
        const int i = min(min(ln, nlines - 1) * 16 + 15, nvalid - 1);
        h[2 * j] = gload(g0 + i);
        h[2 * j + 1] = g1 ? gload(g1 + i) : ((u64)epoch << 32);
      }
    };
    auto landed = [&](const u64 (&h)[2 * NP]) {
      int miss = 0;
#pragma unroll
      for (int j = 0; j < NP; ++j)
        miss += ((unsigned)(h[2 * j] >> 32) != epoch) + ((unsigned)(h[2 * j + 1] >> 32) != epoch);
      // most lines there: the stragglers are at most one poll away, go on with full passes (one round trip fewer at the end)
      return __popcll(__ballot(miss != 0)) * 16 <= nlines * c.hint_miss16;
    };
    issue(pa);
    if (c.hint_pipe) __builtin_amdgcn_s_sleep(16);        // ~0.4 us: half a poll round trip
    for (unsigned spins = 0;; spins += 2) {
      if (c.hint_pipe) {
        issue(pb);
        if (landed(pa)) break;
        issue(pa);
        if (landed(pb)) break;
      } else {
        if (landed(pa)) break;
        if (spins < 16) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4);
        issue(pa);
      }
      if (spins > SPIN_MAX || *st_abort(c) || (spins & 1022u) == 1022u && __hip_atomic_load(c.err, RLX_AGENT) != 0u) {
        mega_fail(c.err, c.lane, epoch, code | 0x100u);
        return false;
      }
    }
  }
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int i = k * 64 + c.lane;
      v0[k] = 0u; v1[k] = 0u;
      if (i < nvalid) {
        const u64 x = gload(g0 + i);
        v0[k] = (unsigned)x;
        ok &= (unsigned)(x >> 32) == epoch;
        if (g1) {
          const u64 y = gload(g1 + i);
          v1[k] = (unsigned)y;
          ok &= (unsigned)(y >> 32) == epoch;
        }
      }
    }
    if (__all(ok)) return true;
    if (spins > SPIN_MAX || *st_abort(c) || (spins & 1023u) == 1023u && __hip_atomic_load(c.err, RLX_AGENT) != 0u) {
      mega_fail(c.err, c.lane, epoch, code);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

template <int N>
__device__ __forceinline__ bool sweep(const Ctx& c, gu64* g, int nvalid, unsigned epoch, unsigned (&v)[N], unsigned code,
                                      bool hint = true) {
  unsigned dummy[N];
  return sweep2<N>(c, g, nullptr, nvalid, epoch, v, dummy, code, hint);
}

// A row's payload is read with 16-BYTE loads: NQ requests per lane instead of 2 NQ.  Lane m,
// piece j holds granules j * 128 + 2 m and + 1 (dwords: value, tag, value, tag); each 8-byte half is still one untorn granule
// with its own tag.  Measured on hop D alone (8 KB per row): 569 -> 514 us per step -- the sc1 path is bound by REQUESTS, an
// 8-byte-per-lane load moves 512 B per wave instruction.  The loads and their wait are ONE asm statement (the compiler never
// sees a half-loaded register); it waits for everything the wave has in flight, which at a hop is nothing else.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NQ> __device__ __forceinline__ void wide_loads(const gu64* p0, u4v (&q)[NQ]);
template <> __device__ __forceinline__ void wide_loads<2>(const gu64* p0, u4v (&q)[2]) {
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(q[0]), "=&v"(q[1]) : "v"(p0) : "memory");
}
template <> __device__ __forceinline__ void wide_loads<4>(const gu64* p0, u4v (&q)[4]) {
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
               "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(p0) : "memory");
}
template <> __device__ __forceinline__ void wide_loads<8>(const gu64* p0, u4v (&q)[8]) {
  const gu64* p1 = p0 + 512;
  asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
               "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
               "global_load_dwordx4 %4, %9, off sc1\n\tglobal_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
               "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
               : "v"(p0), "v"(p1) : "memory");
}
// The hint line of a member (GSV_MEGA_HINT bit 7, the default): the last line of publisher (member + 16) % 32 -- ANOTHER
// member's line, and a different one for every member of the group.  With every member polling the same line (round 2: the
// last line of the row) 32 workgroups re-read one address while its publisher was trying to write it: 297 -> 288 us per step
// when the polls are spread; two or four lines per member, polls in flight, slower paces all lose (profiles/r03_ab_hint_spread.txt).
#ifndef GSV_HINT_OFFSET
#define GSV_HINT_OFFSET 16          // whose line a member polls: member + this.  A publisher on ANOTHER XCD (members 4 x .. 4 x + 3 run
                                    // on XCD x): +1 / -1 cost 8 %, +4 / +8 / +16 are equal (profiles/r03_ab_hint_spread.txt, 11.)
#endif
template <int NLINES>
__device__ __forceinline__ int hint_line(const Ctx& c) {
  return ((c.member + GSV_HINT_OFFSET) & 31) * NLINES / 32 + (NLINES >= 64 ? 1 : 0);
}
// pauses (s_sleep units of 64 clocks) between hint polls and between full passes of sweep_wide (compile-time: A/B by library)
#ifndef GSV_HINT_SLEEP
#define GSV_HINT_SLEEP 1
#endif
#ifndef GSV_PASS_SLEEP
#define GSV_PASS_SLEEP 2
#endif
// one wave, one row of NQ * 128 granules
template <int NQ>
__device__ __forceinline__ bool sweep_wide(const Ctx& c, gu64* g, unsigned epoch, u4v (&q)[NQ], unsigned code, bool hint) {
  constexpr int NLINES = NQ * 8;                       // 16 granules per 128-byte line
  if (hint) {
    // ONE line per row is polled before the full pass (round 2 found every extra poll to cost: every 2nd, 4th, ... line, two
    // polls in flight, one polling wave per workgroup were all measured and are gone from this loop)
    const int ln = c.hint_spread ? hint_line<NLINES>(c) : NLINES - 1;
    gu64* hp = g + ln * 16 + 15;
    for (unsigned spins = 0;; ++spins) {
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)(gload(hp) >> 32)) == epoch) break;     // one address for the wave
      if (spins > SPIN_MAX || *st_abort(c) || (spins & 1023u) == 1023u && __hip_atomic_load(c.err, RLX_AGENT) != 0u) {
        mega_fail(c.err, c.lane, epoch, code | 0x100u);
        return false;
      }
      if (spins < 8) __builtin_amdgcn_s_sleep(GSV_HINT_SLEEP); else __builtin_amdgcn_s_sleep(4);
    }
  }
  const gu64* p0 = g + 2 * c.lane;
  for (unsigned spins = 0;; ++spins) {
    wide_loads<NQ>(p0, q);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NQ; ++j) ok &= q[j][1] == epoch && q[j][3] == epoch;
    if (__all(ok)) return true;
    if (spins > SPIN_MAX || *st_abort(c) || (spins & 1023u) == 1023u && __hip_atomic_load(c.err, RLX_AGENT) != 0u) {
      mega_fail(c.err, c.lane, epoch, code);
      return false;
    }
    __builtin_amdgcn_s_sleep(GSV_PASS_SLEEP);
  }
}

// LayerNorm of one row delivered by sweep_wide<4>: lane m holds elements j * 128 + 2 m + {0, 1}; gm / bt in the same order;
// gamma == null: identity (layer 0's input is the embedding itself)
__device__ __forceinline__ void ln_row_wide(const Ctx& c, int row, const u4v (&v)[4], const float* gm, const float* bt) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { x[2 * j] = __uint_as_float(v[j][0]); x[2 * j + 1] = __uint_as_float(v[j][2]); }
  if (gm) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    const float mean = wave_sum_dpp(s) * (1.f / D);
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float dl = x[k] - mean; qq += dl * dl; }
    const float rstd = rsqrtf(wave_sum_dpp(qq) * (1.f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (x[k] - mean) * rstd * gm[k] + bt[k];
  }
  float* xr = (float*)(c.smem + L_XRES) + row * D + 2 * c.lane;
  _Float16* xs = (_Float16*)(c.smem + L_XS) + row * XS_LD + 2 * c.lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *(f2*)(xr + j * 128) = (f2){x[2 * j], x[2 * j + 1]};
    *(h2*)(xs + j * 128) = (h2){(_Float16)x[2 * j], (_Float16)x[2 * j + 1]};
  }
}

// compute waves: NT tiles x 4 k-steps of this wave's 128-wide K chunk `kc` against the activation image `act`
// (half [RMAX][ld]); partial tiles go to red[(slot0 + t)][lane]
// a parked tile: lanes with (lane & 15) < RMAX, i.e. 4 row-lanes x 4 column groups = 16 f4 per tile
__device__ __forceinline__ int red_idx(int slot, int lane) { return slot * 16 + (lane >> 4) * 4 + (lane & 15); }

template <int NT>
__device__ __forceinline__ void gemm_chunk(const Ctx& c, const h8 (&w)[NT * 4], const _Float16* act, int ld, int kc, int slot0) {
  const int rowl = c.lane & 15, kg = c.lane >> 4;
  const _Float16* bp = act + (rowl & (RMAX - 1)) * ld + kc * 128 + 8 * kg;
  h8 b[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b[ks] = *(const h8*)(bp + 32 * ks);
  f4* red = (f4*)(c.smem + L_RED);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t * 4 + ks], b[ks], acc, 0, 0, 0);
    if (rowl < RMAX) red[red_idx(slot0 + t, c.lane)] = acc;     // only the tile's first RMAX batch rows hold rows of the quad
  }
}

// ONE 16-column tile over the whole K = 512 by one wave: FFN1's four tiles are split over the compute waves by COLUMN, so
// no partial sums meet in LDS (the split-K form cost a barrier + an LDS round trip + two idle waves per layer).  Four chains,
// one per 128-wide K chunk, summed ((c0 + c1) + c2) + c3: the order of the split-K reduce this replaces, bit for bit.  The
// operand fragments of k-step ks + 1 are in flight under the four MFMAs of k-step ks.
__device__ __forceinline__ f4 gemm_tile_k512(const Ctx& c, const h8 (&w)[16], const _Float16* act, int ld) {
  const int rowl = c.lane & 15, kg = c.lane >> 4;
  const _Float16* bp = act + (rowl & (RMAX - 1)) * ld + 8 * kg;
  f4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 bA[4], bB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bA[j] = *(const h8*)(bp + 128 * j);
#pragma unroll
  for (int ks = 0; ks < 4; ks += 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bB[j] = *(const h8*)(bp + 128 * j + 32 * (ks + 1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 * j + ks], bA[j], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 2 < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bA[j] = *(const h8*)(bp + 128 * j + 32 * (ks + 2));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 * j + ks + 1], bB[j], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  f4 v = acc[0];
  v += acc[1]; v += acc[2]; v += acc[3];
  return v;
}

template <int N>
__device__ __forceinline__ void wload(h8 (&dst)[N], const h8* src) {
#pragma unroll
  // DEFAULT cache policy, not non-temporal: the 8 workgroups of an XCD that stream the same slice (one per row group) then
  // share one fill of the XCD's L2; with `nt` loads the line was kept only when the eight requests happened to arrive close
  // together -- that was the two run-to-run modes (~319 / ~343 us per step), and the memory-side traffic swinging between
  // 1.28 and 1.79 x the algorithmic bytes.  Measured, 5 alternating pairs: 334-345 -> 301-303 us per step at B = 32,
  // 838 -> 790 at B = 128 (-DGSV_MEGA_W_NT restores the non-temporal loads).
#ifdef GSV_MEGA_W_NT
  for (int i = 0; i < N; ++i) dst[i] = __builtin_nontemporal_load(src + (size_t)i * 64);
#else
  for (int i = 0; i < N; ++i) dst[i] = src[(size_t)i * 64];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The two roles are separate functions: a weight array defined under `if (!comm)` inside one body would have to keep
// its old value alive around the whole loop for the other role's lanes, i.e. all 240 weight registers live everywhere.
// Both roles execute the same sequence of workgroup barriers and take the same (LDS-uniform) exit decisions.
// ---------------------------------------------------------------------------------------------------------------
// Make the role's indices opaque to the optimiser at the start of every phase.  Without this, loop-invariant code motion
// hoists every per-lane address (weight, bias, hop, LDS pointers of all phases) out of the step and layer loops and keeps
// ~100 of them alive across the whole body: VGPR / SGPR spills in the critical path.  Recomputing an address is 2-3 ALU ops.
__device__ __forceinline__ void relaunder(Ctx& q) {
  int cw = __builtin_amdgcn_readfirstlane(q.cw), member = __builtin_amdgcn_readfirstlane(q.member),
      group = __builtin_amdgcn_readfirstlane(q.group);
  asm volatile("" : "+v"(q.lane), "+s"(cw), "+s"(member), "+s"(group));
  q.cw = cw; q.member = member; q.group = group; q.head = member >> 1; q.half = member & 1;
  q.tid_c = q.cw * 64 + q.lane;
}

// ONE set of hop buffers: the epoch in every granule tells (step, layer) apart.  Round 2 tried a ring of 16 sets used
// round-robin over the layers, so that the payload could be read with plain loads that allocate in L2 (the 4 members of a group
// on one XCD would fetch a line from the memory side once instead of four times).  MEASURED on MI355X: 16 sets alone cost 4 %
// (the one 0.9 MB set stays resident on the memory side, 14 MB of rotating lines do not), the L2-shared read gained nothing on
// top, and the run-time modulo this function then needed in every phase cost more; the option is gone (round 3).
__device__ __forceinline__ gu64* hop_slot(const Ctx& c, int s, int l) {
  (void)s; (void)l;
  return c.hop_base + (size_t)c.qd * HOP_GROUP;
}

__device__ __forceinline__ bool group_done(const Ctx& c) {
  int any = 0;
  for (int r = 0; r < c.Rtot; ++r) any |= ((lds_int*)(unsigned)L_ST)[r];      // every quad's rows
  return !any;
}
// make quad `qd` the current one: its rows, its state entries (through c.qd), and -- via hop_slot -- its hop buffers
__device__ __forceinline__ void set_quad(Ctx& c, int qd) {
  c.qd = __builtin_amdgcn_readfirstlane(qd);
  c.R = __builtin_amdgcn_readfirstlane(min(RMAX, c.Rtot - RMAX * qd));
}
// batch row of the current quad's local row r
__device__ __forceinline__ int batch_row(const Ctx& c, int r) { return c.group + MG_GROUPS * (RMAX * c.qd + r); }

// Attention of the member's own rows (H4: softmax(q K^T / sqrt(32)) V over the cached keys + this step's key): row ro is
// served by NW waves, its keys dealt to them in 16-key groups, 4 lanes per key (8 dims each).  Two passes over the
// LDS image (scores -> lane maximum -> weights), so there is no per-key rescaling; v_dot2_f32_f16 for q . k.
template <int NW>     // waves per row; aw = ro * NW + hq
__device__ __forceinline__ void attention_part(const MegaArgs& a, const Ctx& q, int l, int aw) {
  unsigned char* smem = q.smem;
  const int ro = aw / NW, hq = aw % NW, r = 2 * ro + q.half;
  const bool rv = r < q.R && st_active(q)[r];
  const int n_old = rv ? st_kvlen(q)[r] : 0;
  const int part = q.lane & 3, slot = q.lane >> 2;
  const float scale = 0.17677669529663687f;            // 1 / sqrt(32)
  const _Float16* qkv_s = (const _Float16*)(smem + L_QKV) + ro * 3 * HD;
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const h8 qv = *(const h8*)(qkv_s + part * 8);
  auto score = [&](const h8& kk) {
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      sc = __builtin_amdgcn_fdot2((h2v){qv[2 * i], qv[2 * i + 1]}, (h2v){kk[2 * i], kk[2 * i + 1]}, sc, false);
    sc += dpp_f<DPP_QUAD_XOR1>(sc);            // the 4 lanes of a key are one quad
    sc += dpp_f<DPP_QUAD_XOR2>(sc);
    return sc * scale;
  };
  const unsigned char* kimg = smem + L_KV + (ro * 2 + 0) * KV_CAP * 64;
  const unsigned char* vimg = smem + L_KV + (ro * 2 + 1) * KV_CAP * 64;
  const int n_img = min(n_old, KV_CAP);
  constexpr int NT = KV_CAP / (16 * NW);               // image keys per lane
  // All 2 NT LDS reads are issued before the first use: taken one at a time (what the compiler emits when loads and uses
  // alternate) each costs a full LDS round trip, 2 NT + 2 of them in a row were ~2.5 us of the layer (in-kernel stamps).
  h8 kk[NT], vv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) kk[t] = *(const h8*)(kimg + ((NW * t + hq) * 16 + slot) * 64 + part * 16);
#pragma unroll
  for (int t = 0; t < NT; ++t) vv[t] = *(const h8*)(vimg + ((NW * t + hq) * 16 + slot) * 64 + part * 16);
  const h8 k_own = *(const h8*)(qkv_s + HD + part * 8), v_own = *(const h8*)(qkv_s + 2 * HD + part * 8);
  asm volatile("" ::: "memory");                       // keep the reads above the arithmetic
  float sc[NT + 1];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = (NW * t + hq) * 16 + slot;
    const float v = score(kk[t]);
    sc[t] = j < n_img ? v : -INFINITY;
    m = fmaxf(m, sc[t]);
  }
  {                                                    // this step's own key: wave hq == 0, key slot 0
    const float v = score(k_own);
    sc[NT] = (rv && hq == 0 && slot == 0) ? v : -INFINITY;
    m = fmaxf(m, sc[NT]);
  }
  // branch-free second pass: a masked key has p = exp(-inf) = 0 and its V is finite (the LDS image is zeroed at kernel start,
  // the K/V arena at creation: unwritten positions hold zeros or stale finite values), so 0 * V adds nothing
  const float mz = m == -INFINITY ? 0.f : m;
  float lsum = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int t = 0; t <= NT; ++t) {
    const h8 v8 = t < NT ? vv[t < NT ? t : 0] : v_own;
    const float p = __expf(sc[t] - mz);
    lsum += p;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += p * (float)v8[i];
  }
  if (n_old > KV_CAP) {                                // long rows: the tail comes straight from HBM (online update)
    const _Float16* kvb = a.kv + ((size_t)batch_row(q, r) * NH + q.head) * (size_t)a.smax * HD + part * 8;
    const _Float16* kg_ = kvb + (size_t)(l * 2 + 0) * a.kv_layer_stride;
    const _Float16* vg_ = kvb + (size_t)(l * 2 + 1) * a.kv_layer_stride;
    for (int j0 = KV_CAP + hq * 16; j0 < n_old; j0 += 16 * NW) {
      const int j = j0 + slot, jc = min(j, n_old - 1);
      const h8 kk = __builtin_nontemporal_load((const h8*)(kg_ + (size_t)jc * HD));
      const h8 vv = __builtin_nontemporal_load((const h8*)(vg_ + (size_t)jc * HD));
      const float v = score(kk);
      if (j < n_old) {
        const float mn = fmaxf(m, v);
        const float corr = m == -INFINITY ? 0.f : __expf(m - mn), p = __expf(v - mn);
        lsum = lsum * corr + p;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] * corr + p * (float)vv[i];
        m = mn;
      }
    }
  }
  const float wm = wave_max_dpp(m);
  const float f = (m == -INFINITY) ? 0.f : __expf(m - wm);
  lsum *= f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] *= f;
  // sum over the 4 key slots of each 16-lane row (lanes of equal `part`): two row shifts, lanes 12..15 hold the row's totals;
  // the 4 rows (and the row's other waves) are added by the combine step from LDS
  lsum += dpp_f<DPP_ROW_SHR4>(lsum);
  lsum += dpp_f<DPP_ROW_SHR8>(lsum);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += dpp_f<DPP_ROW_SHR4>(acc[i]);
    acc[i] += dpp_f<DPP_ROW_SHR8>(acc[i]);
  }
  float* s_m = (float*)(smem + L_ATT);
  float* s_acc = s_m + 8;                              // [aw][row][36]
  if ((q.lane & 15) >= 12) {
    const int row = q.lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) s_acc[(aw * 4 + row) * 36 + part * 8 + i] = acc[i];
    if (part == 0) s_acc[(aw * 4 + row) * 36 + 32] = lsum;
  }
  if (q.lane == 0) s_m[aw] = wm;
}

// K/V image staging by the comm waves: the cached K and V of the member's own rows for the NEXT layer go global -> registers
// (non-temporal 16-B lane loads = 1 KiB per wave instruction) and registers -> LDS after the current layer's attention has
// released the image.  A CU's vector-memory pipeline returns data in issue order ACROSS waves, so while long HBM loads are
// outstanding every poll of a hop waits behind them (measured: +2 us per hop with the loads in dedicated waves a layer ahead,
// 2.6 us of blocked issue per third of the image as LDS-DMA of the compute waves).  Therefore the loads are issued at B1 of
// P1 -- the start of the only poll-free stretch of a layer (QKV GEMM, reduce, attention: ~3 us; hop B cannot complete before
// the attention has finished anyway) -- by the comm waves, which have nothing else to do there and skip P1's inner barriers.
// KiB slot t of wave cw: row t / 10, K|V (t / 5) & 1, KiB 4 (t % 5) + cw.
constexpr int KV_T = KV_CAP / 16;            // KiB-instructions per (row, K|V) image = slots per wave (20)
struct KvStage { h8 r[KV_T]; };

// The row state is read ONCE per call: as written first, every one of the 20 loads sat behind two dependent LDS reads (active,
// kv_len), a readfirstlane and ~40 scalar address instructions -- the ISA showed the "burst" trickling out over ~3.5 us.
struct KvRows { int nki[2]; };
__device__ __forceinline__ KvRows kv_rows(const Ctx& q, int qd, int extra) {     // of quad qd (the one the image is staged for)
  KvRows k;
  const int Rq = min(RMAX, q.Rtot - RMAX * qd);
#pragma unroll
  for (int ro = 0; ro < 2; ++ro) {
    const int r = 2 * ro + q.half;                   // < RMAX
    const int act = ((lds_int*)(unsigned)L_ST)[RMAX * qd + r], len = ((lds_int*)(unsigned)L_ST + ST_N)[RMAX * qd + r];
    k.nki[ro] = __builtin_amdgcn_readfirstlane(r < Rq && act ? (min(len + extra, KV_CAP) + 15) >> 4 : 0);
  }
  return k;
}

__device__ __forceinline__ void kv_stage_load(const MegaArgs& a, const Ctx& q, int layer, int qd, int extra, KvStage& st) {
  const KvRows k = kv_rows(q, qd, extra);
  const h8* base[2][2];
#pragma unroll
  for (int ro = 0; ro < 2; ++ro)
#pragma unroll
    for (int which = 0; which < 2; ++which)
      // the arena is allocated to smax positions per (row, head): a partly valid KiB reads allocated memory
      base[ro][which] = (const h8*)(a.kv + (size_t)(layer * 2 + which) * a.kv_layer_stride +
                                    ((size_t)(q.group + MG_GROUPS * (RMAX * qd + 2 * ro + q.half)) * NH + q.head) * (size_t)a.smax * HD) + q.lane;
#pragma unroll
  for (int t = 0; t < KV_T; ++t) {
    const int ro = t / (KV_T / 2), which = (t / (KV_T / 4)) & 1, i = 4 * (t % (KV_T / 4)) + q.cw;
    if (i < k.nki[ro]) st.r[t] = __builtin_nontemporal_load(base[ro][which] + (size_t)i * (16 * HD / 8));
  }
}

__device__ __forceinline__ void kv_stage_store(const Ctx& q, int qd, int extra, const KvStage& st) {
  const KvRows k = kv_rows(q, qd, extra);
#pragma unroll
  for (int t = 0; t < KV_T; ++t) {
    const int ro = t / (KV_T / 2), which = (t / (KV_T / 4)) & 1, i = 4 * (t % (KV_T / 4)) + q.cw;
    if (i < k.nki[ro]) *((h8*)(q.smem + L_KV + (ro * 2 + which) * KV_CAP * 64 + i * 1024) + q.lane) = st.r[t];
  }
}

// sampling of the group's local row `member` by wave 0 of that member (reference utils.py:140-199, t2s_model.py:706-769): after the
// logits of every quad have been published, so that the samplers of all quads work side by side.  A MACRO, not a function: as an
// (inlined) function taking the context by reference the same text cost > 1000 spilled VGPRs in every kernel variant.
// Expects a, q, sp, s, ep0, EPS, seen in scope.
#define MG_RUN_SAMPLER() do { \
  const int V = a.V, EOS = a.V - 1; \
  const unsigned epE = ep0 + 4 * a.L + 2; \
    relaunder(q); \
    set_quad(q, q.member / RMAX); \
    q.hop = hop_slot(q, s, a.L); \
    const int r = q.member % RMAX, b = batch_row(q, r); \
    const unsigned epN = ep0 + (unsigned)EPS + 1; \
    const int was_active = st_active(q)[r]; \
    int now_active = 0; \
    float x0[8]; \
  _Pragma("unroll") \
    for (int k = 0; k < 8; ++k) x0[k] = 0.f; \
 \
 \
    unsigned lv[17]; \
    const bool ok = sweep<17>(q, q.hop + HOP_E + r * VPAD, V, epE, lv, 7u); \
    if (was_active && ok) { \
      const int step = st_step(q)[r]; \
      const int Veff = step < sp.eos_mask_steps ? V - 1 : V; \
      float x[17]; \
  _Pragma("unroll") \
      for (int i = 0; i < 17; ++i) { \
        const int v = q.lane + 64 * i; \
        if (v < V) a.logits_out[(size_t)b * V + v] = __uint_as_float(lv[i]); \
        x[i] = v < Veff ? __uint_as_float(lv[i]) : -INFINITY; \
        if (sp.rep_penalty != 1.0f && v < Veff && seen[v]) x[i] = x[i] < 0.f ? x[i] * sp.rep_penalty : x[i] / sp.rep_penalty; \
      } \
      const float* nrow = nullptr; \
      if (sp.noise) nrow = sp.noise + ((size_t)step * sp.noise_rows + (sp.noise_rows > 1 ? b : 0)) * V; \
      int smp, amx; \
      sample_core<17>(x, Veff, sp.top_k, sp.top_p, sp.temperature, nrow, sp.seed, b, step, &smp, &amx); \
      if (sp.dump) { \
  _Pragma("unroll") \
        for (int i = 0; i < 17; ++i) { \
          const int v = q.lane + 64 * i; \
          if (v < V) sp.dump[((size_t)step * a.B + b) * V + v] = __uint_as_float(lv[i]); \
        } \
      } \
      if (sp.drawn && q.lane == 0) { int* dr = sp.drawn + ((size_t)step * a.B + b) * 2; dr[0] = smp; dr[1] = amx; } \
      if (sp.force) { smp = sp.force[(size_t)b * sp.max_steps + step]; amx = smp; } \
      const bool fin = smp == EOS || amx == EOS; \
      const bool early = (sp.early_stop_num != -1 && (step + 1) > sp.early_stop_num) || step >= sp.max_steps - 1; \
      const int prev_len = sp.P + step; \
      if (q.lane == 0) { \
        if (prev_len < a.ycap) a.ytok[(size_t)b * a.ycap + prev_len] = smp; \
        if (smp >= 0 && smp < VPAD) seen[smp] = 1; \
        if (fin || early) { \
          a.active[b] = 0; \
          sp.out_len[b] = step; \
          atomicSub(a.n_active, 1); \
        } else { \
          sp.out_tokens[(size_t)b * sp.max_steps + step] = smp; \
          a.kv_len[b] = st_kvlen(q)[r] + 1; \
        } \
        a.step_ctr[b] = step + 1; \
      } \
      if (!(fin || early)) { \
        now_active = 1; \
        const int tok = min(max(smp, 0), V - 1); \
        const float* e = a.e_audio + (size_t)tok * D; \
        const float* p = a.pe + (size_t)(sp.P + step) * D; \
  _Pragma("unroll") \
        for (int k = 0; k < 8; ++k) x0[k] = e[k * 64 + q.lane] + a.alpha_a * p[k * 64 + q.lane]; \
      } \
    } \
    if (ok && s + 1 < a.nsteps) { \
      gu64* hn = hop_slot(q, s + 1, 0); \
  _Pragma("unroll") \
      for (int k = 0; k < 8; ++k) gstore(hn + HOP_A + r * 512 + k * 64 + q.lane, epN, __float_as_uint(x0[k])); \
      if (q.lane == 0) gstore(hn + HOP_ST + r, epN, (unsigned)now_active); \
    } \
} while (0)

// MULTI = false: the group has ONE quad (B <= 32), the quad loops below run once with qd = 0 and fold away.
// MULTI = true (32 < B <= 128): a group serves up to QMAX quads.  Every phase of a layer is run for quad 0, 1, ... in turn with
// the weights the compute waves already hold; each quad has its own hop buffers, so while a member works on quad q the other
// members' publishes of quad q + 1 are already on their way: the hop latency that a single quad waits out four times per layer
// is overlapped with the other quads' compute, and the weights are still streamed once per layer.
template <bool MULTI, bool PROF>
__device__ __forceinline__ void comm_role(const MegaArgs& a, const Ctx& c0, const StepParams& sp) {
  Ctx q = c0;
  unsigned char* smem = c0.smem;
  const int Rtot = c0.Rtot;
  const int nq = MULTI ? (Rtot + RMAX - 1) / RMAX : 1;
  const int V = a.V, EOS = a.V - 1;
  const bool sampler = c0.member < Rtot && c0.cw == 0;   // wave 0 of member m samples the group's local row m (quad m / 4, row m % 4)
  const int EPS = 4 * a.L + 2;                       // hops per step
  unsigned char* seen = smem + L_SEEN;
  KvStage kvs;
  set_quad(q, 0);
  kv_stage_load(a, q, 0, 0, 0, kvs);
  kv_stage_store(q, 0, 0, kvs);                      // layer 0's image of quad 0 (the attention waves read it after B1 + two compute barriers)
  // LayerNorm parameters of the current layer, loaded in P1's poll-free stretch (a load issued right before a sweep would sit
  // in front of its polls: a wave's memory operations return in order): norm1 for hop C, norm2 for the next hop A / the tail
  float gC[8], bC[8], gA[8], bA[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) gC[k] = bC[k] = gA[k] = bA[k] = 0.f;
  const int ra = c0.cw;                              // this wave's row of every quad
  for (int s = 0; s < a.nsteps; ++s) {
    const unsigned ep0 = a.ep_base + (unsigned)s * (unsigned)EPS;  // epoch of hop i of this step = ep0 + i + 1
    for (int l = 0; l < a.L; ++l) {
      const int ab_l = *st_abort(q);       // the abort test of this layer: read here, acted on behind the first barrier
      const float* lp = a.fpack + (size_t)l * FP_LAYER;
      // ================= P1 of every quad: hop A -> LayerNorm; the compute waves run QKV, append and attention
      for (int qd = 0; qd < nq; ++qd) {
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && qd == a.prof_quad;
        MG_STAMP(q, 0);
        // ---- hop A: y of the previous layer (or the step's input embedding) -> LayerNorm -> XS / XRES
        {
          const unsigned ep = ep0 + 4 * l + 1;
          if (ra < R) {
            u4v qa[4];
            const bool has_ln = l > 0;                       // layer 0's input is the embedding itself
            bool ok = true;
            if (s == 0 && l == 0) {
              const float* ya = a.ybuf + (size_t)batch_row(q, ra) * D + 2 * q.lane;
#pragma unroll
              for (int j = 0; j < 4; ++j) { const f2 y2 = *(const f2*)(ya + j * 128); qa[j] = (u4v){__float_as_uint(y2[0]), 0u, __float_as_uint(y2[1]), 0u}; }
            } else {
              ok = sweep_wide<4>(q, q.hop + HOP_A + ra * 512, ep, qa, 1u, a.hint_mask & 1);
            }
            if (ok) ln_row_wide(q, ra, qa, has_ln ? gA : nullptr, bA);
          }
          if (l == 0 && s > 0 && qd == 0 && q.cw == 0) {
            // row state published by the samplers with the embedding: {active} per row; ALL quads' rows here, in front of
            // quad 0's first barrier, so that every later slot of the step (K/V prefetch, exit test) sees the step's state
            for (int q2 = 0; q2 < nq; ++q2) {
              Ctx t = q;
              set_quad(t, q2);
              unsigned sv[1];
              if (sweep<1>(t, hop_slot(t, s, l) + HOP_ST, t.R, ep, sv, 2u) && t.lane < t.R) {
                const int was = st_active(t)[t.lane], now = (int)(sv[0] & 1u);
                if (was) { st_step(t)[t.lane] += 1; if (now) st_kvlen(t)[t.lane] += 1; }
                st_active(t)[t.lane] = now;
              }
            }
          }
        }
        MG_STAMP(q, 1);
        MG_BAR_AL(q);                                                       // B1
        if (l == 0 && s > 0 && qd == 0 && group_done(q)) return;           // every row of the group has finished
        MG_STAMP(q, 2);
        // K/V of the NEXT slot's quad: the next quad of this layer, else quad 0 of the next layer (across the step boundary the
        // rows hold one more position: appended at this step's layer 0); the compute waves run the QKV GEMM, the reduce and
        // the attention meanwhile, synchronised among themselves
        const bool kv_same = qd + 1 < nq;
        const int kv_nq = kv_same ? qd + 1 : 0;
        const int kv_nl = kv_same ? l : (l + 1 < a.L ? l + 1 : 0), kv_extra = (kv_same || l + 1 < a.L) ? 0 : 1;
        const bool kv_more = kv_same || l + 1 < a.L || s + 1 < a.nsteps;
        if (kv_more) kv_stage_load(a, q, kv_nl, kv_nq, kv_extra, kvs);
        if (qd + 1 == nq) {       // after the LAST quad's hop A: every quad's LayerNorm above still needed the previous layer's norm2
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int el = (k >> 1) * 128 + 2 * q.lane + (k & 1);      // the element sweep_wide<4> delivers in slot k
            gC[k] = lp[FP_N1W + el]; bC[k] = lp[FP_N1B + el];
            gA[k] = lp[FP_N2W + el]; bA[k] = lp[FP_N2B + el];
          }
        }
        MG_STAMP(q, 4);
        MG_BAR();                                                          // B4: the attention is done, the K/V image is free
        MG_STAMP(q, 5);
        relaunder(q);
        if (kv_more) kv_stage_store(q, kv_nq, kv_extra, kvs);
      }
      // ================= P2 of every quad: hop B (attention output of all heads) -> AT
      for (int qd = 0; qd < nq; ++qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && qd == a.prof_quad;
        if (ra < q.R) {
          u4v qb[2];
          if (sweep_wide<2>(q, q.hop + HOP_B + ra * 256, ep0 + 4 * l + 2, qb, 3u, a.hint_mask & 2)) {
            unsigned* at = (unsigned*)(smem + L_AT) + ra * (XS_LD / 2) + 2 * q.lane;
#pragma unroll
            for (int j = 0; j < 2; ++j) *(u2v*)(at + j * 128) = (u2v){qb[j][0], qb[j][2]};
          }
        }
        MG_STAMP(q, 6);
        MG_BAR();                                                           // B1
        MG_STAMP(q, 7);
        MG_BAR();                                                          // B2
        MG_STAMP(q, 8);
      }
      // ================= P3 of every quad: hop C (y1) -> LayerNorm1 -> XS / XRES
      for (int qd = 0; qd < nq; ++qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && qd == a.prof_quad;
        if (ra < q.R) {
          u4v qc[4];
          if (sweep_wide<4>(q, q.hop + HOP_C + ra * 512, ep0 + 4 * l + 3, qc, 4u, a.hint_mask & 4)) ln_row_wide(q, ra, qc, gC, bC);
        }
        MG_STAMP(q, 9);
        MG_BAR();                                                           // B1
        MG_STAMP(q, 10);
        if (MULTI) MG_BAR();                                               // B2 (sequential quads only: see compute_role)
        MG_STAMP(q, 11);
      }
      // ================= P4 of every quad: hop D (FFN hidden) -> HS
      for (int qd = 0; qd < nq; ++qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && qd == a.prof_quad;
        if (ra < q.R) {
          u4v qd8[8];
          if (sweep_wide<8>(q, q.hop + HOP_D + ra * 1024, ep0 + 4 * l + 4, qd8, 5u, a.hint_mask & 8)) {
            unsigned* hs = (unsigned*)(smem + L_HS) + ra * (HS_LD / 2) + 2 * q.lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) *(u2v*)(hs + j * 128) = (u2v){qd8[j][0], qd8[j][2]};
          }
        }
        MG_STAMP(q, 12);
        MG_BAR();                                                           // B1
        MG_STAMP(q, 13);
        MG_BAR();                                                          // B2
        MG_STAMP(q, 14);
      }
    }
    // ---- tail: hop A' -> LayerNorm2 of the last layer -> XS, logits by the compute waves; quad after quad
    const unsigned epA = ep0 + 4 * a.L + 1, epE = ep0 + 4 * a.L + 2;
    for (int qd = 0; qd < nq; ++qd) {
      relaunder(q);
      set_quad(q, qd);
      q.hop = hop_slot(q, s, a.L);
      q.prof_on = false;
      if (ra < q.R) {
        u4v qa[4];
        if (sweep_wide<4>(q, q.hop + HOP_A + ra * 512, epA, qa, 6u, a.hint_mask & 1)) ln_row_wide(q, ra, qa, gA, bA);   // norm2 of the last layer
      }
      MG_BAR();                                                             // B1
      MG_BAR();                                                            // B2
    }
    // ---- sampling (see run_sampler)
    if (sampler) MG_RUN_SAMPLER();
  }
}

template <bool MULTI, bool PROF>
__device__ __forceinline__ void compute_role(const MegaArgs& a, const Ctx& c0) {
  Ctx q = c0;
  unsigned char* smem = c0.smem;
  const int Rtot = c0.Rtot;
  const int nq = MULTI ? (Rtot + RMAX - 1) / RMAX : 1;
  const int V = a.V;
  const int EPS = 4 * a.L + 2;
  // own rows of this member (attention): rows 2*ro + half of the current quad, ro = 0, 1
  auto kv_row_base = [&](int layer, int which, int r) -> const _Float16* {
    const int b = batch_row(q, r);
    return a.kv + ((size_t)(layer * 2 + which)) * a.kv_layer_stride + ((size_t)b * NH + q.head) * (size_t)a.smax * HD;
  };
  const h8* wp = a.wpack;
  auto p1_src = [&](int layer) { return wp + ((size_t)layer * LAYER_HALFS + ((size_t)q.head * 4 + q.cw) * WI_P1 * 512) / 8 + q.lane; };
  auto pm_src = [&](int layer, int off) {
    return wp + ((size_t)layer * LAYER_HALFS + P1_HALFS + (((size_t)q.member * 4 + q.cw) * (WI_P2 + WI_P3 + WI_P4) + off) * 512) / 8 + q.lane;
  };
  // residual operands of the out-projection (LN(y) of hop A) and of FFN2 (LN1(y1) of hop C) for the reducing wave's lanes.
  // One quad: read from XRES when needed.  Several quads: XRES is overwritten by the next quad's LayerNorm before the phase
  // that needs it comes round, so the member's 16 columns of every row are parked per quad (f4 per lane, 1 KB per quad).
  auto rs_slot = [&](int which, int qd) { return (f4*)(smem + L_HS + RMAX * HS_LD * 2) + (which * QMAX + qd) * 64 + q.lane; };   // (the second HS copy: unused by these roles)
  // K/V arena append of this step's k, v (own rows), from the LDS copy the attention used
  auto arena_append = [&](int l) {
    const int ro = q.lane >> 4, which = (q.lane >> 3) & 1, e = 4 * (q.lane & 7), r = 2 * ro + q.half;
    if (r < q.R && st_active(q)[r]) {
      const int pos = st_kvlen(q)[r];
      if (pos < a.smax)
        *(h4*)(const_cast<_Float16*>(kv_row_base(l, which, r)) + (size_t)pos * HD + e) =
            *(const h4*)((const _Float16*)(smem + L_QKV) + (ro * 3 + 1 + which) * HD + e);
    }
  };
  // P1's slice is held as two halves (tiles q0 q1 k0 | k1 v0 v1): the second half is requested one phase later than the
  // first, so that FFN2's slice + the whole next P1 slice are never live together (160 + operands would spill)
  h8 wA0[WI_P1 / 2], wA1[WI_P1 / 2], wB[WI_P2], wC[WI_P3], wD[WI_P4];       // the logits slice reuses wA0
  int cgen = 0;                                       // compute_barrier generation
  // Weight slices are requested right AFTER a hop has landed (B1), never right after a publish: a load issued behind a
  // publish sits in the CU's in-order memory pipeline in front of the sweep waves' polls of the next hop.
  wload(wA0, p1_src(0));
  wload(wA1, p1_src(0) + (size_t)(WI_P1 / 2) * 64);

  for (int s = 0; s < a.nsteps; ++s) {
    const unsigned ep0 = a.ep_base + (unsigned)s * (unsigned)EPS;
    for (int l = 0; l < a.L; ++l) {
      const int ab_l = *st_abort(q);       // the abort test of this layer: read here, acted on behind the first barrier
      const float* lp = a.fpack + (size_t)l * FP_LAYER;
      // ================= P1: q,k,v of head `head` for the own rows, attention (K/V append: see P2)
      // (the next phase's weight slice is requested in EVERY quad's slot, not only the first: a request under a run-time
      // `if (qd == 0)` inside the quad loop is a CONDITIONAL definition of the register array, which keeps the array's
      // previous contents alive around the whole loop -- every slice live everywhere, ~100 spilled VGPRs.  The repeated
      // requests hit L2 and rewrite the registers with the same bytes; no slot reads the array it requests)
      for (int qd = 0; qd < nq; ++qd) {
        const bool first = qd == 0;
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && first;
        MG_STAMP(q, 0);
        // lane j < 24 of every wave reduces 4 values of q | k | v of the wave's OWN attention row (see after B2)
        const int rj_which = (q.lane >> 3) % 3, rj_e = 4 * (q.lane & 7);
        const f4 p1_bias = q.lane < 24 ? *(const f4*)(lp + FP_QKVB + rj_which * D + q.head * HD + rj_e) : (f4){0.f, 0.f, 0.f, 0.f};
        MG_STAMP(q, 1);
        MG_BAR_AL(q);                                                       // B1: XS / XRES hold LN(y)
        if (first && l == 0 && s > 0 && group_done(q)) return;
        MG_STAMP(q, 2);
        wload(wB, pm_src(l, 0));                                           // every quad's slot (see the note above the loop)
        gemm_chunk<3>(q, wA0, (const _Float16*)(smem + L_XS), XS_LD, q.cw, q.cw * 6);
        gemm_chunk<3>(q, wA1, (const _Float16*)(smem + L_XS), XS_LD, q.cw, q.cw * 6 + 3);
        if (MULTI && q.cw == 0 && (q.lane & 15) < R)                       // the out-projection's residual operand of this quad
          *rs_slot(0, qd) = *(const f4*)((const float*)(smem + L_XRES) + (q.lane & 15) * D + 16 * q.member + 4 * (q.lane >> 4));
        MG_STAMP(q, 3);
        if (compute_barrier(q, cgen)) return;                              // B2 (compute waves only)
        MG_STAMP(q, 4);
        {
          // split-K reduce of q, k, v (+ bias) by the wave that consumes them: each attention wave sums the 96 values of ITS row
          // (both waves of a row do, and write identical bytes), so that no barrier stands between the reduce and the attention
          // -- the workgroup-wide reduce + compute-only barrier here were 0.7 us of every layer (stamps 4-7).  Same summation
          // order as before: ((w0 + w1) + w2) + w3 + bias.
          const f4* red = (const f4*)(smem + L_RED);
          _Float16* qkv_s = (_Float16*)(smem + L_QKV);
          const int ro = q.cw / 2, r = 2 * ro + q.half;
          if (q.lane < 24 && r < R) {
            const int tile = rj_which * 2 + (rj_e >> 4), ln = r + 16 * ((rj_e & 15) >> 2);
            const f4 v0 = red[red_idx(0 * 6 + tile, ln)], v1 = red[red_idx(1 * 6 + tile, ln)], v2 = red[red_idx(2 * 6 + tile, ln)],
                     v3 = red[red_idx(3 * 6 + tile, ln)];
            f4 v = v0;
            v += v1; v += v2; v += v3;
            v += p1_bias;
            const h4 ov = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *(h4*)(qkv_s + (ro * 3 + rj_which) * HD + rj_e) = ov;    // the arena append of k, v follows (see arena_append)
          }
          asm volatile("" ::: "memory");                      // the attention's reads of qkv_s stay behind the writes (same wave: in order)
          MG_STAMP(q, 5);
        }
        MG_STAMP(q, 6);
        MG_STAMP(q, 7);
        attention_part<2>(a, q, l, q.cw);
        MG_STAMP(q, 8);
        // the partials of a row's two waves meet through LDS.  This barrier is B4 itself (the attention has released the K/V
        // image): the comm waves only have to ISSUE the next layer's K/V loads before it, which takes ~1.5 us since the row
        // state is read once per call (it was 3.5 us, and a compute-only barrier stood here so that the publish did not wait)
        MG_BAR();                                                           // B4
        MG_STAMP(q, 9);
        if (q.cw == 0) {
          // combine the 2 waves of each own row; lane = ro * 32 + e
          const float* s_m = (const float*)(smem + L_ATT);
          const float* s_acc = s_m + 8;
          const int ro = q.lane >> 5, e = q.lane & 31, r = 2 * ro + q.half;
          float M = -INFINITY;
#pragma unroll
          for (int w = 0; w < 2; ++w) M = fmaxf(M, s_m[2 * ro + w]);
          float o = 0.f;
          if (M != -INFINITY) {
            float Lsum = 0.f, num = 0.f;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              const float mw = s_m[2 * ro + w];
              const float ew = mw == -INFINITY ? 0.f : __expf(mw - M);
              const float* pa = s_acc + (2 * ro + w) * 4 * 36;
              Lsum += ((pa[32] + pa[36 + 32]) + (pa[72 + 32] + pa[108 + 32])) * ew;
              num += ((pa[e] + pa[36 + e]) + (pa[72 + e] + pa[108 + e])) * ew;
            }
            o = num / Lsum;
          }
          const float o2 = dpp_f<DPP_QUAD_XOR1>(o);            // lane ^ 1
          // 16 granules (one 128-B line) per own row, written by one wave instruction
          if (r < R && !(e & 1)) gstore(q.hop + HOP_B + r * 256 + (q.head * HD + e) / 2, ep0 + 4 * l + 2, pack_h2(o, o2));
        }
        // several quads: the next quad's reduce overwrites the q/k/v copy before P2 comes round, so the arena append happens
        // here (the comm waves' K/V burst of the next slot was issued a whole attention ago and has drained)
        if (MULTI && q.cw == 1 && q.lane < 32) arena_append(l);
        MG_STAMP(q, 10);
      }
      // ================= P2: out-projection columns [16 member, +16) + bias + residual -> y1
      for (int qd = 0; qd < nq; ++qd) {
        const bool first = qd == 0;
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && first;
        const f4 p_bias = *(const f4*)(lp + FP_OUTB + 16 * q.member + 4 * (q.lane >> 4));
        MG_STAMP(q, 11);
        MG_BAR();                                                           // B1: AT holds the attention output
        MG_STAMP(q, 12);
        wload(wC, pm_src(l, WI_P2));
        gemm_chunk<1>(q, wB, (const _Float16*)(smem + L_AT), XS_LD, q.cw, q.cw);
        MG_STAMP(q, 13);
        MG_BAR();                                                          // B2
        MG_STAMP(q, 14);
        if (q.cw == 0) {
          const f4* red = (const f4*)(smem + L_RED);
          const int r = q.lane & 15, n0 = 16 * q.member + 4 * (q.lane >> 4);
          if (r < R) {
            const f4 v0 = red[red_idx(0, q.lane)], v1 = red[red_idx(1, q.lane)], v2 = red[red_idx(2, q.lane)], v3 = red[red_idx(3, q.lane)];
            const f4 xr = MULTI ? *rs_slot(0, qd) : *(const f4*)((const float*)(smem + L_XRES) + r * D + n0);
            f4 v = v0;
            v += v1; v += v2; v += v3;
            v += p_bias;
            v += xr;
            *(f4*)((float*)(smem + L_STAGE) + r * 16 + 4 * (q.lane >> 4)) = v;
          }
          // transposed through LDS (same wave): lane = row * 16 + column, so each row's 16 granules = one 128-B line
          // written whole by ONE store instruction (scattered 8-B write-through stores made every hop 2-3x slower)
          // test hook (gsv_t2s_debug_stall, tests only): that member of group 0 skips ONE publish, so that the group's
          // bounded waits must end the launch with an error instead of hanging
          const bool stall = a.test_stall && q.member == a.test_stall - 1 && q.group == 0 && s == 2 && l == 3 && qd == 0;
          if (q.lane < R * 16 && !stall)
            gstore(q.hop + HOP_C + (q.lane >> 4) * 512 + 16 * q.member + (q.lane & 15), ep0 + 4 * l + 3,
                   __float_as_uint(((const float*)(smem + L_STAGE))[q.lane]));
        }
        if (!MULTI && q.cw == 1 && q.lane < 32) {   // wave 1 is idle while wave 0 publishes
          // K/V arena append: not in P1 where k, v are produced -- there the store has to queue behind the comm waves' K/V
          // burst of the next layer in the CU's memory pipeline (1.3 us of blocked issue in the stamps); nothing reads the
          // arena row before the next step's staging
          arena_append(l);
        }
      }
      // ================= P3: FFN1 columns [64 member, +64), ReLU -> h
      for (int qd = 0; qd < nq; ++qd) {
        const bool first = qd == 0;
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && first;
        const f4 p_bias = *(const f4*)(lp + FP_B1 + 64 * q.member + 16 * q.cw + 4 * (q.lane >> 4));       // tile cw of the member's four
        MG_STAMP(q, 15);
        MG_BAR();                                                           // B1: XS / XRES hold LN1(y1)
        MG_STAMP(q, 16);
        wload(wD, pm_src(l, WI_P2 + WI_P3));
        f4 v = gemm_tile_k512(q, wC, (const _Float16*)(smem + L_XS), XS_LD);
        if (MULTI && q.cw == 2 && (q.lane & 15) < R)                       // FFN2's residual operand of this quad
          *rs_slot(1, qd) = *(const f4*)((const float*)(smem + L_XRES) + (q.lane & 15) * D + 16 * q.member + 4 * (q.lane >> 4));
        MG_STAMP(q, 17);
        // sequential quads: the comm waves overwrite XS with the next quad's rows right after this barrier.  One quad: none
        // needed -- the next writer of XS is hop A of the next layer, two workgroup barriers away
        if (MULTI) MG_BAR();                                               // B2
        MG_STAMP(q, 18);
        {
          // every wave publishes its own tile: 16 columns = 8 half-pair granules = 64 contiguous bytes per row, the rows of
          // the quad by one store instruction (transposed through the wave's own LDS patch: lane = row * 8 + granule)
          const int r = q.lane & 15;
          unsigned* stage = (unsigned*)(smem + L_STAGE) + q.cw * 64;
          v += p_bias;
          if (r < R) {
            const int w0 = r * 8 + 2 * (q.lane >> 4);
            stage[w0] = pack_h2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
            stage[w0 + 1] = pack_h2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
          }
          if (q.lane < R * 8)
            gstore(q.hop + HOP_D + (q.lane >> 3) * 1024 + 32 * q.member + 8 * q.cw + (q.lane & 7), ep0 + 4 * l + 4, stage[q.lane]);
        }
      }
      // ================= P4: FFN2 columns [16 member, +16) over K = 2048 (each wave chains its 4 chunks of 128, 4 partials) -> y2
      for (int qd = 0; qd < nq; ++qd) {
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        q.prof_on = PROF && s == a.prof_step && l == a.prof_layer && qd == a.prof_quad;
        const f4 p_bias = *(const f4*)(lp + FP_B2 + 16 * q.member + 4 * (q.lane >> 4));
        MG_STAMP(q, 19);
        MG_BAR();                                                           // B1: HS holds the FFN hidden
        MG_STAMP(q, 20);
        if (l + 1 < a.L) wload(wA0, p1_src(l + 1));                        // P1 has finished with wA0 for every quad
        else wload(wA0, a.lpack + (((size_t)q.member * 4 + q.cw) * WI_LG * 512) / 8 + q.lane);
        {
          const int rowl = q.lane & 15, kg = q.lane >> 4;
          f4* red = (f4*)(smem + L_RED);
          // one accumulator over the wave's four 128-wide chunks (the MFMA forwards a dependent accumulator without a stall):
          // ONE partial per wave like the other phases, instead of 16 per workgroup through LDS
          f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
          // the operand reads of chunk cc + 1 are in flight under the MFMAs of chunk cc (two register sets of 4 fragments; same
          // MFMA order, bit-identical): 295.8 -> 292.6 us per step over 4 pairs (3 of 4), profiles/r03_ab_ffn2_operand_prefetch.txt
          const _Float16* bp0 = (const _Float16*)(smem + L_HS) + (rowl & (RMAX - 1)) * HS_LD + (4 * q.cw) * 128 + 8 * kg;
          h8 bA[4], bB[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) bA[ks] = *(const h8*)(bp0 + 32 * ks);
#pragma unroll
          for (int cc = 0; cc < 4; cc += 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bB[ks] = *(const h8*)(bp0 + (cc + 1) * 128 + 32 * ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wD[cc * 4 + ks], bA[ks], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (cc + 2 < 4) {
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) bA[ks] = *(const h8*)(bp0 + (cc + 2) * 128 + 32 * ks);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wD[(cc + 1) * 4 + ks], bB[ks], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (rowl < RMAX) red[red_idx(q.cw, q.lane)] = acc;
        }
        // second half of the next P1 slice: this quad's FFN2 is done, and the request is still in front of the publish
        wload(wA1, p1_src(l + 1 < a.L ? l + 1 : 0) + (size_t)(WI_P1 / 2) * 64);
        MG_STAMP(q, 21);
        MG_BAR();                                                          // B2
        MG_STAMP(q, 22);
        if (q.cw == 0) {
          const f4* red = (const f4*)(smem + L_RED);
          const int r = q.lane & 15, n0 = 16 * q.member + 4 * (q.lane >> 4);
          if (r < R) {
            const f4 v0 = red[red_idx(0, q.lane)], v1 = red[red_idx(1, q.lane)], v2 = red[red_idx(2, q.lane)], v3 = red[red_idx(3, q.lane)];
            const f4 xr = MULTI ? *rs_slot(1, qd) : *(const f4*)((const float*)(smem + L_XRES) + r * D + n0);
            f4 v = v0;
            v += v1; v += v2; v += v3;
            v += p_bias;
            v += xr;
            // y2 feeds hop A of the next layer, or hop A' (the logits' LayerNorm) after the last layer
            *(f4*)((float*)(smem + L_STAGE) + r * 16 + 4 * (q.lane >> 4)) = v;
          }
          const unsigned ep = l + 1 < a.L ? ep0 + 4 * (l + 1) + 1 : ep0 + 4 * a.L + 1;
          if (q.lane < R * 16)
            gstore(hop_slot(q, s, l + 1) + HOP_A + (q.lane >> 4) * 512 + 16 * q.member + (q.lane & 15), ep,
                   __float_as_uint(((const float*)(smem + L_STAGE))[q.lane]));
        }
        MG_STAMP(q, 23);
      }
    }
    // ================= tail: logits split over the members (tiles member, member + 32, and tile 64 on member 0)
    const unsigned epE = ep0 + 4 * a.L + 2;
    for (int qd = 0; qd < nq; ++qd) {
      relaunder(q);
      set_quad(q, qd);
      const int R = q.R;
      q.hop = hop_slot(q, s, a.L);
      q.prof_on = false;
      MG_BAR();                                                             // B1: XS holds LN2(y) of the last layer
      gemm_chunk<3>(q, wA0, (const _Float16*)(smem + L_XS), XS_LD, q.cw, q.cw * 3);
      if (!MULTI) wload(wA0, p1_src(0));                                   // next step's layer 0 (two hops away; wA1 is there already)
      MG_BAR();                                                            // B2
      if (q.cw < 3) {
        const f4* red = (const f4*)(smem + L_RED);
        const int t = q.cw, r = q.lane & 15;
        const int tile = t == 0 ? q.member : (t == 1 ? q.member + 32 : 64);
        float* stage = (float*)(smem + L_STAGE) + q.cw * 256;
        if (r < R) {
          f4 v = red[red_idx(0 * 3 + t, q.lane)];
          v += red[red_idx(1 * 3 + t, q.lane)]; v += red[red_idx(2 * 3 + t, q.lane)]; v += red[red_idx(3 * 3 + t, q.lane)];
          *(f4*)(stage + r * 16 + 4 * (q.lane >> 4)) = v;
        }
        if (q.lane < R * 16 && (t < 2 || q.member == 0) && 16 * tile + (q.lane & 15) < V)
          gstore(q.hop + HOP_E + (q.lane >> 4) * VPAD + 16 * tile + (q.lane & 15), epE, __float_as_uint(stage[q.lane]));
      }
    }
    // several quads: the logits slice is read by every quad's slot, the next step's first slice follows the last one
    if (MULTI) wload(wA0, p1_src(0));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined quads (32 < B <= 128).  The sequential-quad roles above (MULTI = true) wait out every quad's sweep round trip
// (~1.1 us even when the data arrived long ago) in front of its GEMM: a GEMM-phase slot costs 1.8-2.7 us, a layer 44 us
// for four quads (in-kernel stamps, gpurun_out r3_prof128_*).  Here the comm waves sweep quad i + 1's hop into the SECOND
// copy of the operand image while the compute waves multiply, reduce and publish quad i -- one workgroup barrier per slot:
//     comm:    sweep(0) | B | sweep(1)              | B | sweep(2)              | B | ...
//     compute:          | B | gemm(0) cb reduce(0)  | B | gemm(1) cb reduce(1)  | B | ...
// and P1 (QKV + attention) keeps two barriers per slot: B_a (the previous quad's attention has released the K/V image,
// LN(hop A) of this quad is in its XS copy) and B_b (the image of this quad is stored; the QKV partials are parked).
// The member's 16 residual columns of every row are parked per quad by the LayerNorm itself (no XRES image).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f4* rs_at(unsigned char* smem, int which, int qd, int lane) { return (f4*)(smem + L_RS) + (which * QMAX + qd) * 64 + lane; }

// LayerNorm of one row delivered by sweep_wide<4> -> XS copy `buf` (MFMA operand) + the member's residual columns -> rs[which][qd]
__device__ __forceinline__ void ln_row_pipe(const Ctx& c, int row, int buf, int which, const u4v (&v)[4], const float* gm, const float* bt) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { x[2 * j] = __uint_as_float(v[j][0]); x[2 * j + 1] = __uint_as_float(v[j][2]); }
  if (gm) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    const float mean = wave_sum_dpp(s) * (1.f / D);
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float dl = x[k] - mean; qq += dl * dl; }
    const float rstd = rsqrtf(wave_sum_dpp(qq) * (1.f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (x[k] - mean) * rstd * gm[k] + bt[k];
  }
  _Float16* xs = (_Float16*)(c.smem + L_XS) + (buf * RMAX + row) * XS_LD + 2 * c.lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) *(h2*)(xs + j * 128) = (h2){(_Float16)x[2 * j], (_Float16)x[2 * j + 1]};
  // columns [16 member, +16) of this row: chunk j = member / 8, lanes 8 (member % 8) .. + 7, two elements each; the reducing
  // wave's lane r + 16 g holds columns 4 g .. 4 g + 3 of row r
  const int jm = c.member >> 3, m0 = 8 * (c.member & 7), dm = c.lane - m0;
  const float e0 = jm == 0 ? x[0] : jm == 1 ? x[2] : jm == 2 ? x[4] : x[6];
  const float e1 = jm == 0 ? x[1] : jm == 1 ? x[3] : jm == 2 ? x[5] : x[7];
  if (dm >= 0 && dm < 8)
    *(f2*)((float*)rs_at(c.smem, which, c.qd, row + 16 * (dm >> 1)) + 2 * (dm & 1)) = (f2){e0, e1};
}

__device__ __forceinline__ void comm_role_pipe(const MegaArgs& a, const Ctx& c0, const StepParams& sp) {
  Ctx q = c0;
  unsigned char* smem = c0.smem;
  const int Rtot = c0.Rtot;
  const int nq = (Rtot + RMAX - 1) / RMAX;
  const bool sampler = c0.member < Rtot && c0.cw == 0;
  const int EPS = 4 * a.L + 2;
  unsigned char* seen = smem + L_SEEN;
  KvStage kvs;
  set_quad(q, 0);
  kv_stage_load(a, q, 0, 0, 0, kvs);                 // image of (layer 0, quad 0): stored by the first P1 slot like every other
  float gC[8], bC[8], gA[8], bA[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) gC[k] = bC[k] = gA[k] = bA[k] = 0.f;
  const int ra = c0.cw;                              // this wave's row of every quad
  for (int s = 0; s < a.nsteps; ++s) {
    const unsigned ep0 = a.ep_base + (unsigned)s * (unsigned)EPS;
    for (int l = 0; l < a.L; ++l) {
      const int ab_l = *st_abort(q);       // the abort test of this layer: read here, acted on behind the first barrier
      const float* lp = a.fpack + (size_t)l * FP_LAYER;
      // hop A of quad qd -> LayerNorm -> XS copy qd & 1 (+ the residual columns of the out-projection)
      auto sweep_a = [&](int qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        if (ra < q.R) {
          u4v qa[4];
          bool ok = true;
          if (s == 0 && l == 0) {
            const float* ya = a.ybuf + (size_t)batch_row(q, ra) * D + 2 * q.lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const f2 y2 = *(const f2*)(ya + j * 128); qa[j] = (u4v){__float_as_uint(y2[0]), 0u, __float_as_uint(y2[1]), 0u}; }
          } else {
            ok = sweep_wide<4>(q, q.hop + HOP_A + ra * 512, ep0 + 4 * l + 1, qa, 1u, a.hint_mask & 1);
          }
          if (ok) ln_row_pipe(q, ra, qd & 1, 0, qa, l > 0 ? gA : nullptr, bA);
        }
      };
      // ================= P1
      q.prof_on = false;
      sweep_a(0);
      if (l == 0 && s > 0 && q.cw == 0) {
        // row state published by the samplers with the embedding, every quad's rows (see comm_role)
        for (int q2 = 0; q2 < nq; ++q2) {
          Ctx t = q;
          set_quad(t, q2);
          unsigned sv[1];
          if (sweep<1>(t, hop_slot(t, s, l) + HOP_ST, t.R, ep0 + 1, sv, 2u) && t.lane < t.R) {
            const int was = st_active(t)[t.lane], now = (int)(sv[0] & 1u);
            if (was) { st_step(t)[t.lane] += 1; if (now) st_kvlen(t)[t.lane] += 1; }
            st_active(t)[t.lane] = now;
          }
        }
      }
      for (int qd = 0; qd < nq; ++qd) {
        MG_BAR_AL(q);                                                       // B_a
        if (l == 0 && s > 0 && qd == 0 && group_done(q)) return;
        relaunder(q);
        // the image of THIS slot's quad (requested one slot ago) -> LDS.  At (l = 0, qd = 0) of a step after the first the rows'
        // state has just advanced: the image was requested with extra = 1 at the end of the previous step, the store uses the
        // advanced state with extra = 0 -- the same positions
        kv_stage_store(q, qd, 0, kvs);
        MG_BAR();                                                           // B_b
        // the NEXT slot's image (the compute waves run the reduce and the attention meanwhile)
        const bool kv_same = qd + 1 < nq;
        const int kv_nq = kv_same ? qd + 1 : 0;
        const int kv_nl = kv_same ? l : (l + 1 < a.L ? l + 1 : 0), kv_extra = (kv_same || l + 1 < a.L) ? 0 : 1;
        if (kv_same || l + 1 < a.L || s + 1 < a.nsteps) kv_stage_load(a, q, kv_nl, kv_nq, kv_extra, kvs);
        if (kv_same) sweep_a(qd + 1);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int el = (k >> 1) * 128 + 2 * q.lane + (k & 1);
            gC[k] = lp[FP_N1W + el]; bC[k] = lp[FP_N1B + el];
            gA[k] = lp[FP_N2W + el]; bA[k] = lp[FP_N2B + el];
          }
        }
      }
      // ================= P2: hop B -> AT copies
      auto sweep_b = [&](int qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        if (ra < q.R) {
          u4v qb[2];
          if (sweep_wide<2>(q, q.hop + HOP_B + ra * 256, ep0 + 4 * l + 2, qb, 3u, a.hint_mask & 2)) {
            unsigned* at = (unsigned*)(smem + L_AT) + ((qd & 1) * RMAX + ra) * (XS_LD / 2) + 2 * q.lane;
#pragma unroll
            for (int j = 0; j < 2; ++j) *(u2v*)(at + j * 128) = (u2v){qb[j][0], qb[j][2]};
          }
        }
      };
      sweep_b(0);
      for (int qd = 0; qd < nq; ++qd) {
        MG_BAR();
        if (qd + 1 < nq) sweep_b(qd + 1);
      }
      // ================= P3: hop C -> LayerNorm1 -> XS copies (+ the residual columns of FFN2)
      auto sweep_c = [&](int qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        if (ra < q.R) {
          u4v qc[4];
          if (sweep_wide<4>(q, q.hop + HOP_C + ra * 512, ep0 + 4 * l + 3, qc, 4u, a.hint_mask & 4)) ln_row_pipe(q, ra, qd & 1, 1, qc, gC, bC);
        }
      };
      sweep_c(0);                       // no barrier between the phases: this sweep overlaps the last out-projection slot
      for (int qd = 0; qd < nq; ++qd) {
        MG_BAR();
        if (qd + 1 < nq) sweep_c(qd + 1);
      }
      // ================= P4: hop D -> HS copies
      auto sweep_d = [&](int qd) {
        relaunder(q);
        set_quad(q, qd);
        q.hop = hop_slot(q, s, l);
        if (ra < q.R) {
          u4v qd8[8];
          if (sweep_wide<8>(q, q.hop + HOP_D + ra * 1024, ep0 + 4 * l + 4, qd8, 5u, a.hint_mask & 8)) {
            unsigned* hs = (unsigned*)(smem + L_HS) + ((qd & 1) * RMAX + ra) * (HS_LD / 2) + 2 * q.lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) *(u2v*)(hs + j * 128) = (u2v){qd8[j][0], qd8[j][2]};
          }
        }
      };
      sweep_d(0);
      for (int qd = 0; qd < nq; ++qd) {
        MG_BAR();
        if (qd + 1 < nq) sweep_d(qd + 1);
      }
    }
    // ---- tail: hop A' -> LayerNorm2 of the last layer -> XS copies; logits by the compute waves
    auto sweep_t = [&](int qd) {
      relaunder(q);
      set_quad(q, qd);
      q.hop = hop_slot(q, s, a.L);
      if (ra < q.R) {
        u4v qa[4];
        if (sweep_wide<4>(q, q.hop + HOP_A + ra * 512, ep0 + 4 * a.L + 1, qa, 6u, a.hint_mask & 1)) ln_row_pipe(q, ra, qd & 1, 0, qa, gA, bA);
      }
    };
    sweep_t(0);
    for (int qd = 0; qd < nq; ++qd) {
      MG_BAR();
      if (qd + 1 < nq) sweep_t(qd + 1);
    }
    MG_BAR();                 // end of the step's GEMMs: the last logits GEMM has read its XS copy (the next step's first sweep rewrites copy 0)
    if (sampler) MG_RUN_SAMPLER();
  }
}

__device__ __forceinline__ void compute_role_pipe(const MegaArgs& a, const Ctx& c0) {
  Ctx q = c0;
  unsigned char* smem = c0.smem;
  const int Rtot = c0.Rtot;
  const int nq = (Rtot + RMAX - 1) / RMAX;
  const int V = a.V;
  const int EPS = 4 * a.L + 2;
  auto kv_row_base = [&](int layer, int which, int r) -> const _Float16* {
    const int b = batch_row(q, r);
    return a.kv + ((size_t)(layer * 2 + which)) * a.kv_layer_stride + ((size_t)b * NH + q.head) * (size_t)a.smax * HD;
  };
  const h8* wp = a.wpack;
  auto p1_src = [&](int layer) { return wp + ((size_t)layer * LAYER_HALFS + ((size_t)q.head * 4 + q.cw) * WI_P1 * 512) / 8 + q.lane; };
  auto pm_src = [&](int layer, int off) {
    return wp + ((size_t)layer * LAYER_HALFS + P1_HALFS + (((size_t)q.member * 4 + q.cw) * (WI_P2 + WI_P3 + WI_P4) + off) * 512) / 8 + q.lane;
  };
  h8 wA0[WI_P1 / 2], wA1[WI_P1 / 2], wB[WI_P2], wC[WI_P3], wD[WI_P4];
  int cgen = 0;
  wload(wA0, p1_src(0));
  wload(wA1, p1_src(0) + (size_t)(WI_P1 / 2) * 64);
  for (int s = 0; s < a.nsteps; ++s) {
    const unsigned ep0 = a.ep_base + (unsigned)s * (unsigned)EPS;
    for (int l = 0; l < a.L; ++l) {
      const int ab_l = *st_abort(q);       // the abort test of this layer: read here, acted on behind the first barrier
      const float* lp = a.fpack + (size_t)l * FP_LAYER;
      // ================= P1.  The first slot's barrier and the next phase's weight request stand in FRONT of the quad loop: a
      // request inside the loop is either conditional (`if (qd == 0)`: a conditional definition keeps the register array's old
      // contents alive around the whole loop, ~100 spilled VGPRs) or repeated per quad -- and the weight loads are
      // non-temporal, so the repeats miss L2 and every quad streamed the layer's slice again from the memory side: 6 GB per
      // step at B = 128, no faster than the sequential quads (1.07 ms per step both)
      MG_BAR_AL(q);                                                         // B_a of quad 0
      if (l == 0 && s > 0 && group_done(q)) return;
      relaunder(q);
      wload(wB, pm_src(l, 0));
      for (int qd = 0; qd < nq; ++qd) {
        if (qd > 0) {
          MG_BAR();                                                         // B_a: XS copy qd & 1 holds LN(y); the image is free
        }
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        q.prof_on = false;
        const int rj_which = (q.lane >> 3) % 3, rj_e = 4 * (q.lane & 7);
        const f4 p1_bias = q.lane < 24 ? *(const f4*)(lp + FP_QKVB + rj_which * D + q.head * HD + rj_e) : (f4){0.f, 0.f, 0.f, 0.f};
        const _Float16* xs = (const _Float16*)(smem + L_XS) + (qd & 1) * RMAX * XS_LD;
        gemm_chunk<3>(q, wA0, xs, XS_LD, q.cw, q.cw * 6);
        gemm_chunk<3>(q, wA1, xs, XS_LD, q.cw, q.cw * 6 + 3);
        MG_BAR();                                                           // B_b: partials parked, this quad's K/V image stored
        {
          const f4* red = (const f4*)(smem + L_RED);
          _Float16* qkv_s = (_Float16*)(smem + L_QKV);
          const int ro = q.cw / 2, r = 2 * ro + q.half;
          if (q.lane < 24 && r < R) {
            const int tile = rj_which * 2 + (rj_e >> 4), ln = r + 16 * ((rj_e & 15) >> 2);
            const f4 v0 = red[red_idx(0 * 6 + tile, ln)], v1 = red[red_idx(1 * 6 + tile, ln)], v2 = red[red_idx(2 * 6 + tile, ln)],
                     v3 = red[red_idx(3 * 6 + tile, ln)];
            f4 v = v0;
            v += v1; v += v2; v += v3;
            v += p1_bias;
            *(h4*)(qkv_s + (ro * 3 + rj_which) * HD + rj_e) = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          }
          asm volatile("" ::: "memory");
        }
        attention_part<2>(a, q, l, q.cw);
        if (compute_barrier(q, cgen)) return;                                          // the partials of a row's two waves meet (compute waves only)
        if (q.cw == 0) {
          const float* s_m = (const float*)(smem + L_ATT);
          const float* s_acc = s_m + 8;
          const int ro = q.lane >> 5, e = q.lane & 31, r = 2 * ro + q.half;
          float M = -INFINITY;
#pragma unroll
          for (int w = 0; w < 2; ++w) M = fmaxf(M, s_m[2 * ro + w]);
          float o = 0.f;
          if (M != -INFINITY) {
            float Lsum = 0.f, num = 0.f;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              const float mw = s_m[2 * ro + w];
              const float ew = mw == -INFINITY ? 0.f : __expf(mw - M);
              const float* pa = s_acc + (2 * ro + w) * 4 * 36;
              Lsum += ((pa[32] + pa[36 + 32]) + (pa[72 + 32] + pa[108 + 32])) * ew;
              num += ((pa[e] + pa[36 + e]) + (pa[72 + e] + pa[108 + e])) * ew;
            }
            o = num / Lsum;
          }
          const float o2 = dpp_f<DPP_QUAD_XOR1>(o);
          if (r < R && !(e & 1)) gstore(q.hop + HOP_B + r * 256 + (q.head * HD + e) / 2, ep0 + 4 * l + 2, pack_h2(o, o2));
        }
        if (q.cw == 1 && q.lane < 32) {                                    // K/V arena append of this step's k, v (see compute_role)
          const int ro = q.lane >> 4, which = (q.lane >> 3) & 1, e = 4 * (q.lane & 7), r = 2 * ro + q.half;
          if (r < R && st_active(q)[r]) {
            const int pos = st_kvlen(q)[r];
            if (pos < a.smax)
              *(h4*)(const_cast<_Float16*>(kv_row_base(l, which, r)) + (size_t)pos * HD + e) =
                  *(const h4*)((const _Float16*)(smem + L_QKV) + (ro * 3 + 1 + which) * HD + e);
          }
        }
      }
      // ================= P2: out-projection
      MG_BAR();                                                             // AT copy 0 holds quad 0's attention output
      relaunder(q);
      wload(wC, pm_src(l, WI_P2));
      for (int qd = 0; qd < nq; ++qd) {
        if (qd > 0) {
          MG_BAR();                                                         // AT copy qd & 1 holds the attention output
        }
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        const f4 p_bias = *(const f4*)(lp + FP_OUTB + 16 * q.member + 4 * (q.lane >> 4));
        gemm_chunk<1>(q, wB, (const _Float16*)(smem + L_AT) + (qd & 1) * RMAX * XS_LD, XS_LD, q.cw, q.cw);
        if (compute_barrier(q, cgen)) return;
        if (q.cw == 0) {
          const f4* red = (const f4*)(smem + L_RED);
          const int r = q.lane & 15;
          if (r < R) {
            const f4 v0 = red[red_idx(0, q.lane)], v1 = red[red_idx(1, q.lane)], v2 = red[red_idx(2, q.lane)], v3 = red[red_idx(3, q.lane)];
            f4 v = v0;
            v += v1; v += v2; v += v3;
            v += p_bias;
            v += *rs_at(smem, 0, qd, q.lane);
            *(f4*)((float*)(smem + L_STAGE) + r * 16 + 4 * (q.lane >> 4)) = v;
          }
          const bool stall = a.test_stall && q.member == a.test_stall - 1 && q.group == 0 && s == 2 && l == 3 && qd == 0;
          if (q.lane < R * 16 && !stall)
            gstore(q.hop + HOP_C + (q.lane >> 4) * 512 + 16 * q.member + (q.lane & 15), ep0 + 4 * l + 3,
                   __float_as_uint(((const float*)(smem + L_STAGE))[q.lane]));
        }
      }
      // ================= P3: FFN1 + ReLU
      MG_BAR();                                                             // XS copy 0 holds LN1(y1) of quad 0
      relaunder(q);
      wload(wD, pm_src(l, WI_P2 + WI_P3));
      for (int qd = 0; qd < nq; ++qd) {
        if (qd > 0) {
          MG_BAR();                                                         // XS copy qd & 1 holds LN1(y1)
        }
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        const f4 p_bias = *(const f4*)(lp + FP_B1 + 64 * q.member + 16 * q.cw + 4 * (q.lane >> 4));
        f4 v = gemm_tile_k512(q, wC, (const _Float16*)(smem + L_XS) + (qd & 1) * RMAX * XS_LD, XS_LD);
        {
          // tile cw of the member's four, published by the wave that computed it (see compute_role)
          const int r = q.lane & 15;
          unsigned* stage = (unsigned*)(smem + L_STAGE) + q.cw * 64;
          v += p_bias;
          if (r < R) {
            const int w0 = r * 8 + 2 * (q.lane >> 4);
            stage[w0] = pack_h2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
            stage[w0 + 1] = pack_h2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
          }
          if (q.lane < R * 8)
            gstore(q.hop + HOP_D + (q.lane >> 3) * 1024 + 32 * q.member + 8 * q.cw + (q.lane & 7), ep0 + 4 * l + 4, stage[q.lane]);
        }
      }
      // ================= P4: FFN2
      MG_BAR();                                                             // HS copy 0 holds quad 0's FFN hidden
      relaunder(q);
      if (l + 1 < a.L) wload(wA0, p1_src(l + 1));                          // P1 has finished with both halves for every quad
      else wload(wA0, a.lpack + (((size_t)q.member * 4 + q.cw) * WI_LG * 512) / 8 + q.lane);
      wload(wA1, p1_src(l + 1 < a.L ? l + 1 : 0) + (size_t)(WI_P1 / 2) * 64);
      for (int qd = 0; qd < nq; ++qd) {
        if (qd > 0) {
          MG_BAR();                                                         // HS copy qd & 1 holds the FFN hidden
        }
        relaunder(q);
        set_quad(q, qd);
        const int R = q.R;
        q.hop = hop_slot(q, s, l);
        const f4 p_bias = *(const f4*)(lp + FP_B2 + 16 * q.member + 4 * (q.lane >> 4));
        {
          const int rowl = q.lane & 15, kg = q.lane >> 4;
          f4* red = (f4*)(smem + L_RED);
          f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
          const _Float16* bp0 = (const _Float16*)(smem + L_HS) + ((qd & 1) * RMAX + (rowl & (RMAX - 1))) * HS_LD + (4 * q.cw) * 128 + 8 * kg;
          h8 bA[4], bB[4];                                   // chunk cc + 1's operands in flight under chunk cc's MFMAs (see compute_role)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) bA[ks] = *(const h8*)(bp0 + 32 * ks);
#pragma unroll
          for (int cc = 0; cc < 4; cc += 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bB[ks] = *(const h8*)(bp0 + (cc + 1) * 128 + 32 * ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wD[cc * 4 + ks], bA[ks], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (cc + 2 < 4) {
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) bA[ks] = *(const h8*)(bp0 + (cc + 2) * 128 + 32 * ks);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wD[(cc + 1) * 4 + ks], bB[ks], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (rowl < RMAX) red[red_idx(q.cw, q.lane)] = acc;
        }
        if (compute_barrier(q, cgen)) return;
        if (q.cw == 0) {
          const f4* red = (const f4*)(smem + L_RED);
          const int r = q.lane & 15;
          if (r < R) {
            const f4 v0 = red[red_idx(0, q.lane)], v1 = red[red_idx(1, q.lane)], v2 = red[red_idx(2, q.lane)], v3 = red[red_idx(3, q.lane)];
            f4 v = v0;
            v += v1; v += v2; v += v3;
            v += p_bias;
            v += *rs_at(smem, 1, qd, q.lane);
            *(f4*)((float*)(smem + L_STAGE) + r * 16 + 4 * (q.lane >> 4)) = v;
          }
          const unsigned ep = l + 1 < a.L ? ep0 + 4 * (l + 1) + 1 : ep0 + 4 * a.L + 1;
          if (q.lane < R * 16)
            gstore(hop_slot(q, s, l + 1) + HOP_A + (q.lane >> 4) * 512 + 16 * q.member + (q.lane & 15), ep,
                   __float_as_uint(((const float*)(smem + L_STAGE))[q.lane]));
        }
      }
    }
    // ================= tail: logits
    const unsigned epE = ep0 + 4 * a.L + 2;
    for (int qd = 0; qd < nq; ++qd) {
      relaunder(q);
      set_quad(q, qd);
      const int R = q.R;
      q.hop = hop_slot(q, s, a.L);
      MG_BAR();                                                             // XS copy qd & 1 holds LN2(y) of the last layer
      gemm_chunk<3>(q, wA0, (const _Float16*)(smem + L_XS) + (qd & 1) * RMAX * XS_LD, XS_LD, q.cw, q.cw * 3);
      if (compute_barrier(q, cgen)) return;
      if (q.cw < 3) {
        const f4* red = (const f4*)(smem + L_RED);
        const int t = q.cw, r = q.lane & 15;
        const int tile = t == 0 ? q.member : (t == 1 ? q.member + 32 : 64);
        float* stage = (float*)(smem + L_STAGE) + q.cw * 256;
        if (r < R) {
          f4 v = red[red_idx(0 * 3 + t, q.lane)];
          v += red[red_idx(1 * 3 + t, q.lane)]; v += red[red_idx(2 * 3 + t, q.lane)]; v += red[red_idx(3 * 3 + t, q.lane)];
          *(f4*)(stage + r * 16 + 4 * (q.lane >> 4)) = v;
        }
        if (q.lane < R * 16 && (t < 2 || q.member == 0) && 16 * tile + (q.lane & 15) < V)
          gstore(q.hop + HOP_E + (q.lane >> 4) * VPAD + 16 * tile + (q.lane & 15), epE, __float_as_uint(stage[q.lane]));
      }
    }
    MG_BAR();                                                               // end of the step's GEMMs
    wload(wA0, p1_src(0));                                                 // the next step's first slice follows the logits slice's last reader
  }
}

// MODE 0: one quad per group (B <= 32); 1: several quads, one after the other in every phase (kept for A/B: GSV_MEGA_QUADS=seq);
// 2: several quads, pipelined (default for 32 < B <= 128)
// PROF: the in-kernel stamps (tools/mega_prof.py) are compiled in only for measurement launches -- as run-time tests they
// cost ~100 scalar instructions per layer and wave
template <int MODE, bool PROF>
__global__ __launch_bounds__(MG_THREADS, 1) void t2s_mega_kernel(MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Ctx c;
  c.smem = smem;
  const int tid = threadIdx.x;
  c.lane = tid & 63; c.wave = __builtin_amdgcn_readfirstlane(tid >> 6); c.comm = c.wave < 4; c.cw = c.wave & 3; c.tid_c = tid - 256;
  if (a.map_shared == 2) {
    // Roles dealt by the XCD a workgroup actually RUNS on (HW_REG_XCC_ID) and its arrival order there: the 8 workgroups that
    // stream the same weight slice (one per row group) share an L2 whatever the dispatcher did.  blockIdx % 8 is only a label
    // of which blocks share an XCD under strict round-robin placement; when a CU is still busy at launch the dispatcher
    // deviates from it and a slice is then pulled into several L2s (the ~6 % slower of the two modes seen run to run).
    // Speed only: any assignment of the 256 (group, member) roles is correct.  32 workgroups of this footprint fit an XCD
    // (one per CU); a 33rd ticket means the device is not what the census assumed: error exit, the host falls back.
    if (tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      const unsigned t = __hip_atomic_fetch_add((gu32*)a.err + 8 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ((volatile int*)smem)[0] = (int)xcc;
      ((volatile int*)smem)[1] = (int)t;
    }
    __syncthreads();
    const int xcd = __builtin_amdgcn_readfirstlane(((volatile int*)smem)[0]), t = __builtin_amdgcn_readfirstlane(((volatile int*)smem)[1]);
    __syncthreads();                                 // (uniform values: keep them, and everything derived from them, in SGPRs)
    if (t >= 32) {
      if (tid == 0) { __hip_atomic_store((gu32*)a.err + 3, 0x50u, RLX_AGENT); __hip_atomic_store((gu32*)a.err, 1u, RLX_AGENT); }
      return;
    }
    c.member = xcd * 4 + (t >> 3); c.group = t & 7;
  } else if (a.map_shared) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; c.member = xcd * 4 + (slot >> 3); c.group = slot & 7; }
  else { c.group = blockIdx.x & 7; c.member = blockIdx.x >> 3; }
  c.head = c.member >> 1; c.half = c.member & 1;
  c.member = __builtin_amdgcn_readfirstlane(c.member); c.group = __builtin_amdgcn_readfirstlane(c.group);
  c.head = c.member >> 1; c.half = c.member & 1;
  c.Rtot = a.B > c.group ? (a.B - c.group + MG_GROUPS - 1) / MG_GROUPS : 0;     // rows of this group: batch rows group, group + 8, ...
  if (c.Rtot == 0) return;
  set_quad(c, 0);
  c.hop_base = (gu64*)a.hop + (size_t)c.group * HOP_GROUP_ALL;
  c.hop = c.hop_base;
  c.err = (gu32*)a.err;
  c.prof = a.prof ? a.prof + ((size_t)blockIdx.x * 8 + c.wave) * 32 : nullptr;
  c.prof_on = false;
  c.hint_miss16 = (a.hint_mask >> 8) & 31;
  c.hint_pipe = (a.hint_mask >> 4) & 1;
  c.hint_spread = (a.hint_mask >> 7) & 1;
  const int lane = c.lane, Rtot = c.Rtot;
  const StepParams sp = *a.sp;

  // ---- init: zero the activation images, load the row state, build the samplers' seen-map ----
  for (int i = tid; i < L_RED / 4; i += MG_THREADS) ((unsigned*)smem)[i] = 0u;
  for (int i = tid; i < (L_TOTAL - L_KV) / 4; i += MG_THREADS) ((unsigned*)(smem + L_KV))[i] = 0u;   // attention_part relies on finite image cells
  if (tid < 3 * ST_N + 16) ((int*)(smem + L_ST))[tid] = 0;
  __syncthreads();
  if (tid < Rtot) {                                   // local row tid = 4 * quad + r
    const int b = c.group + MG_GROUPS * tid;
    ((int*)(smem + L_ST))[tid] = a.active[b];
    ((int*)(smem + L_ST))[ST_N + tid] = a.kv_len[b];
    ((int*)(smem + L_ST))[2 * ST_N + tid] = a.step_ctr[b];
  }
  if (c.member < Rtot && c.wave == 0) {
    unsigned char* seen = smem + L_SEEN;
    for (int v = lane; v < VPAD; v += 64) seen[v] = 0;
    const int b = c.group + MG_GROUPS * c.member;
    const int prev_len = sp.P + a.step_ctr[b];
    const int* yrow = a.ytok + (size_t)b * a.ycap;
    __builtin_amdgcn_s_waitcnt(0);
    for (int t = lane; t < prev_len; t += 64) {
      const int tok = yrow[t];
      if (tok >= 0 && tok < a.V) seen[tok] = 1;
    }
  }
  __syncthreads();
  if (group_done(c)) return;
  if constexpr (MODE == 2) {
    if (c.comm) comm_role_pipe(a, c, sp);
    else compute_role_pipe(a, c);
  } else {
    if (c.comm) comm_role<MODE == 1, PROF>(a, c, sp);
    else compute_role<MODE == 1, PROF>(a, c);
  }
}

// census: are 256 workgroups of the engine's footprint co-resident?  Every workgroup arrives on a counter and waits
// (bounded) for all of them; a workgroup that gives up reports it.
__global__ __launch_bounds__(MG_THREADS, 1) void mega_census_kernel(unsigned* ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  smem[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)MG_NWG) {
      if (++spins > (1u << 20)) { __hip_atomic_fetch_add(ws + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

}  // namespace

bool mega_shape_ok(int dim, int n_head, int ffn, int vocab) { return dim == D && n_head == NH && ffn == FF && vocab >= 64 && vocab <= VPAD; }
size_t mega_layer_pack_halfs() { return LAYER_HALFS; }
size_t mega_logits_pack_halfs() { return LOGIT_HALFS; }
size_t mega_hop_bytes(int ring) { return (size_t)(ring > 0 ? ring : 1) * MG_GROUPS * HOP_GROUP_ALL * 8; }

// fragment of one KiB-instruction: lane ln holds W[row0 + (ln & 15)][k0 + 8 (ln >> 4) .. +8]
static void pack_frag(const float* w, int ldw, int row0, int nrows, int k0, _Float16* dst) {
  for (int ln = 0; ln < 64; ++ln) {
    const int row = row0 + (ln & 15), k = k0 + 8 * (ln >> 4);
    for (int e = 0; e < 8; ++e) dst[ln * 8 + e] = row < nrows ? (_Float16)w[(size_t)row * ldw + k + e] : (_Float16)0.f;
  }
}

void mega_pack_layer(const float* qkv_w, const float* out_w, const float* w1, const float* w2, _Float16* dst) {
  // P1 by head: tiles q0 q1 k0 k1 v0 v1, wave = K chunk of 128, 4 k-steps of 32
  for (int h = 0; h < NH; ++h)
    for (int w = 0; w < 4; ++w)
      for (int t = 0; t < 6; ++t)
        for (int ks = 0; ks < 4; ++ks)
          pack_frag(qkv_w, D, (t >> 1) * D + h * HD + 16 * (t & 1), 3 * D, 128 * w + 32 * ks,
                    dst + (((size_t)h * 4 + w) * WI_P1 + t * 4 + ks) * 512);
  _Float16* pm = dst + P1_HALFS;
  const int per = WI_P2 + WI_P3 + WI_P4;
  for (int j = 0; j < MG_MEMBERS; ++j)
    for (int w = 0; w < 4; ++w) {
      _Float16* base = pm + ((size_t)j * 4 + w) * per * 512;
      for (int ks = 0; ks < 4; ++ks) pack_frag(out_w, D, 16 * j, D, 128 * w + 32 * ks, base + (size_t)ks * 512);
      // FFN1: wave w owns tile w (columns 64 j + 16 w, +16) over the whole K: slot 4 c + ks = K chunk c, k-step ks
      for (int c = 0; c < 4; ++c)
        for (int ks = 0; ks < 4; ++ks)
          pack_frag(w1, D, 64 * j + 16 * w, FF, 128 * c + 32 * ks, base + (size_t)(WI_P2 + c * 4 + ks) * 512);
      for (int cc = 0; cc < 4; ++cc)
        for (int ks = 0; ks < 4; ++ks)
          pack_frag(w2, FF, 16 * j, D, 128 * (4 * w + cc) + 32 * ks, base + (size_t)(WI_P2 + WI_P3 + cc * 4 + ks) * 512);
    }
}

void mega_pack_logits(const float* pred_w, int V, _Float16* dst) {
  for (int j = 0; j < MG_MEMBERS; ++j)
    for (int w = 0; w < 4; ++w)
      for (int t = 0; t < 3; ++t) {
        const int tile = t == 0 ? j : (t == 1 ? j + 32 : 64);
        for (int ks = 0; ks < 4; ++ks)
          pack_frag(pred_w, D, 16 * tile, V, 128 * w + 32 * ks, dst + (((size_t)j * 4 + w) * WI_LG + t * 4 + ks) * 512);
      }
}

int mega_census(hipStream_t s, unsigned* d_scratch, unsigned* h_pinned) {
  GSV_HIP(hipFuncSetAttribute((const void*)mega_census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL));
  GSV_HIP(hipMemsetAsync(d_scratch, 0, 16, s));
  hipLaunchKernelGGL(mega_census_kernel, dim3(MG_NWG), dim3(MG_THREADS), L_TOTAL, s, d_scratch);
  GSV_HIP(hipGetLastError());
  GSV_HIP(hipMemcpyAsync(h_pinned, d_scratch, 8, hipMemcpyDeviceToHost, s));
  GSV_HIP(hipStreamSynchronize(s));
  return (h_pinned[0] == (unsigned)MG_NWG && h_pinned[1] == 0u) ? 1 : 0;
}

int launch_t2s_mega(const MegaArgs& a, hipStream_t s) {
  // every launch: the attribute belongs to the (function, device) pair, and TTS.set_device may have moved the handle's owner
  // to another GPU of the process since the last launch (mega_census does the same)
  static const bool seq_quads = getenv("GSV_MEGA_QUADS") && !strcmp(getenv("GSV_MEGA_QUADS"), "seq");
#define GSV_MEGA_LAUNCH(MODE, PROF)                                                                                              \
  do {                                                                                                                          \
    GSV_HIP(hipFuncSetAttribute((const void*)t2s_mega_kernel<MODE, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL)); \
    hipLaunchKernelGGL((t2s_mega_kernel<MODE, PROF>), dim3(MG_NWG), dim3(MG_THREADS), L_TOTAL, s, a);                            \
  } while (0)
  const bool prof = a.prof != nullptr;                // measurement launches (tools/mega_prof.py): stamps compiled in
  if (a.B > RMAX * MG_GROUPS && seq_quads) {          // more than one quad per group, one after the other (A/B)
    if (prof) GSV_MEGA_LAUNCH(1, true); else GSV_MEGA_LAUNCH(1, false);
  } else if (a.B > RMAX * MG_GROUPS) {                // pipelined quads (no stamps)
    GSV_MEGA_LAUNCH(2, false);
  } else {
    if (prof) GSV_MEGA_LAUNCH(0, true); else GSV_MEGA_LAUNCH(0, false);
  }
#undef GSV_MEGA_LAUNCH
  GSV_HIP(hipGetLastError());
  return GSV_OK;
}

}  // namespace gsv

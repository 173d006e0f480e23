// Persistent AR decode engine (H4 + H5 in ONE launch for all steps of a decode call), fp16, v1/v2 decoder shape
// (d = 512, 16 heads x 32, FFN 2048, vocab <= 1088).  Interface between t2s.hip (engine state, prefill, step 0) and
// t2s_mega.hip (weight packing, the kernel, its launch).  See the header of t2s_mega.hip for the design.
#pragma once
#include <vector>

#include "common.h"
#include "t2s_sample.h"

namespace gsv {

struct MegaState {        // owned by the t2s handle
  void* wpack = nullptr;          // packed fp16 weights  [layer][member][wave][60 KiB-instructions][64 lanes][8 halfs]
  void* lpack = nullptr;          // packed logits weights [member][wave][12][64][8]
  float* fpack = nullptr;         // [n_layer][6656] fp32: qkv_b | out_b | b1 | b2 | n1w | n1b | n2w | n2b
  unsigned long long* hop = nullptr;   // hop granule buffers, all groups
  size_t hop_bytes = 0;
  int ring = 1;                   // hop buffer sets
  unsigned launch_gen = 0;        // bumps every launch: epochs = launch_gen << 20 + ...
  unsigned* err = nullptr;        // [4]: timeout word, epoch, block, code
  unsigned* h_err = nullptr;      // pinned host copy
  int census = -1;                // -1 not run, 0 failed (mega disabled), 1 all 256 workgroups co-resident
  int* snap = nullptr;            // row state saved in front of a launch: [kv_len | active | step | n_active(4)][out_len]
  unsigned last_err[3] = {0, 0, 0};   // epoch, workgroup, hop code of the last timeout
  int fallbacks = 0;              // launches that ended in a hand-off timeout and were re-run on the launch path
  bool ready = false;
};

struct MegaArgs {
  const h8* wpack; const h8* lpack; const float* fpack;
  _Float16* kv; unsigned long long kv_layer_stride; int smax;     // KV arena [layer][k|v][row][head][pos][32]
  int *kv_len, *active, *step_ctr, *n_active, *ytok; int ycap;
  const StepParams* sp;
  const float *e_audio, *pe; float alpha_a;
  const float* ybuf;              // [B][512] fp32: input embedding of the first step (written by the step-0 tail)
  float* logits_out;              // [B][V] fp32: each row's logits of its last sampled step (test hook gsv_t2s_debug_logits)
  unsigned long long* hop; unsigned* err;
  int B, L, V, nsteps;
  unsigned long long* prof;       // optional [256 workgroups][8 waves][32] s_memtime stamps of one (step, layer); null = off
  int prof_step, prof_layer, prof_quad;
  int hint_mask;                  // hops that poll a hint line before the full pass: bit 0 A, 1 B, 2 C, 3 D; bit 4: two polls in flight (sweep2); bit 7: spread hint lines; bits 8-12: miss threshold (sweep2)
  int map_shared;                 // 1: workgroups reading the same weight slice share an XCD (default), 0: group = XCD
  int ring;                       // hop buffer sets used round-robin over (step, layer); > 1 enables L2-shared payload reads
  unsigned ep_base;               // launch generation << 20: epochs never repeat between launches (stale cached lines cannot match)
  int test_stall;                 // tests only (gsv_t2s_debug_stall): member test_stall - 1 of group 0 skips one publish; 0 = off
};

// shape gate: the persistent engine is specialised for the v1/v2 decoder
bool mega_shape_ok(int dim, int n_head, int ffn, int vocab);
constexpr int MEGA_MAX_B = 128;       // 8 row groups x 4 quads x 4 rows; B <= 32 is the single-quad kernel

// host: pack the fp32 staged weights of one layer into the engine's load order (fp16)
void mega_pack_layer(const float* qkv_w, const float* out_w, const float* w1, const float* w2, _Float16* dst);
void mega_pack_logits(const float* pred_w, int V, _Float16* dst);
size_t mega_layer_pack_halfs();
size_t mega_logits_pack_halfs();
size_t mega_hop_bytes(int ring);
constexpr int MEGA_FP_LAYER = 6656;      // floats per layer in fpack

int mega_census(hipStream_t s, unsigned* d_scratch, unsigned* h_pinned);   // 1 ok, 0 not co-resident, <0 error
int launch_t2s_mega(const MegaArgs& a, hipStream_t s);

}  // namespace gsv

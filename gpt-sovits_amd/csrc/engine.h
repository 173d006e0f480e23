// Engine context shared by the SoVITS decoder (vits.hip), the vocoders and the flow-matching DiT (cfm.hip):
// weight staging / upload, workspace arena, and the conv / attention launch helpers.
#pragma once
#include <math.h>
#include <map>
#include <string>
#include <vector>

#include "common.h"

struct Conv {
  void* w = nullptr;     // T [cout][taps*cin]
  float* b = nullptr;    // fp32 [cout_real] or null
  int cin = 0, cout = 0, taps = 1;
  int ups_u = 0, ups_pad = 0, ups_cout = 0;
};

struct AttnLayerW { Conv qkv, o; float *rel_k = nullptr, *rel_v = nullptr; float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; Conv f1, f2; };
struct WNW { Conv in[4], res[4], skip[4]; float* in_bias_eff[4] = {nullptr, nullptr, nullptr, nullptr}; Conv cond; };
struct FlowW { Conv pre, post; WNW wn; };

struct Buf { void* p = nullptr; size_t cap = 0; };

struct gsv_vits {
  gsv_vits_config cfg;
  int dtype;
  bool finalized = false, has_ref = false;
  std::map<std::string, std::vector<float>> staged;
  std::vector<void*> allocs;
  // weights
  Conv ssl_proj_enc, proj, c_pre, text_pre, c_post, mq, mkv, mo;
  std::vector<AttnLayerW> enc_ssl, enc_text, enc2;
  float *text_emb = nullptr, *codebook = nullptr, *code_ee = nullptr;
  Conv top_ssl_proj;
  void* codebook_t = nullptr;
  FlowW flows[4];
  Conv conv_pre, conv_post, cond;
  float* conv_pre_bias_eff = nullptr;
  std::vector<Conv> ups;
  std::vector<Conv> rb1, rb2;  // [stage][j][c]
  // v3 / v4: bridge + wns1 (Encoder with an 8-layer WN)
  Conv bridge, w1_pre, w1_proj, w1_cond;
  std::vector<Conv> w1_in, w1_res;
  std::vector<float*> w1_in_bias_eff;
  // v2Pro: speaker-verification conditioning
  Conv sv_emb, ge_to512;
  float *prelu_w = nullptr, *ge_ref = nullptr, *sv_proj = nullptr, *ge512 = nullptr;
  void* sv_t = nullptr;
  // ref_enc
  Conv r_sp0, r_sp3, r_t0, r_t1, r_qkv, r_fc, r_out;
  float* ge = nullptr;         // fp32 [gin]
  void* ge_t = nullptr;        // T [gin]
  float* mo_bias_eff = nullptr;
  // workspace
  std::map<std::string, Buf> bufs;
  // last decode bookkeeping
  int lastF = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float last_total_ms = 0.f, last_gen_ms = 0.f;
};

struct ConvOpt {
  int dil = 1, pad = -1, stride = 1;
  int pre_act = gsv::ACT_NONE; float pre_slope = 0.1f;
  int post_act = gsv::ACT_NONE; float scale = 1.f; int accumulate = 0;
  int out_f32 = 0; const void* res = nullptr; int res_f32 = 0; int ldr = 0;
  int ldy = 0, y_col0 = 0;
  const float* bias_override = nullptr; bool no_bias = false;
  const float* gate = nullptr;   // per-output-channel gate: y = ((W x + b) * gate + res) * scale
  int w_nt = 0;                  // stream the weights non-temporal (ConvArgs::w_nt)
  int w_row0 = 0, cout = -1;   // use a row slice of the weight matrix
  // DiT QKV projection: rotary embedding + transposed V from the GEMM's epilogue (ConvArgs::vt_out ...)
  void* vt_out = nullptr; int vt_col0 = 0, vt_ld = 0;
  const float* rope_cs = nullptr; int rope_half = 0, rope_q0 = 0, rope_k0 = 0;
};

#define GSV_DISPATCH(h, call_f16, call_f32) \
  do { if ((h)->dtype == GSV_F16) { call_f16; } else { call_f32; } } while (0)

namespace gsveng {

inline int nblk(long long n, int b = 256) { return (int)((n + b - 1) / b); }
inline size_t esz(const gsv_vits* h) { return gsv::dt_size(h->dtype); }

int dalloc(gsv_vits* h, void** p, size_t bytes);
int up_f32(gsv_vits* h, const float* v, size_t n, float** out);
int up_t(gsv_vits* h, const std::vector<float>& v, void** out);
// staged tensor (or a folded weight_g / weight_v pair) as an fp32 host vector
bool fetch(gsv_vits* h, const std::string& name, size_t n, int dim0, std::vector<float>& out);
int make_conv(gsv_vits* h, const std::string& name, int cout, int cin, int k, bool bias, Conv* c);
int make_conv_padded(gsv_vits* h, const std::string& name, int cout, int cin, int cin_pad, int k, bool bias, Conv* c);
int make_stacked(gsv_vits* h, const std::vector<std::string>& names, int cout_each, int cin, Conv* c);
int make_ups(gsv_vits* h, const std::string& name, int cin, int cout, int k, int u, Conv* c);
int make_vec(gsv_vits* h, const std::string& name, size_t n, float** out);
int need(gsv_vits* h, const char* name, size_t bytes, void** out);
int conv(gsv_vits* h, hipStream_t s, const Conv& c, const void* x, int ldx, int T_in, void* y, int T_out, const ConvOpt& o);
int attention(gsv_vits* h, hipStream_t s, const void* q, int ldq, int qcol0, const void* kv, int ldkv, int kcol0, int vcol0,
              int Tq, int Tk, int nh, int kc, float scale, const float* rel_k, const float* rel_v, void* out, int ldo);
void free_ctx(gsv_vits* h);

}  // namespace gsveng

/* gsv.h -- C ABI of libgsv_hip.so, the MI355X (gfx950) GPT-SoVITS synthesis hot path.
 *
 * Drop-in boundary.  Each group of entry points replaces one seam
 * of the reference (paths under /root/reference/GPT_SoVITS):
 *
 *   gsv_t2s_*    <- Text2SemanticDecoder.infer_panel_batch_infer / infer_panel_naive
 *                   AR/models/t2s_model.py:583-779, 814-918 (called from
 *                   TTS_infer_pack/TTS.py:1215-1227 and inference_webui.py:878);
 *                   sampling AR/models/utils.py:140-199
 *   gsv_vits_*   <- SynthesizerTrn.decode / extract_latent, module/models.py:961-1010
 *                   (called from TTS_infer_pack/TTS.py:1271, 818; inference_webui.py:920)
 *   gsv_aa_act_forward <- anti_alias_activation_cuda.forward, the reference's only native
 *                   FFI: BigVGAN/alias_free_activation/cuda/anti_alias_activation.cpp:19-22,
 *                   anti_alias_activation_cuda.cu:212-246
 *   gsv_op_*     <- single kernels exposed for parity tests
 *
 * Conventions: plain C, no torch types.  Pointers marked [dev] are device (HBM) pointers owned
 * by the caller; [host] are host pointers.  `stream` is a hipStream_t passed as void*.
 * Every function returns GSV_OK (0) or a negative error code; gsv_last_error() returns a
 * thread-local message.  Handles are re-entrant per handle: one handle, one stream at a time.
 * There is no CPU fallback: every entry point fails with GSV_ERR_HIP if no gfx950 device exists.
 */
#ifndef GSV_H_
#define GSV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSV_OK 0
#define GSV_ERR_ARG (-1)
#define GSV_ERR_HIP (-2)
#define GSV_ERR_STATE (-3)

#define GSV_F32 0 /* fp32 storage, exact-f32 MFMA: parity mode                      */
#define GSV_F16 1 /* fp16 storage, fp32 accumulation: production mode (is_half=True) */

typedef void* gsv_stream_t;

int gsv_init(int device);
const char* gsv_last_error(void);
int gsv_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * AR semantic-token decoder (H1-H5)
 * ------------------------------------------------------------------------------------- */
typedef struct gsv_t2s gsv_t2s_t;

typedef struct {
  int n_layer;        /* 24 */
  int dim;            /* 512 */
  int n_head;         /* 16 */
  int ffn_dim;        /* 4*dim (t2s_model.py:304) */
  int vocab;          /* 1025, EOS = vocab-1 */
  int phoneme_vocab;  /* 732 (v2) / 512 (v1) */
  int bert_dim;       /* 1024 */
} gsv_t2s_config;

typedef struct {
  int top_k;                /* <=0: disabled */
  float top_p;              /* >=1: disabled */
  float temperature;
  float repetition_penalty;
  int early_stop_num;       /* -1: none; else stop once generated > early_stop_num (t2s_model.py:747) */
  int eos_mask_steps;       /* EOS column dropped while step < this: 1 for the batched loop
                               (t2s_model.py:708-710), 11 for infer_panel_naive (:888-889) */
  int max_steps;            /* reference hard cap 1500 (t2s_model.py:701) */
  uint64_t seed;            /* counter-RNG key for the Exp(1) race when no noise is injected */
} gsv_sampling_params;

int gsv_t2s_create(const gsv_t2s_config* cfg, int dtype, int max_batch, int max_seq, gsv_t2s_t** out);
void gsv_t2s_destroy(gsv_t2s_t* h);
/* name = reference state-dict key without the "model." prefix;
 * data [host] fp32, numel elements. */
int gsv_t2s_load_tensor(gsv_t2s_t* h, const char* name, const float* data, int64_t numel);
int gsv_t2s_finalize(gsv_t2s_t* h);

/* Prefill (H2+H3): phones [dev] int32 packed, phone_lens [host] int32[B], bert [dev] fp32
 * [sum(X_b)][bert_dim] row-major (token-major; NULL = all zeros, the non-zh path), prompts [dev]
 * int32 [B][P]; P == 0 (prompts may be NULL) is the reference's prompt-free mode (t2s_model.py:849-856: the audio sequence
 * starts empty at position 0).  Leaves the KV cache and per-row state ready for gsv_t2s_decode. */
int gsv_t2s_prefill(gsv_t2s_t* h, const int32_t* phones, const int32_t* phone_lens, int B,
                    const float* bert, const int32_t* prompts, int P, gsv_stream_t stream);

/* Decode loop (H4+H5).  noise [dev] fp32 Exp(1) draws [max_steps][noise_rows][vocab] or NULL
 * (noise_rows is 1 = shared by all rows, or B).  out_tokens [dev] int32 [B][max_steps]: generated
 * tokens (the finishing EOS / overflow token is not counted); out_len [dev] int32 [B] = the
 * reference's idx_list.  steps_run [host] receives the number of steps executed. */
int gsv_t2s_decode(gsv_t2s_t* h, const gsv_sampling_params* sp, const float* noise, int noise_rows,
                   int32_t* out_tokens, int32_t* out_len, int* steps_run, gsv_stream_t stream);

/* How the last gsv_t2s_decode call ran: mode 1 = the persistent engine (csrc/t2s_mega.hip: fp16, d=512/16 heads/FFN 2048,
 * B <= 128 -- up to four quads of rows per row group, B <= 32 is one quad; all steps after step 0 in ONE launch, hand-offs on the chip), mode 0 = one hipGraph of 122 launches per step
 * (fp32, other shapes, GSV_T2S_NO_MEGA=1, or a device on which the engine's 256 workgroups are not co-resident);
 * device_ms = HIP-event time of the persistent launch, steps = decode steps it covered (mode 1 only). */
int gsv_t2s_decode_info(gsv_t2s_t* h, int* mode, float* device_ms, int* steps);
/* A/B switch inside one process: on = 0 makes later decode calls of this handle use the launch-per-phase step */
int gsv_t2s_set_mega(gsv_t2s_t* h, int on);
/* Robustness report.  A persistent launch whose hand-off timed out (a member workgroup was not running: its CU was held
 * by another kernel) does not fail the request: the row state is restored, the batch is re-run on the launch-per-phase
 * step and the handle stops using the engine.  engine_available = 0 after that (or when the engine was never built),
 * fallbacks = launches that ended that way, last_error3 [host] = {epoch, workgroup, hop code} of the last timeout. */
int gsv_t2s_engine_stats(gsv_t2s_t* h, int* engine_available, int* fallbacks, unsigned* last_error3);
/* Parity hooks for the NEXT gsv_t2s_decode call only (both decode paths, every dtype; not a reference feature -- SURVEY.md
 * section 7 asks for a teacher-forced logit comparison of the fp16 engine):
 *   force_tokens [dev] int32 [B][max_steps] or NULL: step s of row b continues with force_tokens[b][s] instead of the
 *     token it sampled (EOS / early-stop bookkeeping then sees the forced token);
 *   logits_dump [dev] fp32 [max_steps][B][vocab] or NULL: raw logits (before the repetition penalty) of every step run;
 *   drawn_dump [dev] int32 [max_steps][B][2] or NULL: (token the sampler drew, argmax of the penalised logits) of every
 *     step run, recorded BEFORE forcing -- lets a test replay the engine's sampling (noise indexing, counter RNG, repetition
 *     bookkeeping) step by step with gsv_op_sample / the oracle on the dumped logits. */
int gsv_t2s_set_debug(gsv_t2s_t* h, const int32_t* force_tokens, float* logits_dump, int32_t* drawn_dump);
/* test hook: member `member` (0..31) of row group 0 skips ONE hand-off publish in the next persistent launch, which must then
 * end through its bounded waits (never hang), be re-run on the launch-per-phase step and be counted by gsv_t2s_engine_stats */
int gsv_t2s_debug_stall(gsv_t2s_t* h, int member);
/* test hook: logits of each row's last sampled step [dev] fp32 [B][vocab] (either decode path) */
int gsv_t2s_debug_logits(gsv_t2s_t* h, float* out, gsv_stream_t stream);
/* per-kernel timing of the decode step: average device time (ms) of one step over `iters`
 * replays at the current cache length, and of the decode-attention kernel alone. */
int gsv_t2s_time_step(gsv_t2s_t* h, int iters, float* step_ms, float* attn_ms, gsv_stream_t stream);
/* measurement hook: B rows with kv_len cached positions each (zeroed cache), for kernel timing sweeps */
int gsv_t2s_debug_set_state(gsv_t2s_t* h, int B, int kv_len);
/* algorithmic HBM bytes of one decode step at the current state */
int64_t gsv_t2s_step_bytes(gsv_t2s_t* h, int64_t* attn_bytes);

/* ---------------------------------------------------------------------------------------
 * SoVITS v2 waveform decoder (H6-H12)
 * ------------------------------------------------------------------------------------- */
typedef struct gsv_vits gsv_vits_t;

typedef struct {
  int inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size;
  int gin_channels, n_symbols, ssl_dim, n_bins, upsample_initial_channel;
  int n_ups;
  int up_rates[8];
  int up_kernels[8];
  int n_resblocks;          /* resblocks per stage (3) */
  int rb_kernels[4];
  int rb_dilations[4][3];
  int ref_bins;             /* 704 (v2): leading spectrogram bins fed to ref_enc */
  int flavor;               /* 0 = v1/v2 SynthesizerTrn (flow + generator); 1 = v3, 2 = v4 SynthesizerTrnV3 (bridge + wns1,
                               nearest x1.875 / x2; no flow / generator weights: the mel comes from gsv_cfm_inference) */
  int v2pro;                /* 1 = v2Pro / v2ProPlus conditioning (module/models.py:895-899): sv_emb 20480 -> gin, PReLU(gin),
                               ge_to512 for the MRTE; gin_channels may then differ from 512 */
} gsv_vits_config;

int gsv_vits_create(const gsv_vits_config* cfg, int dtype, gsv_vits_t** out);
void gsv_vits_destroy(gsv_vits_t* h);
int gsv_vits_load_tensor(gsv_vits_t* h, const char* name, const float* data, int64_t numel);
int gsv_vits_finalize(gsv_vits_t* h); /* folds weight-norm, repacks conv weights */

/* ge = mean over references of ref_enc(spec[:ref_bins]) (models.py:962-984).  specs [host] array
 * of n_refs [dev] fp32 pointers, each [bins][frames[i]] channels-first as the reference holds it. */
int gsv_vits_set_refer(gsv_vits_t* h, const float* const* specs, const int* frames, int bins, int n_refs,
                       gsv_stream_t stream);
/* codes [dev] int32 [T], phones [dev] int32 [L]; frames F = 2T when speed == 1, else int(2T/speed)+1
 * (linear interpolation of the encoder output, models.py:226-228); noise [dev] fp32 [inter][F]
 * channels-first (the randn_like draw of models.py:1000; NULL = counter RNG keyed by seed),
 * wav [dev] fp32 [F * prod(up_rates)]. */
int gsv_vits_decode(gsv_vits_t* h, const int32_t* codes, int T, const int32_t* phones, int L, const float* noise,
                    float noise_scale, double speed, uint64_t seed, float* wav, gsv_stream_t stream);
/* ssl [dev] fp32 [ssl_dim][T50] channels-first -> codes [dev] int32 [T50/2] */
int gsv_vits_extract_latent(gsv_vits_t* h, const float* ssl, int T50, int32_t* codes, gsv_stream_t stream);
/* test hook: copy a named intermediate of the last decode ("ge","m_p","logs_p","z","stage0".."stage4")
 * into out [dev] fp32 in channels-first [C][T] order; returns element count via *numel. */
int gsv_vits_debug_tensor(gsv_vits_t* h, const char* name, float* out, int64_t cap, int64_t* numel,
                          gsv_stream_t stream);
/* v3 / v4 (H14, SynthesizerTrnV3.decode_encp, module/models.py:1243-1267): codes / phones as in gsv_vits_decode ->
 * fea [dev] fp32 [512][F] (channels-first, what the reference returns; ge comes from gsv_vits_set_refer).
 * F = gsv_vits_encp_frames(h, T, speed) = floor(frames_after_speed * (1.875 | 2)); returns -1 on bad arguments. */
int gsv_vits_encp_frames(gsv_vits_t* h, int T, double speed);
int gsv_vits_decode_encp(gsv_vits_t* h, const int32_t* codes, int T, const int32_t* phones, int L, double speed, float* fea,
                         gsv_stream_t stream);
/* v2Pro / v2ProPlus (N4, module/models.py:971-975): as gsv_vits_set_refer, plus one speaker-verification embedding per
 * reference (sv_embs [host] array of n_refs [dev] fp32 pointers to 20480 values): ge_r = PReLU(ref_enc(spec_r) + sv_emb(sv_r)),
 * ge = mean_r ge_r; the MRTE receives ge_to512(ge). */
int gsv_vits_set_refer_sv(gsv_vits_t* h, const float* const* specs, const int* frames, int bins, const float* const* sv_embs,
                          int n_refs, gsv_stream_t stream);
/* per-kernel timing hooks for bench.py: device ms of the last decode's generator section */
int gsv_vits_last_timing(gsv_vits_t* h, float* total_ms, float* generator_ms);

/* ---------------------------------------------------------------------------------------
 * v3 / v4 vocoders (H15, H16): mel [in_channels][F] -> waveform.
 *   kind 0 = HiFi-GAN `Generator` as used for v4 (TTS_infer_pack/TTS.py:631-648, module/models.py:407-471:
 *            leaky-relu 0.1, conv_post bias, tanh); kind 1 = BigVGAN-v2 (BigVGAN/bigvgan.py:226-355:
 *            AMPBlock1 with anti-aliased Snake/SnakeBeta, clamp or tanh at the end).
 * Tensor names = the reference state-dict keys (weight-norm pairs or folded weights both accepted).
 * ------------------------------------------------------------------------------------- */
typedef struct gsv_vocoder gsv_vocoder_t;

typedef struct {
  int kind;
  int in_channels;               /* 100 mel bands */
  int upsample_initial_channel;
  int n_ups;
  int up_rates[8];
  int up_kernels[8];
  int n_resblocks;
  int rb_kernels[4];
  int rb_dilations[4][3];
  int bias_at_final;             /* conv_post bias */
  int tanh_at_final;             /* else clamp to [-1, 1] */
  int snake_logscale;            /* BigVGAN: alpha/beta stored in log scale */
} gsv_vocoder_config;

int gsv_vocoder_create(const gsv_vocoder_config* cfg, int dtype, gsv_vocoder_t** out);
void gsv_vocoder_destroy(gsv_vocoder_t* h);
int gsv_vocoder_load_tensor(gsv_vocoder_t* h, const char* name, const float* data, int64_t numel);
int gsv_vocoder_finalize(gsv_vocoder_t* h);
/* mel [dev] fp32 [in_channels][F] channels-first (as the reference passes it), wav [dev] fp32 [F * prod(up_rates)] */
int gsv_vocoder_forward(gsv_vocoder_t* h, const float* mel, int F, float* wav, gsv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * v3 / v4 flow-matching mel decoder (H14): CFM.inference (module/models.py:1027-1085, inference_cfg_rate = 0 as
 * every caller in the reference passes it) over the DiT estimator (f5_tts/model/backbones/dit.py:88-194).
 * Tensor names = the DiT state-dict keys (the part after "cfm.estimator." in a v3/v4 SoVITS checkpoint).
 * The rotary embedding follows x_transformers' published definition (un-vendored dependency: parity unpinned).
 * ------------------------------------------------------------------------------------- */
typedef struct gsv_cfm gsv_cfm_t;

typedef struct {
  int dim, depth, heads, dim_head, ff_mult;   /* 1024, 22, 16, 64, 2 (module/models.py:1219-1222) */
  int mel_dim, text_dim, conv_layers;         /* 100, 512, 4 */
} gsv_dit_config;

int gsv_cfm_create(const gsv_dit_config* cfg, int dtype, gsv_cfm_t** out);
void gsv_cfm_destroy(gsv_cfm_t* h);
int gsv_cfm_load_tensor(gsv_cfm_t* h, const char* name, const float* data, int64_t numel);
int gsv_cfm_finalize(gsv_cfm_t* h);
/* mu [dev] fp32 [B][T][text_dim] (the `fea` tensor as CFM.inference receives it), prompt [dev] fp32 [B][mel_dim][Tp]
 * (the reference mel, frames >= Tp are generated), noise [dev] fp32 [B][mel_dim][T] = the randn draw of models.py:1030,
 * or NULL to draw it on the device from `seed`; out [dev] fp32 [B][mel_dim][T] (prompt frames are zero, as in the
 * reference).  n_steps Euler steps with d = 1/n_steps. */
int gsv_cfm_inference(gsv_cfm_t* h, const float* mu, const float* prompt, int B, int T, int Tp, int n_steps,
                      const float* noise, float temperature, uint64_t seed, float* out, gsv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * SOLA stitching of the chunked v3/v4 vocoder output (H17, TTS.sola_algorithm, TTS_infer_pack/TTS.py:1611-1637).
 * frags [dev] fp32: the n fragments back to back (lens [host] samples each, every one >= 2 * overlap); modified in
 * place (cross-faded heads).  out [dev] fp32, capacity sum(lens); *out_len [host] = stitched length.  Synchronises
 * the stream (the length depends on the argmax offsets found on the device).
 * ------------------------------------------------------------------------------------- */
int gsv_sola(float* frags, const int* lens, int n, int overlap, float* out, int* out_len, gsv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * H13 audio post-processing (replaces TTS.audio_postprocess, TTS_infer_pack/TTS.py:1377-1429): per fragment divide by
 * its peak when the peak exceeds 1, append `gap` zero samples, concatenate in the order given, scale by 32768 in the
 * fragments' dtype and truncate to int16 (numpy's astype wrap).  frags [host] n device pointers (output order),
 * lens [host] n sample counts, out [dev] int16 with capacity sum(lens) + n * gap.  Asynchronous on `stream`.
 * ------------------------------------------------------------------------------------- */
int gsv_postprocess(const void* const* frags, const int* lens, int n, int dtype, int gap, int16_t* out, gsv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * BigVGAN anti-aliased snake activation (v3 vocoder), the reference's one native kernel.
 * x,y [dev] [B][C][T] of `dtype`; up12/dn12 [dev] 12 filter taps; log_alpha/log_beta [dev] [C].
 * T == 0 returns GSV_OK without a launch (anti_alias_activation_cuda.cu:193-196).
 * ------------------------------------------------------------------------------------- */
int gsv_aa_act_forward(const void* x, void* y, const void* up12, const void* dn12, const void* log_alpha,
                       const void* log_beta, int B, int C, int T, int dtype, gsv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * single-kernel test entry points
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const void* x; const void* w; const float* bias; void* y; const void* res;
  int T_in, T_out, Cin, Cout, taps, stride, dil, pad;
  int pre_act; float pre_slope; int post_act; float scale; int accumulate; int out_f32;
  int ups_u, ups_pad;
  /* optional (0 = defaults): batched GEMM over Z slices with element strides, explicit leading dims */
  int Z; long long xz, wz, yz; int ldx, ldw, ldy;
  const float* gate;   /* optional per-output-channel gate: y = ((W x + b) * gate + res) * scale */
  int bz;              /* grouped convs (Z > 1): element stride of bias between slices (0 = one bias shared by all) */
  long long rz; int ldr; /* residual: element stride between slices (0 = yz) and leading dim (0 = ldy) */
} gsv_conv_desc;
/* fused softmax attention of the DiT blocks alone (fp16, head dim 64): qkv [dev] f16 [T][3*heads*64] (q | k | v column
 * blocks), vt_scratch [dev] heads*64*ceil32(T) halfs, out [dev] f16 [T][heads*64] */
int gsv_op_flash_attn64(const void* qkv, int T, int heads, float scale, void* vt_scratch, void* out, gsv_stream_t stream);
/* enc_p self-attention with window-4 relative positions alone (fp16, head dim 96, module/attentions.py:227-258):
 * qkv [dev] f16 [T][3*heads*96], rel_k / rel_v [dev] fp32 [9][96], vt_scratch heads*96*ceil32(T) halfs, out f16 [T][heads*96] */
int gsv_op_flash_rel96(const void* qkv, int T, int heads, float scale, const float* rel_k, const float* rel_v, void* vt_scratch,
                       void* out, gsv_stream_t stream);
/* the AR decode-step attention alone (reference t2s_model.py:176-221, one query per row over its cached keys; head dim 32):
 * q [dev] [B][H*32]; kc / vc [dev] [B][H][smax][32]; kv_len [dev] int32 [B]: row b attends to keys 0..kv_len[b] (the
 * current token's K/V already stored at position kv_len[b]); active [dev] int32 [B] (0 = row skipped, out untouched);
 * out [dev] [B][H*32].  dtype GSV_F16 / GSV_F32 applies to q, kc, vc, out. */
int gsv_op_decode_attn(const void* q, const void* kc, const void* vc, const int32_t* kv_len, const int32_t* active, int B, int H,
                       int smax, int dtype, void* out, gsv_stream_t stream);
/* channels-last conv1d: x [T_in][Cin], w [Cout][taps*Cin] (tap-major, cin fastest), y [T_out][Cout] */
int gsv_op_conv1d(const gsv_conv_desc* d, int dtype, gsv_stream_t stream);
/* y = LN(x (+res)) over the last dim C; all buffers of `dtype`, gamma/beta fp32 */
int gsv_op_layernorm(const void* x, const void* res, const float* gamma, const float* beta, void* y, int rows,
                     int C, float eps, int dtype, gsv_stream_t stream);
/* reference-audio front-end helpers (host orchestration in gsv/module/mel_processing.py and
 * gsv/feature_extractor/cnhubert.py, the GEMMs are gsv_op_conv1d):
 *  gsv_op_frame: out[t][k] = x[reflect(t*hop + k - pad)], k < frame_len, zero up to ld; x [dev] fp32 [n]; out [T_out][ld] of dtype
 *    (torch.stft's reflect framing, reference module/mel_processing.py:55-71; pad = 0: the operand of a strided Conv1d(1, C, k))
 *  gsv_op_magnitude: re_im [dev] fp32 [T][2*bins] (re | im) -> sqrt(re^2 + im^2 + eps) (:73) as spec [dev] fp32 [bins][T]
 *    (frame_ld = 0), or frame-major [T][frame_ld] zero-filled beyond bins (the operand of the mel-filterbank GEMM, :138-140);
 *    eps < 0: the power spectrum re^2 + im^2 (Kaldi fbank use_power, eres2net/kaldi.py:612-614)
 *  gsv_op_channel_norm: channels-last [T][C]: per-channel mean / biased variance over T, affine, activation (ACT codes of
 *    gsv_conv_desc.post_act) -- torch.nn.GroupNorm(C, C) of the HuBERT feature extractor; scratch [dev] 128*C floats
 *  gsv_op_aff_mix: out = x (1 + t) + y (1 - t) over n fp32 elements (eres2net/fusion.py:22-27, t = tanh of the attention branch)
 *  gsv_op_time_mean: x [dev] fp32 [T][ld] -> out[ld] = mean over T (ERes2NetV2.forward3, eres2net/ERes2NetV2.py:258) */
int gsv_op_frame(const float* x, int n, int frame_len, int hop, int pad, int ld, int T_out, void* out, int dtype, gsv_stream_t stream);
int gsv_op_magnitude(const float* re_im, int T, int bins, float eps, int frame_ld, float* spec, gsv_stream_t stream);
/* one HiFi-GAN ResBlock pair of the generator's narrow stages in one kernel (fp16, C = 16 or 32, reference module/models.py:262-283):
 * y = (convs2(lrelu(convs1(lrelu(x)))) + x) * scale [+ y]; x, y [dev] f16 [T][C] (y must not alias x); w1 / w2 [dev] f16 [C][taps*C]
 * tap-major, convs1 dilated by `dil`, convs2 dilation 1; b1 / b2 [dev] fp32 [C].  Same rounding points as two gsv_op_conv1d launches. */
int gsv_op_conv_pair(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int T, int C, int taps,
                     int dil, float scale, int accumulate, gsv_stream_t stream);
int gsv_op_aff_mix(const float* x, const float* y, const float* t, long long n, float* out, gsv_stream_t stream);
int gsv_op_time_mean(const float* x, int T, int ld, float* out, gsv_stream_t stream);
int gsv_op_channel_norm(const void* x, int T, int C, const float* gamma, const float* beta, float eps, int act, float* scratch,
                        void* y, int dtype, gsv_stream_t stream);
/* sampling kernel alone: logits [dev] fp32 [B][vocab], prev [dev] int32 [B][prev_len], noise [dev]
 * fp32 [B][vocab] or NULL; outputs [dev] int32 [B]: sampled token, argmax of penalised logits */
int gsv_op_sample(const float* logits, int B, int vocab, int vocab_eff, const int32_t* prev, int prev_len,
                  const gsv_sampling_params* sp, const float* noise, int step, int32_t* sampled,
                  int32_t* argmax_tok, gsv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSV_H_ */

"""TEST INFRASTRUCTURE ONLY -- fixtures of the pipeline glue, produced by the REFERENCE's own `TTS` methods (build container
only; VERDICT r2 missing 1 / next-round 3):

  tests/golden/tts_glue_v3.npz, tts_glue_v4.npz   TTS.using_vocoder_synthesis (TTS.py:1431-1494) and
                                                  TTS.using_vocoder_synthesis_batched_infer (:1496-1609) incl. sola_algorithm
  tests/golden/tts_glue_host.npz                  TTS.sola_algorithm (:1611-1635), TTS.audio_postprocess (:1377-1429),
                                                  TTS.to_batch (:842-955), TTS.recovery_order (:957-973)

The module `TTS_infer_pack.TTS` is imported with stubs for the absent packages its import block names
(oracle/ref_import.py::tts_module); the methods are called UNBOUND on a `SimpleNamespace` that carries the reference's own
stage classes (SynthesizerTrnV3, CFM over DiT, BigVGAN / Generator) loaded with gsv.synthetic weights.  Two things are not
the reference's: the rotary embedding inside DiT (x_transformers absent: oracle/rope.py, "parity unpinned" as before) and the
prompt mel (`mel_fn` needs librosa.filters.mel, absent: the module-level `mel_fn` / `mel_fn_v4` are replaced by a function
that returns a fixed synthetic mel -- the mel itself is pinned by tests/golden/mel_v3.npz / mel_v4.npz).  The CFM's
`torch.randn` draw is replaced by the seeded noise both sides regenerate (oracle/glue_cases.py::noise_fn).

    python oracle/gen_golden_tts_glue.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import glue_cases as G  # noqa: E402
from oracle import ref_import  # noqa: E402


def reference_stages(version):
    vcfg, vsd, dit, ocfg, osd, kind = G.models(version)
    from module.models import CFM, SynthesizerTrnV3, Generator
    from GPT_SoVITS.f5_tts.model.backbones.dit import DiT
    d = vcfg["data"]
    mk = {k: v for k, v in vcfg["model"].items() if k != "version"}
    model = SynthesizerTrnV3(d["filter_length"] // 2 + 1, vcfg["train"]["segment_size"] // d["hop_length"],
                             n_speakers=d["n_speakers"], version=version, **mk)
    # the class hard-codes the full-size DiT (models.py:1219-1222); the fixture uses the reduced one the tests use
    model.cfm = CFM(100, DiT(dim=dit["dim"], depth=dit["depth"], heads=dit["heads"], dim_head=dit["dim_head"], ff_mult=dit["ff_mult"],
                             mel_dim=dit["mel_dim"], text_dim=dit["text_dim"], conv_layers=dit["conv_layers"]))
    res = model.load_state_dict(vsd, strict=False)
    bad = [k for k in res.missing_keys if not (k.startswith("enc_q") or "_codebook" in k)]
    assert not bad and not res.unexpected_keys, (bad[:5], res.unexpected_keys[:5])
    model.eval()
    if kind == "hifigan":
        voc = Generator(initial_channel=ocfg["initial_channel"], resblock=ocfg["resblock"],
                        resblock_kernel_sizes=ocfg["resblock_kernel_sizes"], resblock_dilation_sizes=ocfg["resblock_dilation_sizes"],
                        upsample_rates=ocfg["upsample_rates"], upsample_initial_channel=ocfg["upsample_initial_channel"],
                        upsample_kernel_sizes=ocfg["upsample_kernel_sizes"], gin_channels=0, is_bias=True)
        voc.remove_weight_norm()
    else:
        from BigVGAN import bigvgan as rb
        from BigVGAN.env import AttrDict
        voc = rb.BigVGAN(AttrDict({k: v for k, v in ocfg.items() if k != "kind"}))
        voc.remove_weight_norm()
    r2 = voc.load_state_dict(osd, strict=False)
    assert not [k for k in r2.missing_keys if "filter" not in k] and not r2.unexpected_keys
    voc.eval()
    return model, voc, ocfg


class _Randn:
    """stands in for torch.randn inside CFM.inference (models.py:1030): call i returns glue_cases.noise_fn(i, shape)"""

    def __init__(self):
        self.call = 0
        self.orig = torch.randn

    def __call__(self, *size, **kw):
        shape = size[0] if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else size
        out = G.noise_fn(self.call, tuple(int(v) for v in shape))
        self.call += 1
        return out.to(kw.get("dtype") or torch.float32)


def namespace(RT, version):
    model, voc, ocfg = reference_stages(version)
    refer, psem, pph, ref_mel = G.prompt()
    tgt_sr = 24000 if version == "v3" else 32000
    ns = SimpleNamespace()
    ns.configs = SimpleNamespace(device="cpu", version=version, sampling_rate=tgt_sr)
    ns.precision = torch.float32
    ns.prompt_cache = {"prompt_semantic": psem, "phones": pph, "refer_spec": [(refer, None)],
                       "raw_audio": torch.zeros(1, 100), "raw_sr": tgt_sr}
    ns.vits_model = model
    ns.vocoder = voc
    ns.vocoder_configs = dict(G.VC, sr=tgt_sr, upsample_rate=G.upsample_rate(ocfg))
    ns.sola_algorithm = lambda frags, ov: RT.TTS.sola_algorithm(ns, frags, ov)
    ns.recovery_order = lambda data, bil: RT.TTS.recovery_order(ns, data, bil)
    RT.mel_fn = lambda x: ref_mel.clone()
    RT.mel_fn_v4 = lambda x: ref_mel.clone()
    return ns


def run_with_noise(fn):
    r = _Randn()
    torch.randn = r
    try:
        with torch.no_grad():
            return fn()
    finally:
        torch.randn = r.orig


def main():
    RT = ref_import.tts_module()
    torch.set_num_threads(8)
    for version in ("v3", "v4"):
        ns = namespace(RT, version)
        sem, ph = G.single_inputs()
        wav = run_with_noise(lambda: RT.TTS.using_vocoder_synthesis(ns, sem, ph, 1.0, 3))
        out = {"single": wav.numpy().astype(np.float32)}
        for case in ("ragged", "exact"):
            idx, sems, phs = G.batched_inputs(case)
            frags = run_with_noise(lambda: RT.TTS.using_vocoder_synthesis_batched_infer(ns, idx, sems, phs, 1.0, 2))
            out[f"batched_{case}_lens"] = np.array([int(f.numel()) for f in frags], dtype=np.int64)
            out[f"batched_{case}"] = (torch.cat(frags).numpy() if sum(int(f.numel()) for f in frags) else np.zeros(0)).astype(np.float32)
            print(f"[gen_golden] tts_glue_{version} batched {case}: fragment lengths {out[f'batched_{case}_lens'].tolist()}")
        print(f"[gen_golden] tts_glue_{version}: single fragment {wav.shape[0]} samples, rms {wav.pow(2).mean().sqrt():.4f}")
        np.savez_compressed(os.path.join(GOLD, f"tts_glue_{version}.npz"), **out)
    # ---- host-side glue (no models) ----
    ns = SimpleNamespace(configs=SimpleNamespace(device="cpu", sampling_rate=32000), precision=torch.float32)
    ns.recovery_order = lambda data, bil: RT.TTS.recovery_order(ns, data, bil)
    host = {}
    for ci, (n, length, ov) in enumerate(G.SOLA_CASES):
        frags = G.sola_fragments(n, length, ov)
        host[f"sola{ci}"] = RT.TTS.sola_algorithm(ns, [f.clone() for f in frags], ov).numpy().astype(np.float32)
    for name, dtype in (("f32", torch.float32), ("f16", torch.float16)):
        for sb in (True, False):
            audio, bil = G.postprocess_inputs(dtype)
            ns.precision = dtype
            sr, a16 = RT.TTS.audio_postprocess(ns, audio, 32000, bil, 1.0, sb, 0.3, False)
            host[f"post_{name}_{'bucket' if sb else 'flat'}"] = a16
            assert sr == 32000 and a16.dtype == np.int16
    ns.precision = torch.float32
    for ci, (lens, bs, thr, sb) in enumerate(G.TO_BATCH_CASES):
        data, prompt_data = G.to_batch_data(lens)
        batches, bil = RT.TTS.to_batch(ns, data, prompt_data, bs, thr, sb, torch.device("cpu"), torch.float32)
        flat = [i for b in bil for i in b]
        host[f"tb{ci}_index"] = np.array(flat, dtype=np.int64)
        host[f"tb{ci}_sizes"] = np.array([len(b) for b in bil], dtype=np.int64)
        host[f"tb{ci}_max_len"] = np.array([b["max_len"] for b in batches], dtype=np.int64)
        host[f"tb{ci}_all_len"] = np.concatenate([b["all_phones_len"].numpy() for b in batches])
        host[f"tb{ci}_all_phones"] = np.concatenate([p.numpy() for b in batches for p in b["all_phones"]])
        rec = RT.TTS.recovery_order(ns, [[f"{i}" for i in b] for b in bil], bil)
        assert rec == [str(i) for i in range(len(lens))]
    np.savez_compressed(os.path.join(GOLD, "tts_glue_host.npz"), **host)
    print(f"[gen_golden] tts_glue_host: {len(host)} arrays")


if __name__ == "__main__":
    main()

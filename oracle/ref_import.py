"""TEST INFRASTRUCTURE ONLY (oracle pinning) -- never imported by the product path.

Imports the reference's hot-path classes from /root/reference in THIS container so
that the CPU restatement in oracle/ can be pinned against them and golden fixtures
generated.  The reference cannot travel to the GPU box, so nothing
under tests/ -m gpu, smoke() or bench.py may import this module.

Third-party modules the reference imports at module scope but never uses on the
v2 inference path are replaced by empty stubs (torchmetrics: t2s_model.py:9;
x_transformers / torchaudio / librosa: f5_tts/model/*.py).
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "GPT_SoVITS"))


_done = False


def setup():
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    if "torchmetrics" not in sys.modules:
        _stub("torchmetrics")
        _stub("torchmetrics.classification", MulticlassAccuracy=_Dummy)
    if "x_transformers" not in sys.modules:
        # x_transformers is an un-vendored, un-pinned dependency of the reference (requirements.txt:38):
        # its RotaryEmbedding / apply_rotary_pos_emb are RESTATED in oracle/rope.py from the library's
        # published definition, so everything the reference's DiT computes with them is "parity unpinned"
        # with respect to the real library.
        from oracle import rope
        _stub("x_transformers")
        _stub("x_transformers.x_transformers", RotaryEmbedding=rope.RotaryEmbedding,
              apply_rotary_pos_emb=rope.apply_rotary_pos_emb)
    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        ta.transforms = _stub("torchaudio.transforms", MelSpectrogram=_Dummy)
    if "librosa" not in sys.modules:
        lb = _stub("librosa")
        lb.filters = _stub("librosa.filters", mel=None)
        lb.util = _stub("librosa.util", normalize=None, pad_center=None, tiny=None)
    for p in (os.path.join(REF_ROOT, "GPT_SoVITS"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    _done = True


def t2s_decoder_cls():
    setup()
    from AR.models.t2s_model import Text2SemanticDecoder  # noqa
    return Text2SemanticDecoder


def synthesizer_cls():
    setup()
    from module.models import SynthesizerTrn  # noqa
    return SynthesizerTrn


def tts_module():
    """The reference's `TTS_infer_pack.TTS` module (pipeline glue: to_batch, recovery_order, audio_postprocess,
    using_vocoder_synthesis*, sola_algorithm; TTS.py:842-973, 1377-1635).  Its module-scope imports (TTS.py:10-37) name
    packages that are absent here and that none of the pinned methods touch: ffmpeg, peft, pytorch_lightning (through
    AR.models.t2s_lightning_module), tools.audio_sr, tools.i18n, sv, TextPreprocessor's G2P stack; they are replaced by empty
    stubs, like torchmetrics / x_transformers above.  The methods are then called UNBOUND on a namespace that carries the
    reference's own stage classes (oracle/gen_golden_tts_glue.py)."""
    import transformers  # noqa: F401  (imported before the torchaudio stub exists: its availability probe needs a real spec or nothing)
    setup()
    import torch

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    class _I18n:
        def __init__(self, language=None):
            pass

        def __call__(self, key):
            return key

    sys.modules["torchaudio"].transforms.Resample = _Dummy
    for name, attrs in (("ffmpeg", {}), ("pytorch_lightning", {"LightningModule": torch.nn.Module}),
                        ("peft", {"LoraConfig": _Dummy, "get_peft_model": _Dummy}), ("sv", {"SV": _Dummy}),
                        ("tools.audio_sr", {"AP_BWE": _Dummy}),
                        ("tools.i18n.i18n", {"I18nAuto": _I18n, "scan_language_list": lambda: []}),
                        ("TTS_infer_pack.TextPreprocessor", {"TextPreprocessor": _Dummy})):
        if name not in sys.modules:
            parts = name.split(".")
            for i in range(1, len(parts)):
                pk = ".".join(parts[:i])
                if pk not in sys.modules and pk not in ("TTS_infer_pack",):
                    m = _stub(pk)
                    m.__path__ = []
            _stub(name, **attrs)
    import TTS_infer_pack.TTS as ref_tts  # noqa
    return ref_tts

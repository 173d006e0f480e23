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

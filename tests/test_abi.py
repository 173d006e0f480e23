"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gsv.h declares.
No compute is called here (there is no GPU in the build container)."""
import os
import re

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gsv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gsv import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    l = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/gsv.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert l.gsv_abi_version() == 1


def test_product_path_fails_loudly_without_gpu():
    """no CPU fallback: constructing the engine on a CPU device must raise"""
    import pytest
    import torch
    from gsv.AR.models.t2s_model import Text2SemanticDecoder
    from gsv import synthetic as S
    with pytest.raises(RuntimeError):
        Text2SemanticDecoder(S.small_t2s_config(), device="cpu")
    if not torch.cuda.is_available():
        from gsv import _lib
        assert _lib.lib().gsv_init(0) != 0
        assert b"HIP" in _lib.lib().gsv_last_error() or b"device" in _lib.lib().gsv_last_error()

"""gsv: MI355X-native GPT-SoVITS synthesis engine (host side).

Python here is orchestration only; every hot-path operation runs in the HIP
library `libgsv_hip.so` (built from ../csrc) through its C ABI (include/gsv.h).
"""
__all__ = ["synthetic"]

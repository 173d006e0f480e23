"""Times one Linear-shaped launch of gsv_op_conv1d (taps = 1) with HIP events over back-to-back launches; used to probe
the GEMM kernels (GSV_SK_MODE / GSV_NO_GEMM_SK select variants).  Not part of the product."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpt-sovits_amd"))
import torch  # noqa: E402

from gsv import _lib  # noqa: E402


def main():
    _lib.init(0)
    shapes = [(934, 1024, 3072), (934, 1024, 1024), (934, 2048, 1024), (934, 1024, 2048), (5760, 512, 1536)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for T, K, N in shapes:
        x = torch.randn(T, K, device="cuda", dtype=torch.float16)
        w = torch.randn(N, K, device="cuda", dtype=torch.float16) / K ** 0.5
        y = torch.zeros(T, N, device="cuda", dtype=torch.float16)
        d = _lib.ConvDesc(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, T, T, K, N, 1, 1, 1, 0, 0, 0.0, 0, 1.0, 0, 0, 0, 0,
                          0, 0, 0, 0, 0, 0, 0, None)
        st = torch.cuda.current_stream()
        for _ in range(3):
            _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), 1, C.c_void_p(st.cuda_stream)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            _lib.lib().gsv_op_conv1d(C.byref(d), 1, C.c_void_p(st.cuda_stream))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(json.dumps({"T": T, "K": K, "N": N, "us": round(us, 2), "tflops": round(2 * T * K * N / us / 1e6, 1),
                          "mode": os.environ.get("GSV_SK_MODE", "0"), "sk": "GSV_NO_GEMM_SK" not in os.environ}))


if __name__ == "__main__":
    main()

"""Profiling probe: runs the generator's representative conv shapes through gsv_op_conv1d so that
rocprofv3 (--kernel-trace / --pmc) can attribute time and counters per shape.  Not part of the product."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpt-sovits_amd"))
import torch  # noqa: E402
from gsv import _lib  # noqa: E402

DEV = "cuda:0"
_lib.init(0)
SHAPES = [(16, 16, 3, 1, 4096000), (16, 16, 11, 5, 4096000), (32, 32, 7, 3, 2048000), (64, 64, 11, 5, 1024000),
          (128, 128, 7, 3, 512000), (256, 256, 11, 5, 64000), (256, 256, 3, 1, 64000)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if len(sys.argv) > 2:            # explicit shapes: CinxCoutxkxdilxT ...
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
for shape in SHAPES:
    (Cin, Cout, k, dil, T), post = shape[:5], (shape[5] if len(shape) > 5 else 0)       # optional 6th field: post_act code (2 = tanh)
    f32out = 1 if Cout == 1 else 0
    x = torch.randn(T, Cin, device=DEV, dtype=torch.float16)
    w = (torch.randn(Cout, k * Cin, device=DEV) / (Cin * k) ** 0.5).half()
    b = torch.randn(Cout, device=DEV)
    r = torch.randn(T, Cout, device=DEV, dtype=torch.float16)
    y = torch.empty(T, Cout, device=DEV, dtype=torch.float32 if f32out else torch.float16)
    pad = (k * dil - dil) // 2
    d = _lib.ConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None if f32out else r.data_ptr(), T, T, Cin, Cout, k, 1, dil, pad,
                      3, 0.1, post, 1.0, 0, f32out, 0, 0)
    for _ in range(2):
        _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), 1, None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(_lib.lib().gsv_op_conv1d(C.byref(d), 1, None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flops = 2.0 * Cin * Cout * k * T
    byts = 2.0 * T * (Cin + 2 * Cout)
    print(f"C={Cin:4d} k={k:2d} d={dil} T={T:8d}: {dt*1e6:8.1f} us  {flops/dt/1e12:7.1f} TFLOP/s  {byts/dt/1e9:7.1f} GB/s (algorithmic)", flush=True)
    # pure copy of the same bytes for reference
    t0 = time.perf_counter()
    for _ in range(reps):
        y.copy_(r)
    torch.cuda.synchronize()
    dc = (time.perf_counter() - t0) / reps
    print(f"      torch copy of one tensor ({T*Cout*2/1e6:.0f} MB): {dc*1e6:8.1f} us  {2*T*Cout*2/dc/1e9:7.1f} GB/s", flush=True)

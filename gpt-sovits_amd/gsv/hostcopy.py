"""Device -> host copies of result buffers (audio, gathered shards)."""
import numpy as np
import torch


def to_host(t: torch.Tensor) -> np.ndarray:
    """Copy a device tensor into page-locked memory from torch's caching host allocator and return it as a numpy array
    (the array owns the block; it goes back to the allocator's pool, not to the OS, when the caller drops it).

    Why not `.cpu()`: that hands the runtime a fresh pageable block per call.  For copies above ~1 MB the runtime
    registers those pages with the driver instead of staging them, and when glibc later unmaps the block the driver's MMU
    notifier quiesces this process's GPU queues: 20-35 ms of idle GPU with work queued, about one pipeline pass in eight
    (DESIGN.md section 8).  Page-locked blocks from the pool are never unmapped, so the queues are never quiesced."""
    if t.device.type == "cpu":          # host-logic unit tests feed host tensors: nothing to copy
        return t.numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()

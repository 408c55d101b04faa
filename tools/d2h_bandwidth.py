"""Device -> pinned host copy time by size (hipEvents), and the same bytes written by a
kernel straight into the pinned buffer's device mapping."""
import time, torch
dev = torch.device("cuda")
for n in (307200, 5 * 307200):
    src = torch.rand(n, device=dev)
    host = torch.empty(n, pin_memory=True)
    for _ in range(3): host.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): host.copy_(src, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    t0 = time.perf_counter(); host.copy_(src, non_blocking=True); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print("%8d floats (%.1f MB): %.3f ms per copy on the stream (%.1f GB/s); one copy + sync from the host %.3f ms"
          % (n, n * 4 / 1e6, ms, n * 4 / ms / 1e6, wall))

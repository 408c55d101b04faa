"""Does a non_blocking device->pinned-host copy return before earlier stream work is done?"""
import time, torch
dev = torch.device("cuda")
src = torch.zeros(480 * 640, device=dev)
big = torch.zeros(64 * 1024 * 1024, device=dev)
host = torch.empty(480 * 640, pin_memory=True)
def t(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return dt * 1e6
print("copy_ non_blocking, idle stream:        %.1f us" % t(lambda: host.copy_(src, non_blocking=True)))
def busy():
    for _ in range(20): big.add_(1.0)          # ~20 x 60 us of queued kernels
    t0 = time.perf_counter(); host.copy_(src, non_blocking=True); return time.perf_counter() - t0
torch.cuda.synchronize()
d = [busy() for _ in range(10)]; torch.cuda.synchronize()
print("copy_ non_blocking behind ~1 ms of work: %.1f us (host time of the copy call)" % (1e6 * sum(d) / len(d)))
side = torch.cuda.Stream()
def busy_side():
    for _ in range(20): big.add_(1.0)
    ev = torch.cuda.Event(); ev.record()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        side.wait_event(ev); host.copy_(src, non_blocking=True)
    return time.perf_counter() - t0
torch.cuda.synchronize()
d = [busy_side() for _ in range(10)]; torch.cuda.synchronize()
print("same on a side stream after wait_event:  %.1f us" % (1e6 * sum(d) / len(d)))

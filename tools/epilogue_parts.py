import time, torch
dev = torch.device("cuda")
V, HW, world = 5, 480*640, 8
npad = 45312
n_all = V * npad
flat = torch.randn(world * n_all + 1, device=dev)
src = torch.randint(0, world * n_all, (V * HW,), device=dev)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("index_select 6MB int64 idx: %.3f ms" % t(lambda: flat.index_select(0, src)))
src32 = src.int()
print("index_select int32 idx: %.3f ms" % t(lambda: flat.index_select(0, src32)))
maps = flat.index_select(0, src)
def d2h():
    host = torch.empty((V * HW,), dtype=torch.float32, pin_memory=True)
    host.copy_(maps, non_blocking=True)
    e = torch.cuda.Event(); e.record(); e.synchronize()
    return host
print("pinned alloc + D2H 6 MB + sync: %.3f ms" % t(d2h))
host = torch.empty((V * HW,), dtype=torch.float32, pin_memory=True)
def d2h2():
    host.copy_(maps, non_blocking=True)
    torch.cuda.current_stream().synchronize()
print("D2H 6 MB into a held pinned buffer + sync: %.3f ms" % t(d2h2))
h1 = torch.empty((HW,), dtype=torch.float32, pin_memory=True)
def d2h3():
    h1.copy_(maps[:HW], non_blocking=True)
    torch.cuda.current_stream().synchronize()
print("D2H 1.2 MB + sync: %.3f ms" % t(d2h3))
def alloc():
    return torch.empty((V * HW,), dtype=torch.float32, pin_memory=True)
print("pinned alloc only: %.3f ms" % t(alloc))
def ag():
    f = torch.empty((world * n_all + 1,), dtype=torch.float32, device=dev); f[-1] = 0.0
    return f
print("flat alloc + tail zero: %.3f ms" % t(ag))
x = torch.zeros(8, device=dev)
def sync_only():
    x.add_(1); torch.cuda.current_stream().synchronize()
print("tiny kernel + sync: %.3f ms" % t(sync_only))

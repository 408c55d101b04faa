"""What ONE rank of an N-rank run has to do, timed on one GPU: the sharded code path of
RayNetForwardPass with the collectives stubbed out (all_reduce: nothing, all_gather: local
copy).  The depth maps are wrong by construction (partial accumulators); the launches, their
sizes and the host work are those of rank `RANK` of `N`.  Gives the ceiling of the strong
scaling before any xGMI traffic:  t_1 / (N * t_N)."""
import os, sys, time, types
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raynet_amd import _lib
if os.environ.get("RN_FLAGS"):      # A/B build of the library for this run only
    import subprocess
    so = os.path.join(REPO, "tools", "libraynet_hip_ab.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + os.environ["RN_FLAGS"].split() +
                          ["-I", os.path.join(REPO, "include"),
                           os.path.join(_lib.CSRC, "raynet_hip.hip"), "-o", so])
    _lib.LIB_PATH = so
import raynet_amd.forward_pass as F
from raynet_amd.common.generation_parameters import GenerationParameters
from raynet_amd.synthetic import make_synthetic_scene
# CONFIG=config4: BASELINE.json configs[3] (9 views, 128 planes, 256^3, M = 768)
if os.environ.get("CONFIG", "config2") == "config4":
    H, W, V, D_, M_, G_ = 480, 640, 9, 128, 768, 256
else:
    H, W, V, D_, M_, G_ = 480, 640, 5, 64, 384, 128
scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, F=32, padding=11, focal=1.5 * H, seed=1234)
gp = GenerationParameters(depth_planes=D_, neighbors=min(4, V - 1) if V <= 5 else V - 1,
                          grid_shape=np.array([G_] * 3, np.int32),
                          max_number_of_marched_voxels=M_, padding=11, gamma_mrf=0.05)


class FakeDist(object):
    ReduceOp = types.SimpleNamespace(SUM=0, MIN=1, MAX=2)

    def __init__(self, world):
        self.world = world

    capturable = True       # the stand-ins are stream work: the step is captured like RCCL's would be

    def get_backend(self):
        return "fake"

    def all_reduce(self, t, op=None):
        pass

    def all_to_all_single(self, out, inp, out_split=None, in_split=None):
        # this rank receives out_split[q] rows from every rank q: its own block, from itself
        n = out_split[0]
        if n:
            out.view(self.world, n).copy_(inp[:n].view(1, n).expand(self.world, n))

    def reduce_scatter_tensor(self, out, inp, op=None):
        out.copy_(inp.view(self.world, -1)[0])

    def all_gather_into_tensor(self, out, inp, async_op=False):
        out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))
        if async_op:        # like RCCL's work handle: wait() makes the CURRENT stream wait
            ev = torch.cuda.Event()
            ev.record()
            return types.SimpleNamespace(wait=lambda: torch.cuda.current_stream().wait_event(ev))


res = {}
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    for rank in (range(world) if os.environ.get("ALL_RANKS") else sorted({0, world // 2})):
        fake = FakeDist(world)
        F._dist = (lambda f=fake, r=rank, w=world: (f, r, w)) if world > 1 else (lambda: (None, 0, 1))
        from raynet_amd.hip_implementations.options import PathOptions
        fp = F.get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0,
                                                  options=PathOptions.from_env())
        def step():
            for _ in fp.forward_pass(scene, (0, V, 1)):
                pass
        for _ in range(int(os.environ.get("WARM", "8"))):      # (the step is captured once the scatter has settled)
            step()
        import gc
        gc.collect()
        gc.disable()        # a generation-2 collection is a 40 ms pause once every ~20 passes
        ctx = fp._ctx
        torch.cuda.synchronize()
        if not os.environ.get("NO_PROF"):      # what the per-launch event pairs themselves cost
            ctx.prof_begin(capacity=1024)
        t0 = time.perf_counter()
        n = 20
        per_step = []
        for _ in range(n):
            t1 = time.perf_counter()
            step()
            per_step.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        if max(per_step) > 2 * float(np.median(per_step)):
            print("   (step times ms: median %.3f max %.3f at step %d)" % (
                float(np.median(per_step)), max(per_step), int(np.argmax(per_step))))
        fam = {}
        launches = ctx.prof_end() if not os.environ.get("NO_PROF") else []
        for name, _, k_ms in launches:
            fam[name] = fam.get(name, 0.0) + k_ms / n
        if os.environ.get("TIMELINE") == "%d" % world and rank == 0:
            end_prev = 0.0
            per = len(launches) // n
            for (name, _, k_ms), st in list(zip(launches, ctx.prof_starts))[per:3 * per]:
                print("   %-10s start %8.3f  gap %7.3f  dur %7.3f" % (name, st, st - end_prev, k_ms))
                end_prev = st + k_ms
        res[(world, rank)] = ms
        print("world %d rank %d%s: %.3f ms/step  kernels %.3f  (%s)  scatter level/chunks/overflowed %s" % (
            world, rank, " (captured)" if fp.captured else "", ms, sum(fam.values()), " ".join("%s=%.3f" % kv for kv in sorted(fam.items())),
            ctx.scatter_state()))
        if fp.shard_balance is not None and rank == 0:
            bal = np.array(fp.shard_balance, dtype=np.float64).sum(0)
            rows = [fp._plan["bounds"][0][q + 1] - fp._plan["bounds"][0][q] for q in range(world)]
            print("   voxel visits per rank / mean: %s   rows per image: %s" % (
                np.round(bal / bal.mean(), 3).tolist(), rows))
t1 = res.get((1, 0))
for world in ([2, 4, 8] if t1 else []):
    if (world, 0) not in res:
        continue
    t = max(v for (w, r), v in res.items() if w == world)
    print("N=%d: compute-only ceiling of the strong scaling %.0f %%  (%.1f M rays/s)" % (
        world, 100.0 * t1 / (world * t), V * H * W / t / 1e3))

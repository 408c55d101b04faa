"""Time of the MV-CNN twin (raynet_amd/models.py, PyTorch-ROCm / MIOpen) on the 5 zero-padded
480x640 views of config 2 -- the stage in front of the hot path (forward_pass.py:181-198)."""
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import numpy as np   # noqa: E402
import torch         # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raynet_amd.models import get_nn   # noqa: E402

H, W, p, V = 480, 640, 11, 5
model = get_nn("simple_cnn")().cuda().eval()
x = torch.randn((V, 3, H + 2 * p, W + 2 * p), device="cuda")
with torch.no_grad():
    for _ in range(3):
        y = model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = model(x)
    torch.cuda.synchronize()
print("MV-CNN twin, %d views of %dx%d (+%d padding): %.2f ms, features %s" % (
    V, H, W, p, (time.perf_counter() - t0) / 10 * 1e3, tuple(y.shape)))

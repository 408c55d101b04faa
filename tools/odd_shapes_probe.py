"""Diagnostic: message / accumulator differences HIP vs oracle after 1, 2, 3 BP iterations on
small planted scenes of odd shapes."""
import sys, os, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import oracle
from test_forward_pass_gpu import _gp, _oracle_forward
from raynet_amd.forward_pass import get_forward_pass_factory
from raynet_amd.synthetic import make_synthetic_scene
H, W, V = 24, 32, 5
for robust in (False, True):
  oracle.Oracle.set_robust_messages(robust)
  for D,M,grid,nb in [(48, 16, (32, 32, 32), 2),(100, 96, (30, 33, 17), 3),(64, 96, (32,32,32), 4)]:
    scene, bank = make_synthetic_scene(H=H, W=W, n_views=V, focal=1.5 * H)
    gp = _gp(D, M, grid, neighbors=nb)
    for iters in (1, 2, 3):
        fp = get_forward_pass_factory("raynet")(bank, gp, "sample_in_bbox", (H, W), 0, bp_iterations=iters)
        depths = list(fp.forward_pass(scene, (0, 3, 1)))
        acc, msgs, depths_o, dists = _oracle_forward(oracle, scene, bank, gp, [0, 1, 2], H, W, iters=iters)
        worst = 0; where=None
        for r in range(3):
            m = np.zeros((H * W, M), np.float32)
            m[fp.ray_index[r].cpu().numpy().astype(np.int64)] = fp.messages[r].cpu().numpy()
            d = np.abs(m - msgs[r])
            if d.max() > worst:
                worst = d.max(); i = np.unravel_index(np.argmax(d), d.shape); where = (r,)+tuple(int(x) for x in i)+(float(msgs[r][i]), float(m[i]))
        print('robust' if robust else 'literal', D, M, grid, 'iters', iters, 'acc diff %.3g (max %.3g)' % (np.abs(fp.accumulator.cpu().numpy() - acc).max(), np.abs(acc).max()), 'msg diff %.3g at' % worst, where)
oracle.Oracle.set_robust_messages(False)

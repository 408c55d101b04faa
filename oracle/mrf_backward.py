"""Float64 statement of the differentiable MRF block and of its analytic backward.

TEST INFRASTRUCTURE ONLY (never imported by raynet_amd).

Forward (what the reference's training graph builds with TF ops,
raynet/tf_implementations/forward_backward_pass.py:194-230 + raynet/mrf/mrf_tf.py, and
differentiates by autodiff; here with the NumPy/CUDA conventions of SURVEY.md section 9):

    S [n, D]  (softmax output)
      -> planes->voxels linear interpolation, normalised      (planes_voxels_mapping.cu:6-92)
      -> clip to [1e-5, 1-1e-5] and renormalise               (mrf_bp.cu:103-111)
      -> `iters` sum-product BP sweeps over THESE rays         (mrf_bp.cu:88-177)
      -> per-ray depth distribution                            (mrf_bp.cu:37-86)

The reference has no hand-written backward; the backward below is the
exact reverse-mode derivative of this forward (clip gradients pass where not clipped, like
tf.clip_by_value).  tests/test_mrf_backward.py checks it against central finite differences
of `forward` in float64; the HIP backward is then compared with it.
"""
import numpy as np

LO_S, HI_S = np.float64(np.float32(1e-5)), np.float64(np.float32(1 - 1e-5))
LO_O, HI_O = np.float64(np.float32(1e-4)), np.float64(np.float32(1 - 1e-4))


def _interp_weights(voxel_centers, start, end, D, left=None):
    """Per-voxel (left plane, c1, c2) of the monotone walk (planes_voxels_mapping.cu:48-84)."""
    ray = end - start
    t = np.clip((voxel_centers - start).dot(ray) / ray.dot(ray), 1e-4, 1 - 1e-4)
    step = 1.0 / (D - 1)
    if left is None:
        left = np.minimum(np.ceil(t / step).astype(np.int64) - 1, D - 2)
        left = np.maximum.accumulate(np.maximum(left, 0))
    left = np.asarray(left, np.int64)
    ld = np.abs(t - left * step)
    rd = np.abs(t - (left + 1) * step)
    return left, 1.0 - ld / (ld + rd), 1.0 - rd / (ld + rd)


def _sigmoid(mu):
    return 0.5 * (1.0 + np.tanh(0.5 * mu))


def _ray_messages(s, a, m):
    """One ray of one BP sweep.  Returns the new messages and what the backward needs."""
    mu = a - m
    sig = _sigmoid(mu)
    o = np.clip(sig, LO_O, HI_O)
    q = 1.0 - o
    T = np.concatenate([[1.0], np.cumprod(q)[:-1]])
    w = o * T * s
    C = np.concatenate([[0.0], np.cumsum(w)[:-1]])
    U = np.cumsum(w[::-1])[::-1] - w
    pos = C + T * s
    neg = C + U / q
    return np.log(pos) - np.log(neg), dict(sig=sig, o=o, q=q, T=T, w=w, C=C, U=U, pos=pos, neg=neg)


def _chain_from_wbar(wbar, Tbar, sbar, qbar, s, c):
    """Shared tail of both backward passes: from dL/dw (+ partial dL/dT, dL/ds, dL/dq) to
    (dL/ds, dL/dmu)."""
    o, q, T, sig = c["o"], c["q"], c["T"], c["sig"]
    obar = wbar * T * s
    Tbar = Tbar + wbar * o * s
    sbar = sbar + wbar * o * T
    TT = Tbar * T
    suffix_TT = np.cumsum(TT[::-1])[::-1] - TT          # sum_{j>k} Tbar_j T_j
    qbar = qbar + suffix_TT / q
    obar = obar - qbar
    inside = (sig > LO_O) & (sig < HI_O)
    mubar = obar * sig * (1.0 - sig) * inside
    return sbar, mubar


def _ray_messages_bwd(g, s, c):
    pos, neg, q, T, U = c["pos"], c["neg"], c["q"], c["T"], c["U"]
    pbar = g / pos
    nbar = -g / neg
    Cbar = pbar + nbar
    Tbar = pbar * s
    sbar = pbar * T
    Ubar = nbar / q
    qbar = -nbar * U / (q * q)
    wbar = (np.cumsum(Cbar[::-1])[::-1] - Cbar) + (np.cumsum(Ubar) - Ubar)
    return _chain_from_wbar(wbar, Tbar, sbar, qbar, s, c)


def forward(S, voxel_centers, rvi, rvc, starts, ends, grid_shape, gamma=0.05, iters=3,
            keep=False, planes=None):
    """S [n, D] float64 -> depth distributions [n, M] float64 (zero beyond count / for
    rays with count <= 1).  voxel_centers: [gx, gy, gz, 3].  planes: optional [n, M] left
    plane indices (e.g. the fp32 ones of the C oracle, so that a float64 / fp32 comparison
    does not trip over a plane boundary)."""
    n, D = S.shape
    M = rvi.shape[1]
    prior = np.log(gamma) - np.log(1 - gamma)
    rays = []
    for r in range(n):
        c = int(rvc[r])
        if c <= 1:
            rays.append(None)
            continue
        idx = tuple(rvi[r, :c].T)
        left, c1, c2 = _interp_weights(voxel_centers[idx].astype(np.float64),
                                       starts[r].astype(np.float64), ends[r].astype(np.float64), D,
                                       None if planes is None else planes[r, :c])
        z = c1 * S[r, left] + c2 * S[r, left + 1]
        x = z / z.sum()
        y = np.clip(x, LO_S, HI_S)
        s = y / y.sum()
        rays.append(dict(idx=idx, left=left, c1=c1, c2=c2, z=z, x=x, y=y, s=s, c=c))
    acc = np.full(tuple(grid_shape), prior, np.float64)
    msgs = [np.zeros((n, M), np.float64)]
    accs = [acc]
    caches = []
    for it in range(iters):
        new = np.full(tuple(grid_shape), prior, np.float64)
        m_out = np.zeros((n, M), np.float64)
        cache_it = []
        for r in range(n):
            ray = rays[r]
            if ray is None:
                cache_it.append(None)
                continue
            mo, cch = _ray_messages(ray["s"], accs[-1][ray["idx"]], msgs[-1][r, :ray["c"]])
            m_out[r, :ray["c"]] = mo
            np.add.at(new, ray["idx"], mo)
            cache_it.append(cch)
        msgs.append(m_out)
        accs.append(new)
        caches.append(cache_it)
    out = np.zeros((n, M), np.float64)
    dcache = []
    for r in range(n):
        ray = rays[r]
        if ray is None:
            dcache.append(None)
            continue
        _, cch = _ray_messages(ray["s"], accs[-1][ray["idx"]], msgs[-1][r, :ray["c"]])
        W = cch["w"].sum()
        out[r, :ray["c"]] = cch["w"] / W
        cch["W"] = W
        dcache.append(cch)
    if keep:
        return out, dict(rays=rays, accs=accs, msgs=msgs, caches=caches, dcache=dcache)
    return out


def backward(G, S, voxel_centers, rvi, rvc, starts, ends, grid_shape, gamma=0.05, iters=3,
             planes=None, with_prior=False):
    """dL/dS [n, D] for dL/d(out) = G [n, M]; with_prior: also dL/d(prior log-odds) (every
    accumulator starts from the prior, so it collects the sum of every accumulator's
    gradient)."""
    out, k = forward(S, voxel_centers, rvi, rvc, starts, ends, grid_shape, gamma, iters, keep=True,
                     planes=planes)
    n, D = S.shape
    M = rvi.shape[1]
    rays, accs, msgs = k["rays"], k["accs"], k["msgs"]
    s_bar = [np.zeros(ray["c"]) if ray is not None else None for ray in rays]
    acc_bar = np.zeros(tuple(grid_shape))
    m_bar = np.zeros((n, M))
    # depth distribution d = w / W
    for r in range(n):
        ray = rays[r]
        if ray is None:
            continue
        c, cch = ray["c"], k["dcache"][r]
        d = cch["w"] / cch["W"]
        g = G[r, :c]
        wbar = (g - (g * d).sum()) / cch["W"]
        sb, mub = _chain_from_wbar(wbar, np.zeros(c), np.zeros(c), np.zeros(c), ray["s"], cch)
        s_bar[r] += sb
        np.add.at(acc_bar, ray["idx"], mub)
        m_bar[r, :c] = -mub
    prior_bar = acc_bar.sum()
    # BP iterations in reverse
    for it in range(iters - 1, -1, -1):
        acc_bar_prev = np.zeros(tuple(grid_shape))
        m_bar_prev = np.zeros((n, M))
        for r in range(n):
            ray = rays[r]
            if ray is None:
                continue
            c = ray["c"]
            g = m_bar[r, :c] + acc_bar[ray["idx"]]       # message used directly + via acc
            sb, mub = _ray_messages_bwd(g, ray["s"], k["caches"][it][r])
            s_bar[r] += sb
            np.add.at(acc_bar_prev, ray["idx"], mub)
            m_bar_prev[r, :c] = -mub
        acc_bar, m_bar = acc_bar_prev, m_bar_prev
        prior_bar += acc_bar.sum()
    # clip + renorm, mapping
    dS = np.zeros((n, D))
    for r in range(n):
        ray = rays[r]
        if ray is None:
            continue
        sb, s, y, x, z = s_bar[r], ray["s"], ray["y"], ray["x"], ray["z"]
        ybar = (sb - (sb * s).sum()) / y.sum()
        xbar = ybar * ((x > LO_S) & (x < HI_S))
        zbar = (xbar - (xbar * x).sum()) / z.sum()
        np.add.at(dS[r], ray["left"], ray["c1"] * zbar)
        np.add.at(dS[r], ray["left"] + 1, ray["c2"] * zbar)
    if with_prior:
        return dS, prior_bar
    return dS

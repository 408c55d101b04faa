"""NumPy restatement of the reference's CPU path for the MRF block -- TEST INFRASTRUCTURE and
bench.py's `cpu_baseline` leg ONLY (see oracle/oracle.py); the product never imports it.

What it restates: raynet/mrf/mrf_np.py -- `clip_and_renorm` (:4-8),
`single_ray_belief_propagation` (:11-126), `single_ray_depth_estimate` (:129-203),
`belief_propagation` (:243-330) and `compute_depth_distribution` (:333-385) -- i.e. the
path `BPInference("numpy")` runs (raynet/mrf/bp_inference.py) and the one BASELINE.md 3(i)
names as the reference's own CPU implementation.  Kept on purpose, because they are what
the reference's numbers are made of:
  * the per-ray Python loop (one NumPy call sequence per ray and sweep);
  * the mixed precision: occupancies in float32, the transmittance as a float32 cumulative
    product widened to float64, every cumulative SUM in float64, messages rounded to
    float32 before the logit;
  * rays with <= 1 voxel send nothing and keep an all-zero row (:300, :376);
  * NumPy-1 casting of the prior: the accumulators are float32.
Pinned: tests/test_cpu_reference.py holds it BIT-EQUAL to the outputs of the reference's own
functions (tests/golden/ref_mrf_np.npz, ref_mrf_np_saturated.npz)."""
import numpy as np

_LO, _HI = 1e-4, 1 - 1e-4


def renormalised(column, eps=1e-5):
    """mrf_np.py:4-8."""
    c = np.clip(column, eps, 1 - eps)
    return c / c.sum()


def _ray_terms(voxels, acc, msg, s):
    """Occupancy o (float32), transmittance T_i = prod_{k<i}(1 - o_k) and the weights
    w_i = o_i T_i s_i (float64) of one ray -- the part :53-82 and :170-201 share."""
    mu = acc[voxels[:, 0], voxels[:, 1], voxels[:, 2]] - msg
    top = np.maximum(0.0, mu)
    e0, e1 = np.exp(0.0 - top), np.exp(mu - top)
    o = np.clip(e1 / (e1 + e0), _LO, _HI)
    T = np.concatenate(([1.0], np.cumprod(1 - o)))[:-1]       # float32 product, float64 storage
    return o, T, o * T * s


def ray_messages(voxels, acc, msg, s):
    """New ray->occupancy log-odds messages of one ray (mrf_np.py:11-126)."""
    o, T, w = _ray_terms(voxels, acc, msg, s)
    before = np.concatenate(([0.0], w))[:-1].cumsum()          # sum_{j<i} w_j, float64
    after = np.concatenate((w, [0.0]))[::-1].cumsum()[::-1][1:]  # sum_{j>i} w_j, float64
    base = before.astype(np.float32)
    pos = (base + T * s).astype(np.float32)
    neg = (base + after / (1 - o)).astype(np.float32)
    p = pos / (pos + neg)
    return np.log(p) - np.log(1 - p)


def ray_depth_distribution(voxels, acc, msg, s):
    """mrf_np.py:129-203."""
    _, _, w = _ray_terms(voxels, acc, msg, s)
    return w / w.sum()


def prior_log_odds(gamma):
    return np.float32(np.log(gamma) - np.log(1 - gamma))


def belief_propagation(S, rvi, rvc, msgs, grid_shape, gamma=0.05, bp_iterations=3,
                       callback=None):
    """mrf_np.py:243-330: messages zeroed, accumulators start at the prior, one pass over
    the rays per iteration, accumulator hand-over + prior refill after each."""
    msgs.fill(0)
    prior = prior_log_odds(gamma)
    acc_prev = np.full(tuple(grid_shape), prior, np.float32)
    acc_new = np.full(tuple(grid_shape), prior, np.float32)
    for it in range(bp_iterations):
        for r in range(len(rvc)):
            c = int(rvc[r])
            if c <= 1:
                continue
            v = rvi[r, :c]
            m = ray_messages(v, acc_prev, msgs[r, :c], renormalised(S[r, :c]))
            acc_new[v[:, 0], v[:, 1], v[:, 2]] += m
            msgs[r, :c] = m
        acc_prev[:] = acc_new
        acc_new.fill(prior)
        if callback is not None:
            callback(it, acc_prev, msgs)
    return acc_prev, msgs


def compute_depth_distribution(S, rvi, rvc, msgs, acc):
    """mrf_np.py:333-385."""
    S_new = np.zeros_like(S)
    for r in range(len(rvc)):
        c = int(rvc[r])
        if c <= 1:
            continue
        S_new[r, :c] = ray_depth_distribution(rvi[r, :c], acc, msgs[r, :c], renormalised(S[r, :c]))
    return S_new

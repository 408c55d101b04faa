"""float64 statement of one BP sweep, shared by the parity tests."""
import numpy as np


def bp_truth_f64(S, rvi, rvc, acc, msgs):
    """One BP sweep in float64 on the same fp32 inputs: the value
    both fp32 implementations approximate.  Returns messages and, per entry, the
    amplification of an ulp-of-W error in the reference's (cumsum1 - cumsum2)."""
    out = np.zeros(msgs.shape, np.float64)
    cancel = np.zeros(msgs.shape, np.float64)
    lo, hi = np.float32(1e-5), np.float32(1 - 1e-5)
    for r in range(len(rvc)):
        c = int(rvc[r])
        if c <= 1:
            continue
        s = np.clip(S[r, :c], lo, hi).astype(np.float64)
        s /= s.sum()
        idx = tuple(rvi[r, :c].T)
        mu = acc[idx].astype(np.float64) - msgs[r, :c]
        o = np.clip(1.0 / (1.0 + np.exp(-mu)), np.float32(1e-4), np.float32(1 - 1e-4))
        T = np.concatenate([[1.0], np.cumprod(1 - o)[:-1]])
        w = o * T * s
        C = np.concatenate([[0.0], np.cumsum(w)[:-1]])
        suf = np.cumsum(w[::-1])[::-1] - w
        pos = C + T * s
        neg = C + suf / (1 - o)
        out[r, :c] = np.log(pos) - np.log(neg)
        cancel[r, :c] = w.sum() / ((1 - o) * neg)
    return out, cancel

"""f1: the MV-CNN twin (raynet_amd/models.py) against an independent NumPy evaluation of the
reference's network (raynet/models.py:90-111: 5 x [Conv2D 3x3 valid 32, BatchNormalization],
ReLU after the first four; Keras layouts and defaults: NHWC, HWIO kernels, correlation,
BN epsilon 1e-3 on the moving statistics), with weights handed over in the reference's own
order (models.py:329-339)."""
import io

import numpy as np
import pytest


def reference_weights(rng, in_channels=3, filters=32, blocks=5):
    w = []
    c = in_channels
    for _ in range(blocks):
        w += [(rng.standard_normal((3, 3, c, filters)) * np.sqrt(2.0 / (9 * c))).astype(np.float32),
              (rng.standard_normal(filters) * 0.1).astype(np.float32),
              (1 + 0.2 * rng.standard_normal(filters)).astype(np.float32),      # gamma
              (0.1 * rng.standard_normal(filters)).astype(np.float32),          # beta
              (0.1 * rng.standard_normal(filters)).astype(np.float32),          # moving_mean
              (0.5 + rng.random(filters)).astype(np.float32)]                   # moving_variance
        c = filters
    return w


def numpy_forward(x, weights, eps=1e-3):
    """x: [n, H, W, C] float64 -> [n, H-10, W-10, 32]; plain loops over the 3x3 taps."""
    y = x.astype(np.float64)
    nb = len(weights) // 6
    for b in range(nb):
        k, bias, gamma, beta, mean, var = (a.astype(np.float64) for a in weights[6 * b:6 * b + 6])
        n, H, W, _ = y.shape
        out = np.zeros((n, H - 2, W - 2, k.shape[3]))
        for di in range(3):
            for dj in range(3):
                out += y[:, di:di + H - 2, dj:dj + W - 2, :] @ k[di, dj]
        out += bias
        out = gamma * (out - mean) / np.sqrt(var + eps) + beta
        y = np.maximum(out, 0.0) if b < nb - 1 else out
    return y


def _check(device):
    import torch
    from raynet_amd.models import SimpleCNN
    rng = np.random.default_rng(7)
    weights = reference_weights(rng)
    buf = io.BytesIO()
    np.savez(buf, *weights)                       # how converted Keras weights are stored
    buf.seek(0)
    net = SimpleCNN().to(device).load_reference_weights(buf)
    x = rng.random((2, 30, 34, 3)).astype(np.float32)
    got = net.predict(x)
    assert tuple(got.shape) == (2, 20, 24, 32) and got.dtype == torch.float32
    want = numpy_forward(x, weights)
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # the order is the reference's: round trip, and a permuted list is refused by its shapes
    back = net.reference_weights()
    assert len(back) == 30 and all(np.array_equal(a, b) for a, b in zip(back, weights))
    with pytest.raises(ValueError):
        SimpleCNN().load_reference_weights(weights[1:] + weights[:1])
    with pytest.raises(ValueError):
        SimpleCNN().load_reference_weights(weights[:29])


def test_simple_cnn_matches_numpy_evaluation_cpu():
    _check("cpu")


@pytest.mark.gpu
def test_simple_cnn_matches_numpy_evaluation_gpu():
    import torch
    assert torch.cuda.is_available()
    _check("cuda")


@pytest.mark.gpu
def test_forward_pass_script_takes_reference_ordered_weights(tmp_path):
    """`--weight_file x.npz` (the reference's weight order) reaches the network the path uses."""
    import torch
    from raynet_amd.scripts.forward_pass import load_model
    rng = np.random.default_rng(3)
    weights = reference_weights(rng)
    path = str(tmp_path / "w.npz")
    np.savez(path, *weights)
    net = load_model(path, device="cuda")
    x = rng.random((1, 26, 28, 3)).astype(np.float32)
    want = numpy_forward(x, weights)
    assert np.abs(net.predict(x).cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_patch_batches_take_the_row_matrix_path_with_the_same_results():
    """Config 5 feeds the twin 32,000 patches of 11 x 11 per view and call
    (tf_implementations/forward_backward_pass.py:177-182).  `SimpleCNN.forward` sends such a
    batch through `forward_patches` -- per layer one GEMM on im2col rows + batch normalisation
    over the rows -- instead of MIOpen's small-image convolutions: the same function (float64:
    to rounding, values and every gradient; float32: to the dot products' summation order),
    the same running statistics, train and eval mode."""
    import copy
    import torch
    from raynet_amd.models import SimpleCNN
    torch.manual_seed(0)
    m = SimpleCNN().double().train()
    m2 = copy.deepcopy(m)
    x = torch.randn(SimpleCNN.PATCH_MIN_BATCH, 3, 11, 11, dtype=torch.float64, requires_grad=True)
    t = torch.randn(len(x), 32, 1, 1, dtype=torch.float64)
    y1 = m.net(x)
    g1 = torch.autograd.grad(((y1 - t) ** 2).sum(), [x] + list(m.parameters()))
    x2 = x.detach().clone().requires_grad_(True)
    y2 = m2(x2)                       # dispatches on the shape
    assert y2.shape == y1.shape == (len(x), 32, 1, 1)
    g2 = torch.autograd.grad(((y2 - t) ** 2).sum(), [x2] + list(m2.parameters()))
    assert float((y1 - y2).abs().max()) < 1e-12
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-10 * max(1.0, float(a.abs().max()))    # (conv biases under a batch norm: zero gradient, rounding noise)
    for b1, b2 in zip(m.net, m2.net):
        if isinstance(b1, torch.nn.BatchNorm2d):
            assert torch.allclose(b1.running_mean, b2.running_mean, atol=1e-14)
            assert torch.allclose(b1.running_var, b2.running_var, atol=1e-14)
            assert int(b1.num_batches_tracked) == int(b2.num_batches_tracked) == 1
    m.eval()
    m2.eval()
    assert float((m.net(x) - m2(x)).abs().max()) < 1e-12
    # a whole image (or a small batch) keeps the convolution path
    calls = []
    m2.forward_patches = lambda z: calls.append(1)
    m2(torch.randn(2, 3, 40, 40, dtype=torch.float64))
    m2(torch.randn(8, 3, 11, 11, dtype=torch.float64))
    assert not calls
    # float32: the dot products' summation order only
    f = SimpleCNN().train()
    xf = torch.randn(SimpleCNN.PATCH_MIN_BATCH, 3, 11, 11)
    f2 = copy.deepcopy(f)
    assert float((f.net(xf) - f2(xf)).abs().max()) < 1e-4

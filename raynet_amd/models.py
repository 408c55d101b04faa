"""MV-CNN twin in PyTorch-ROCm (north_star: the feature extractor stays in the framework).

Architecture of raynet/models.py:90-111 `create_simple_cnn`: 5 x [Conv 3x3 valid, 32
filters, BatchNorm], ReLU after the first four.  Five valid 3x3 convolutions shrink a
(H+2p) x (W+2p) zero-padded image with p = 11 to (H+p+1) x (W+p+1), the feature-map
extent the kernels index (feature_similarities.cu:73-74).  The reference ships no
weights, so this is random-initialised; `predict` mirrors Keras' NHWC in / NHWC out.

Trained reference weights come in through `load_reference_weights`: the reference reads
them from its HDF5 file in the order `[w for l in model.layers for w in l.weights]`
(raynet/models.py:329-339), which for this network is, per block, Conv2D `kernel`
[3, 3, c_in, c_out] and `bias` [c_out], then BatchNormalization `gamma`, `beta`,
`moving_mean`, `moving_variance` [c_out] -- 30 arrays.  HDF5 cannot be read in this image
(no h5py); the same list saved with `numpy.savez(path, *weights)` can.
"""
import numpy as np
import torch
from torch import nn


class SimpleCNN(nn.Module):
    def __init__(self, in_channels=3, filters=32):
        super(SimpleCNN, self).__init__()
        layers = []
        c = in_channels
        for i in range(5):
            layers.append(nn.Conv2d(c, filters, kernel_size=3))
            layers.append(nn.BatchNorm2d(filters, eps=1e-3, momentum=0.01))
            if i < 4:
                layers.append(nn.ReLU())
            c = filters
        self.net = nn.Sequential(*layers)
        self.filters = filters

    # Below this spatial size and above this batch an input is a batch of PATCHES (config 5: the
    # 11 x 11 patches around the rays' projected sample points, 32,000 per view and call,
    # tf_implementations/forward_backward_pass.py:177-182) and takes the row-matrix path.
    PATCH_MAX_SIDE, PATCH_MIN_BATCH = 16, 2048
    # "auto": by the shape rule above; True / False: always / never the row-matrix path (a caller
    # that knows what it feeds -- the batch provider of config 5 -- need not rely on the guess)
    patch_path = "auto"

    def forward(self, x):           # NCHW
        use = self.patch_path
        if use == "auto":
            use = x.dim() == 4 and max(x.shape[-2:]) <= self.PATCH_MAX_SIDE and \
                x.shape[0] >= self.PATCH_MIN_BATCH
        if use and x.dim() == 4 and min(x.shape[-2:]) >= 3:
            return self.forward_patches(x)
        return self.net(x)

    def forward_patches(self, x):
        """The same five [Conv 3x3 valid, BatchNorm(, ReLU)] blocks for a large batch of small
        patches, as row matrices: every layer is ONE GEMM  [B * h' * w', 9 C] x [9 C, 32]  on the
        im2col rows (channels-last, so a 3 x 3 x C window is a contiguous-inner gather), batch
        normalisation over the rows (= BatchNorm2d's statistics over (N, h, w)), ReLU.  MIOpen's
        convolution solvers for 32,000 images of 11 x 11 take 160 ms per view forward + backward
        (profiles/r03_train_bench_config5.txt); five skinny GEMMs and their unfold copies take a
        fraction of that.  Same arithmetic as `self.net` up to the summation order of a dot
        product of 9 C terms (tests/test_models.py).  NCHW in, NCHW out."""
        from torch.nn import functional as F
        y = x.permute(0, 2, 3, 1)                                  # [B, h, w, C] (a view)
        blocks = self._blocks()
        for i, (conv, bn) in enumerate(blocks):
            B, h, w, C = y.shape
            # [B, h-2, w-2, C, 3, 3] windows -> rows [B (h-2) (w-2), 3 * 3 * C] in (ky, kx, c) order
            cols = y.unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3)
            rows = cols.reshape(B * (h - 2) * (w - 2), 9 * C)
            wm = conv.weight.permute(2, 3, 1, 0).reshape(9 * C, conv.out_channels)
            out = torch.addmm(conv.bias, rows, wm)
            # nn.BatchNorm2d.forward's own bookkeeping: momentum None = cumulative average
            factor = 0.0 if bn.momentum is None else bn.momentum
            if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
                if bn.momentum is None:
                    factor = 1.0 / float(bn.num_batches_tracked)
            out = F.batch_norm(out, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                               bn.training or not bn.track_running_stats, factor, bn.eps)
            if i + 1 < len(blocks):         # the last block has no ReLU (models.py:90-111)
                out = F.relu(out)
            y = out.view(B, h - 2, w - 2, conv.out_channels)
        return y.permute(0, 3, 1, 2)

    # ---- weights in the reference's (Keras) order and layouts -----------------------
    def _blocks(self):
        mods = list(self.net)
        return [(mods[i], mods[i + 1]) for i in range(len(mods)) if isinstance(mods[i], nn.Conv2d)]

    @torch.no_grad()
    def load_reference_weights(self, weights):
        """weights: the 30 arrays of `model.get_weights()` / models.py:333-337, as a list, or
        the path of (or an open) `.npz` written with `numpy.savez(path, *weights)`."""
        if isinstance(weights, (str, bytes)) or hasattr(weights, "read"):
            weights = np.load(weights)
        if hasattr(weights, "files"):        # NpzFile: arr_0, arr_1, ... in positional order
            weights = [weights["arr_%d" % i] for i in range(len(weights.files))]
        weights = [np.asarray(w, dtype=np.float32) for w in weights]
        blocks = self._blocks()
        if len(weights) != 6 * len(blocks):
            raise ValueError("expected %d arrays (kernel, bias, gamma, beta, moving_mean, "
                             "moving_variance per block), got %d" % (6 * len(blocks), len(weights)))
        for b, (conv, bn) in enumerate(blocks):
            kernel, bias, gamma, beta, mean, var = weights[6 * b:6 * b + 6]
            want = (3, 3, conv.in_channels, conv.out_channels)
            if tuple(kernel.shape) != want:
                raise ValueError("block %d: kernel %s, expected %s" % (b, kernel.shape, want))
            for name, a in (("bias", bias), ("gamma", gamma), ("beta", beta),
                            ("moving_mean", mean), ("moving_variance", var)):
                if tuple(a.shape) != (conv.out_channels,):
                    raise ValueError("block %d: %s has shape %s" % (b, name, a.shape))
            dev = conv.weight.device
            # Keras HWIO -> torch OIHW; both frameworks correlate (no kernel flip)
            conv.weight.copy_(torch.from_numpy(kernel).permute(3, 2, 0, 1).contiguous().to(dev))
            conv.bias.copy_(torch.from_numpy(bias).to(dev))
            bn.weight.copy_(torch.from_numpy(gamma).to(dev))
            bn.bias.copy_(torch.from_numpy(beta).to(dev))
            bn.running_mean.copy_(torch.from_numpy(mean).to(dev))
            bn.running_var.copy_(torch.from_numpy(var).to(dev))
        return self

    @torch.no_grad()
    def reference_weights(self):
        """The inverse: this network's weights as the reference's list of 30 arrays."""
        out = []
        for conv, bn in self._blocks():
            out += [conv.weight.permute(2, 3, 1, 0).contiguous().cpu().numpy(),
                    conv.bias.cpu().numpy(), bn.weight.cpu().numpy(), bn.bias.cpu().numpy(),
                    bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy()]
        return out

    @torch.no_grad()
    def predict(self, images_nhwc):
        """Keras-style: (N, H, W, C) array -> (N, H-10, W-10, F) float32 CUDA tensor."""
        self.eval()
        dev = next(self.parameters()).device
        x = torch.as_tensor(np.asarray(images_nhwc), dtype=torch.float32, device=dev)
        y = self.net(x.permute(0, 3, 1, 2).contiguous())
        return y.permute(0, 2, 3, 1).contiguous()


def get_nn(name):
    # raynet/models.py:473-479
    if name == "simple_cnn":
        return SimpleCNN
    raise NotImplementedError(name)

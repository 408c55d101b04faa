"""MV-CNN twin in PyTorch-ROCm (north_star: the feature extractor stays in the framework).

Architecture of raynet/models.py:90-111 `create_simple_cnn`: 5 x [Conv 3x3 valid, 32
filters, BatchNorm], ReLU after the first four.  Five valid 3x3 convolutions shrink a
(H+2p) x (W+2p) zero-padded image with p = 11 to (H+p+1) x (W+p+1), the feature-map
extent the kernels index (feature_similarities.cu:73-74).  The reference ships no
weights, so this is random-initialised; `predict` mirrors Keras' NHWC in / NHWC out.
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

    def forward(self, x):           # NCHW
        return self.net(x)

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

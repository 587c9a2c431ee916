"""EDSR_1D and ESPCN_1D, the two comparison networks of the reference that ride on SampleShuffle1D
(models/edsr_1d.py:8-45, models/espcn_1d.py:8-36; selected by main.py:139-142).

Their convolutions are stock ATen calls in the reference and stay stock ATen (MIOpen on ROCm) here -- they are not on
the accelerated StofNet path ("baselines riding on the new shuffle"); the sub-pixel
step is the gfx950 SampleShuffle1D kernel, with the inverse permutation as its backward so that the networks train.
Constructor arguments, parameter names (checkpoints load with strict=True) and initialisation follow the reference."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .sample_shuffle import SampleShuffle1D


class ResidualBlock(nn.Module):
    """conv3 - ReLU - conv3 plus identity (models/edsr_1d.py:8-19)."""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = nn.Conv1d(channels, channels, kernel_size=3, stride=1, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv1d(channels, channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x))) + x


class EDSR_1D(nn.Module):
    """models/edsr_1d.py:22-45: input conv + ReLU, `num_blocks` residual blocks, mid conv with the long skip,
    SampleShuffle1D (C = num_features / upscale_factor channels survive), output conv."""

    def __init__(self, num_channels=1, num_features=64, num_blocks=8, upscale_factor=4):
        super().__init__()
        self.conv_input = nn.Conv1d(num_channels, num_features, kernel_size=3, stride=1, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.residual_blocks = nn.ModuleList([ResidualBlock(num_features) for _ in range(num_blocks)])
        self.conv_mid = nn.Conv1d(num_features, num_features, kernel_size=3, stride=1, padding=1)
        self.upscale = SampleShuffle1D(upscale_factor)
        self.conv_output = nn.Conv1d(num_features // upscale_factor, num_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        first = self.relu(self.conv_input(x))
        out = first
        for block in self.residual_blocks:
            out = block(out)
        out = self.conv_mid(out) + first
        return self.conv_output(self.upscale(out))


class ESPCN_1D(nn.Module):
    """models/espcn_1d.py:8-36: conv5 - tanh - conv3 - tanh - conv3 - SampleShuffle1D - sigmoid, with the reference's
    normal initialisation (std 0.001 for the layer fed by 32 channels, He-style otherwise, zero biases)."""

    def __init__(self, upscale_factor):
        super().__init__()
        self.conv1 = nn.Conv1d(1, 64, 5, 1, 2)
        self.conv2 = nn.Conv1d(64, 32, 3, 1, 1)
        self.conv3 = nn.Conv1d(32, upscale_factor, 3, 1, 1)
        self.sample_shuffle = SampleShuffle1D(upscale_factor)
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                std = 0.001 if m.in_channels == 32 else (2.0 / (m.out_channels * m.weight[0][0].numel())) ** 0.5
                nn.init.normal_(m.weight.data, 0.0, std)
                nn.init.zeros_(m.bias.data)

    def forward(self, x):
        x = torch.tanh(self.conv1(x))
        x = torch.tanh(self.conv2(x))
        return torch.sigmoid(self.sample_shuffle(self.conv3(x)))

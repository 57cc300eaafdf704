"""Pre-activation ResNet back-end as a torch module over the HIP front end.

Mirrors model/resnet.py:47-190 (PreActBlock / PreActBottleneck / ResNet) and the glue of model/wav2vec2_resnet_nll.py:36-74
(first_bn -> SELU on the [bz, 1, T, 128] feature map, then the ResNet; first_bn1 is defined and unused there) with the same
parameter names, so reference checkpoints load.  Like the AASIST back-end it is torch-composed for now; its 2-D convolutions
(1-2 GMAC per utterance) are the heavier of the two and the natural next MFMA kernel.
Reference detail kept: `_make_layer` builds a `downsample` Sequential and hands it to the block positionally, where it is
swallowed by *args (resnet.py:150-157,51): it is never registered, so it has no state-dict entries here either.
"""
import torch
import torch.nn.functional as F
from torch import nn

DEFAULT_RESNET = {"num_nodes": 3, "enc_dim": 256, "resnet_type": "18", "nclasses": 2}


class PreActBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False))

    def forward(self, x):
        out = F.relu(self.bn1(x))
        shortcut = self.shortcut(out) if hasattr(self, "shortcut") else x
        out = self.conv2(F.relu(self.bn2(self.conv1(out))))
        return out + shortcut


class PreActBottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, self.expansion * planes, kernel_size=1, bias=False)
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False))

    def forward(self, x):
        out = F.relu(self.bn1(x))
        shortcut = self.shortcut(out) if hasattr(self, "shortcut") else x
        out = self.conv1(out)
        out = self.conv2(F.relu(self.bn2(out)))
        out = self.conv3(F.relu(self.bn3(out)))
        return out + shortcut


RESNET_CONFIGS = {"18": ([2, 2, 2, 2], PreActBlock), "28": ([3, 4, 6, 3], PreActBlock), "34": ([3, 4, 6, 3], PreActBlock),
                  "50": ([3, 4, 6, 3], PreActBottleneck), "101": ([3, 4, 23, 3], PreActBottleneck)}


class ResNet(nn.Module):
    def __init__(self, num_nodes=3, enc_dim=256, resnet_type="18", nclasses=2):
        super().__init__()
        layers, block = RESNET_CONFIGS[str(resnet_type)]
        self.in_planes = 16
        self.conv1 = nn.Conv2d(1, 16, kernel_size=(9, 3), stride=(3, 1), padding=(1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.conv5 = nn.Conv2d(512 * block.expansion, 256, kernel_size=(num_nodes, 3), stride=(1, 1), padding=(0, 1), bias=False)
        self.bn5 = nn.BatchNorm2d(256)
        self.fc = nn.Linear(256, nclasses)

    def _make_layer(self, block, planes, num_blocks, stride):
        blocks = [block(self.in_planes, planes, stride)]
        self.in_planes = planes * block.expansion
        for _ in range(1, num_blocks):
            blocks.append(block(self.in_planes, planes, 1))
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = F.relu(self.bn5(self.conv5(x))).squeeze(2)
        if x.dim() == 3:
            x = x.unsqueeze(2)
        emb = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
        return self.fc(emb), emb


class ResNetHead(nn.Module):
    """feats [bz, T, 128] (LL output) -> (logits [bz, nclasses], emb [bz, 256])."""

    def __init__(self, cfg=None):
        super().__init__()
        self.first_bn = nn.BatchNorm2d(1)
        self.first_bn1 = nn.BatchNorm2d(64)      # defined, never used (wav2vec2_resnet_nll.py:38): state-dict compatibility
        self.resnet = ResNet(**(cfg or DEFAULT_RESNET))

    def forward(self, feats):
        return self.resnet(F.selu(self.first_bn(feats.unsqueeze(1))))

"""ORACLE (test infrastructure; never imported by the product path).  CPU restatement of the `wav2vec2_resnet_nll` back-end as ONE
functional forward over a state dict: model/wav2vec2_resnet_nll.py:51-74 (first_bn -> SELU on the [bz, 1, T, 128] map) and
model/resnet.py:47-191 (conv 9x3 / stride (3,1) -> BN -> ReLU -> four stages of pre-activation blocks -> conv (num_nodes x 3) ->
BN -> ReLU -> global average -> fc).  The network's shape is read off the parameter names (which are the reference's), not
re-declared: a stage `layerS` has as many blocks as the dict holds, a block with `conv3` is a bottleneck, a block with
`shortcut.0.weight` projects its skip path, stage S > 1 strides its first block by 2.  Pinned to the reference's own Model by
tests/test_resnet_cpu.py through tests/golden/resnet.npz (oracle/gen_golden.py::gen_resnet imports the reference).
"""
import torch
import torch.nn.functional as F
from torch import nn


class ParamTree(nn.Module):
    """Parameters / buffers registered under dotted state-dict names (so load_state_dict / named_parameters work), nothing else."""

    def __init__(self, shapes):
        super().__init__()
        for name, shape in shapes.items():
            mod, parts = self, name.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            leaf = parts[-1]
            if leaf in ("running_mean", "running_var"):
                mod.register_buffer(leaf, torch.zeros(shape))
            elif leaf == "num_batches_tracked":
                mod.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
            else:
                mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))

    def tensors(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d


def forward(t, feats, training):
    """t: {reference state-dict name: tensor} (running statistics are updated in place when training); feats [bz, T, 128]
    -> (logits [bz, nclasses], emb [bz, 256])."""
    def bn(x, name):
        if training:
            t[name + ".num_batches_tracked"] += 1
        return F.batch_norm(x, t[name + ".running_mean"], t[name + ".running_var"], t[name + ".weight"], t[name + ".bias"], training, 0.1, 1e-5)

    x = F.selu(bn(feats.unsqueeze(1), "first_bn"))
    x = F.relu(bn(F.conv2d(x, t["resnet.conv1.weight"], None, (3, 1), (1, 1)), "resnet.bn1"))
    for s in (1, 2, 3, 4):
        j = 0
        while "resnet.layer%d.%d.bn1.weight" % (s, j) in t:
            p = "resnet.layer%d.%d." % (s, j)
            stride = 2 if (s > 1 and j == 0) else 1
            a = F.relu(bn(x, p + "bn1"))
            skip = F.conv2d(a, t[p + "shortcut.0.weight"], None, stride) if (p + "shortcut.0.weight") in t else x
            if (p + "conv3.weight") in t:           # bottleneck: 1x1, 3x3 (strided), 1x1
                h = F.conv2d(a, t[p + "conv1.weight"])
                h = F.conv2d(F.relu(bn(h, p + "bn2")), t[p + "conv2.weight"], None, stride, 1)
                h = F.conv2d(F.relu(bn(h, p + "bn3")), t[p + "conv3.weight"])
            else:
                h = F.conv2d(a, t[p + "conv1.weight"], None, stride, 1)
                h = F.conv2d(F.relu(bn(h, p + "bn2")), t[p + "conv2.weight"], None, 1, 1)
            x = h + skip
            j += 1
    x = F.relu(bn(F.conv2d(x, t["resnet.conv5.weight"], None, 1, (0, 1)), "resnet.bn5"))
    emb = x.mean(dim=(2, 3))
    return F.linear(emb, t["resnet.fc.weight"], t["resnet.fc.bias"]), emb

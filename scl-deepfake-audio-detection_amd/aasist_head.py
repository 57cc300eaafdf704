"""AASIST graph-attention back-end on HIP kernels, channels-last.

The arithmetic is the reference's — model/wav2vec2_aasist.py:62-155 (GraphAttentionLayer), :158-332 (HtrgGraphAttentionLayer), :336-374
(GraphPool), :377-433 (Residual_block), :436-604 (Model) — and so are the parameter names, so its checkpoints load.  The execution
is not torch's: the spectro-temporal map lives as [B, 42, T/3, C] (channels last); the 2-D convolutions are implicit GEMMs on the
exact-fp32 matrix-core kernel (hipnn.conv2d), every BatchNorm(+SELU) is a fused HIP kernel pair (hipnn.batch_norm), the pairwise
attention scores are the fused kernels of csrc/gat.hip, projections / att @ x / the 1x1 attention convolutions are the fp32 GEMM
(hipnn.linear / hipnn.bmm), the 3x3 max pool is csrc/nn.hip.  torch carries tensors, the autograd tape and the remaining
element-wise glue (soft-max over <= 66 nodes, top-k, gathers).  nn.Linear / nn.BatchNorm* objects below are parameter and buffer
containers under the reference's state-dict names; their torch forwards are never called.

Reference quirks kept on purpose: Residual_block applies conv1 to its INPUT (the bn1+SELU result is discarded, :414-420; in training
that BatchNorm still updates its running statistics); the temporal / spectral attention pools share one 1x1-conv score map.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import graph, hipnn, resstack
from .gat import gat_score
from .resnet_head import ConvWeight

FUSED_STACK = os.environ.get("SCL_AASIST_FUSED", "1") == "1"      # 0: one autograd Function per layer (the round-2 composition)
FUSED_GRAPH = os.environ.get("SCL_AASIST_FUSED_GRAPH", os.environ.get("SCL_AASIST_FUSED", "1")) == "1"

UPSTREAM_AASIST = {"filts": [128, [1, 32], [32, 32], [32, 64], [64, 64]], "gat_dims": [64, 32],
                   "pool_ratios": [0.5, 0.5, 0.5, 0.5], "temperatures": [2.0, 2.0, 100.0, 100.0], "nclasses": 2}


def _xavier(*size):
    p = nn.Parameter(torch.empty(*size))
    nn.init.xavier_normal_(p)
    return p


def _lin(layer, x):
    return hipnn.linear(x, layer.weight, layer.bias)


class GraphAttentionLayer(nn.Module):
    def __init__(self, in_dim, out_dim, temperature=1.0):
        super().__init__()
        self.att_proj = nn.Linear(in_dim, out_dim)
        self.att_weight = _xavier(out_dim, 1)
        self.proj_with_att = nn.Linear(in_dim, out_dim)
        self.proj_without_att = nn.Linear(in_dim, out_dim)
        self.bn = nn.BatchNorm1d(out_dim)
        self.input_drop = nn.Dropout(p=0.2)
        self.temp = temperature

    def forward(self, x):
        x = self.input_drop(x)
        # s[i][j] = tanh(att_proj(x_i * x_j)) . att_weight, soft-max over j (fused: nothing of size N*N*D reaches HBM)
        att = F.softmax(gat_score(x, self.att_proj.weight, self.att_proj.bias, self.att_weight.t()) / self.temp, dim=-1)
        y = _lin(self.proj_with_att, hipnn.bmm(att, x)) + _lin(self.proj_without_att, x)
        return hipnn.batch_norm(y, self.bn, hipnn.ACT_SELU)


class HtrgGraphAttentionLayer(nn.Module):
    def __init__(self, in_dim, out_dim, temperature=1.0):
        super().__init__()
        self.proj_type1 = nn.Linear(in_dim, in_dim)
        self.proj_type2 = nn.Linear(in_dim, in_dim)
        self.att_proj = nn.Linear(in_dim, out_dim)
        self.att_projM = nn.Linear(in_dim, out_dim)
        self.att_weight11 = _xavier(out_dim, 1)
        self.att_weight22 = _xavier(out_dim, 1)
        self.att_weight12 = _xavier(out_dim, 1)
        self.att_weightM = _xavier(out_dim, 1)
        self.proj_with_att = nn.Linear(in_dim, out_dim)
        self.proj_without_att = nn.Linear(in_dim, out_dim)
        self.proj_with_attM = nn.Linear(in_dim, out_dim)
        self.proj_without_attM = nn.Linear(in_dim, out_dim)
        self.bn = nn.BatchNorm1d(out_dim)
        self.input_drop = nn.Dropout(p=0.2)
        self.temp = temperature

    def forward(self, x1, x2, master=None):
        n1, n2 = x1.size(1), x2.size(1)
        x = torch.cat([_lin(self.proj_type1, x1), _lin(self.proj_type2, x2)], dim=1)
        if master is None:
            master = x.mean(dim=1, keepdim=True)
        x = self.input_drop(x)
        # heterogeneous attention map: one weight vector per (type, type) block of the pair matrix
        a = torch.cat([self.att_weight11, self.att_weight22, self.att_weight12], dim=1).t()          # [3, D']
        att = F.softmax(gat_score(x, self.att_proj.weight, self.att_proj.bias, a, n1) / self.temp, dim=-1)
        # master node: attention of every node towards the master
        am = (torch.tanh(_lin(self.att_projM, x * master)) * self.att_weightM.t()).sum(-1, keepdim=True)     # [B, N, 1]
        am = F.softmax(am / self.temp, dim=-2)
        master = _lin(self.proj_with_attM, hipnn.bmm(am.transpose(1, 2), x)) + _lin(self.proj_without_attM, master)
        y = _lin(self.proj_with_att, hipnn.bmm(att, x)) + _lin(self.proj_without_att, x)
        y = hipnn.batch_norm(y, self.bn, hipnn.ACT_SELU)
        return y[:, :n1], y[:, n1:n1 + n2], master


class GraphPool(nn.Module):
    def __init__(self, k, in_dim, p):
        super().__init__()
        self.k = k
        self.proj = nn.Linear(in_dim, 1)
        self.drop = nn.Dropout(p=p) if p > 0 else nn.Identity()

    def forward(self, h):
        scores = torch.sigmoid(_lin(self.proj, self.drop(h)))                       # [B, N, 1]
        keep = max(int(h.size(1) * self.k), 1)
        idx = torch.topk(scores, keep, dim=1)[1].expand(-1, -1, h.size(2))
        return torch.gather(h * scores, 1, idx)


class Residual_block(nn.Module):
    def __init__(self, nb_filts, first=False):
        super().__init__()
        self.first = first
        if not first:
            self.bn1 = nn.BatchNorm2d(nb_filts[0])
        self.conv1 = ConvWeight(nb_filts[0], nb_filts[1], (2, 3), 1, (1, 1), bias=True)
        self.bn2 = nn.BatchNorm2d(nb_filts[1])
        self.conv2 = ConvWeight(nb_filts[1], nb_filts[1], (2, 3), 1, (0, 1), bias=True)
        self.downsample = nb_filts[0] != nb_filts[1]
        if self.downsample:
            self.conv_downsample = ConvWeight(nb_filts[0], nb_filts[1], (1, 3), 1, (0, 1), bias=True)

    def run(self, x):
        """x [B, H, W, C] channels-last."""
        if not self.first and self.training:
            hipnn.batch_norm(x.detach(), self.bn1, hipnn.ACT_SELU)       # result discarded by the reference (:414-420): running stats only
        dt = torch.float32
        out = self.conv2.conv(hipnn.batch_norm(self.conv1.conv(x, dt), self.bn2, hipnn.ACT_SELU), dt)
        return out + (self.conv_downsample.conv(x, dt) if self.downsample else x)


class AasistHead(nn.Module):
    """feats [B, T, 128] (LL output) -> (logits [B, nclasses], last_hidden [B, 5*gat_dims[1]])."""

    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or UPSTREAM_AASIST
        filts, gat, pr, temp = cfg["filts"], cfg["gat_dims"], cfg["pool_ratios"], cfg["temperatures"]
        self.first_bn = nn.BatchNorm2d(1)
        self.first_bn1 = nn.BatchNorm2d(64)
        self.drop = nn.Dropout(0.5)
        self.drop_way = nn.Dropout(0.2)
        self.encoder = nn.Sequential(nn.Sequential(Residual_block(filts[1], first=True)), nn.Sequential(Residual_block(filts[2])),
                                     nn.Sequential(Residual_block(filts[3])), nn.Sequential(Residual_block(filts[4])),
                                     nn.Sequential(Residual_block(filts[4])), nn.Sequential(Residual_block(filts[4])))
        # containers for attention.{0,2,3}: 1x1 conv (64 -> 128, bias), SELU, BatchNorm2d(128), 1x1 conv (128 -> 64, bias)
        self.attention = nn.Sequential(ConvWeight(64, 128, 1, bias=True), nn.SELU(), nn.BatchNorm2d(128), ConvWeight(128, 64, 1, bias=True))
        self.pos_S = nn.Parameter(torch.randn(1, 42, filts[-1][-1]))
        self.master1 = nn.Parameter(torch.randn(1, 1, gat[0]))
        self.master2 = nn.Parameter(torch.randn(1, 1, gat[0]))
        self.GAT_layer_S = GraphAttentionLayer(filts[-1][-1], gat[0], temperature=temp[0])
        self.GAT_layer_T = GraphAttentionLayer(filts[-1][-1], gat[0], temperature=temp[1])
        self.HtrgGAT_layer_ST11 = HtrgGraphAttentionLayer(gat[0], gat[1], temperature=temp[2])
        self.HtrgGAT_layer_ST12 = HtrgGraphAttentionLayer(gat[1], gat[1], temperature=temp[2])
        self.HtrgGAT_layer_ST21 = HtrgGraphAttentionLayer(gat[0], gat[1], temperature=temp[2])
        self.HtrgGAT_layer_ST22 = HtrgGraphAttentionLayer(gat[1], gat[1], temperature=temp[2])
        self.pool_S = GraphPool(pr[0], gat[0], 0.3)
        self.pool_T = GraphPool(pr[1], gat[0], 0.3)
        self.pool_hS1 = GraphPool(pr[2], gat[1], 0.3)
        self.pool_hT1 = GraphPool(pr[2], gat[1], 0.3)
        self.pool_hS2 = GraphPool(pr[2], gat[1], 0.3)
        self.pool_hT2 = GraphPool(pr[2], gat[1], 0.3)
        self.out_layer = nn.Linear(5 * gat[1], cfg["nclasses"])

    def forward(self, feats):
        # [B, T, 128] -> the single-channel map [B, 128 (frequency bins), T] -> 3x3 max pool -> [B, 42, T/3, 1] channels last
        x = hipnn.max_pool3(feats.transpose(1, 2)).unsqueeze(-1)
        x = hipnn.batch_norm(x, self.first_bn, hipnn.ACT_SELU)
        blocks = [blk[0] for blk in self.encoder]
        fused_stack = FUSED_STACK and resstack.supported(blocks, x.shape[2])
        if fused_stack and resstack.attn_supported(self.first_bn1, self.attention, self.pos_S, blocks[-1].conv2.weight.shape[0]) and x.shape[1] == self.pos_S.shape[1]:
            # the six Residual_blocks, first_bn1 + SELU, the attention block and both attention poolings as one autograd node (csrc/resstack.hip)
            e_S, e_T = resstack.res_stack_pool(x, blocks, self.first_bn1, self.attention, self.pos_S)
        else:
            if fused_stack:
                x = resstack.res_stack(x, blocks)          # the six Residual_blocks as one autograd node
            else:
                for blk in blocks:
                    x = blk.run(x)
            x = hipnn.batch_norm(x, self.first_bn1, hipnn.ACT_SELU)                    # [B, 42, T/3, 64]
            a0, a2, a3 = self.attention[0], self.attention[2], self.attention[3]
            w = hipnn.linear(x, a0.weight.view(a0.weight.shape[0], -1), a0.bias)
            w = hipnn.batch_norm(F.selu(w), a2, hipnn.ACT_NONE)
            w = hipnn.linear(w, a3.weight.view(a3.weight.shape[0], -1), a3.bias)      # one score map for both poolings
            e_S = (x * F.softmax(w, dim=2)).sum(2) + self.pos_S                        # spectral nodes  [B, 42, 64]
            e_T = (x * F.softmax(w, dim=1)).sum(1)                                     # temporal nodes  [B, T/3, 64]
        if FUSED_GRAPH and graph.supported(self, e_S.shape[1], e_T.shape[1]):
            return graph.graph_module(e_S, e_T, self)          # the whole graph module as one autograd node (csrc/graph.hip)
        return AasistHead.graph_unfused(self, e_S, e_T)

    def graph_unfused(self, e_S, e_T):
        """The graph module as a composition of per-operation autograd Functions (round 2): the fall-back for graph sizes the fused
        kernels do not take, and the arm tests/test_graph_gpu.py compares them with."""
        out_S = self.pool_S(self.GAT_layer_S(e_S))
        out_T = self.pool_T(self.GAT_layer_T(e_T))
        # two heterogeneous branches
        res = []
        for l1, l2, pS, pT, m in ((self.HtrgGAT_layer_ST11, self.HtrgGAT_layer_ST12, self.pool_hS1, self.pool_hT1, self.master1),
                                  (self.HtrgGAT_layer_ST21, self.HtrgGAT_layer_ST22, self.pool_hS2, self.pool_hT2, self.master2)):
            t, s, mm = l1(out_T, out_S, master=m)
            s, t = pS(s), pT(t)
            t_aug, s_aug, m_aug = l2(t, s, master=mm)
            res.append((self.drop_way(t + t_aug), None, mm + m_aug, s + s_aug))
        (t1, _, m1, s1), (t2, _, m2, s2) = res
        s1, s2, m1, m2 = self.drop_way(s1), self.drop_way(s2), self.drop_way(m1), self.drop_way(m2)
        out_T, out_S, master = torch.max(t1, t2), torch.max(s1, s2), torch.max(m1, m2)
        last_hidden = torch.cat([out_T.abs().max(dim=1)[0], out_T.mean(dim=1), out_S.abs().max(dim=1)[0], out_S.mean(dim=1),
                                 master.squeeze(1)], dim=1)
        last_hidden = self.drop(last_hidden)
        return _lin(self.out_layer, last_hidden), last_hidden

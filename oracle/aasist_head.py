"""ORACLE (test infrastructure; never imported by the product path).  CPU restatement in plain torch of the AASIST back-end —
model/wav2vec2_aasist.py:62-155 (GraphAttentionLayer), :158-332 (HtrgGraphAttentionLayer), :336-374 (GraphPool), :377-433
(Residual_block), :436-604 (Model) — with the reference's parameter names.  Pinned to the reference's own classes by
tests/test_aasist_cpu.py through tests/golden/aasist.npz (oracle/gen_golden.py::gen_aasist imports the reference).  Reference quirks
kept: Residual_block applies conv1 to its INPUT (the bn1 + SELU result is discarded, :414-420); the temporal / spectral attention
pools share one 1x1-conv score map.  The GPU tests use it as the fp32 reference of scl_amd/aasist_head.py (HIP kernels).
"""
import torch
import torch.nn.functional as F
from torch import nn

UPSTREAM_AASIST = {"filts": [128, [1, 32], [32, 32], [32, 64], [64, 64]], "gat_dims": [64, 32],
                   "pool_ratios": [0.5, 0.5, 0.5, 0.5], "temperatures": [2.0, 2.0, 100.0, 100.0], "nclasses": 2}


def _xavier(*size):
    p = nn.Parameter(torch.empty(*size))
    nn.init.xavier_normal_(p)
    return p


def _pairwise(x):
    """[B, N, D] -> [B, N, N, D] element-wise products of every node pair."""
    return x.unsqueeze(2) * x.unsqueeze(1)


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
        att = torch.tanh(self.att_proj(_pairwise(x))) @ self.att_weight          # [B, N, N, 1]
        att = F.softmax(att / self.temp, dim=-2).squeeze(-1)
        y = self.proj_with_att(att @ x) + self.proj_without_att(x)
        y = self.bn(y.reshape(-1, y.shape[-1])).view_as(y)
        return F.selu(y)


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
        x = torch.cat([self.proj_type1(x1), self.proj_type2(x2)], dim=1)
        if master is None:
            master = x.mean(dim=1, keepdim=True)
        x = self.input_drop(x)
        # heterogeneous attention map: one weight vector per (type, type) block
        h = torch.tanh(self.att_proj(_pairwise(x)))                                # [B, N, N, D']
        board = torch.zeros_like(h[..., :1])
        board[:, :n1, :n1] = h[:, :n1, :n1] @ self.att_weight11
        board[:, n1:, n1:] = h[:, n1:, n1:] @ self.att_weight22
        board[:, :n1, n1:] = h[:, :n1, n1:] @ self.att_weight12
        board[:, n1:, :n1] = h[:, n1:, :n1] @ self.att_weight12
        att = F.softmax(board / self.temp, dim=-2).squeeze(-1)
        # master node: attention of every node towards the master
        am = torch.tanh(self.att_projM(x * master)) @ self.att_weightM             # [B, N, 1]
        am = F.softmax(am / self.temp, dim=-2)
        master = self.proj_with_attM(am.squeeze(-1).unsqueeze(1) @ x) + self.proj_without_attM(master)
        y = self.proj_with_att(att @ x) + self.proj_without_att(x)
        y = F.selu(self.bn(y.reshape(-1, y.shape[-1])).view_as(y))
        return y[:, :n1], y[:, n1:n1 + n2], master


class GraphPool(nn.Module):
    def __init__(self, k, in_dim, p):
        super().__init__()
        self.k = k
        self.proj = nn.Linear(in_dim, 1)
        self.drop = nn.Dropout(p=p) if p > 0 else nn.Identity()

    def forward(self, h):
        scores = torch.sigmoid(self.proj(self.drop(h)))                             # [B, N, 1]
        keep = max(int(h.size(1) * self.k), 1)
        idx = torch.topk(scores, keep, dim=1)[1].expand(-1, -1, h.size(2))
        return torch.gather(h * scores, 1, idx)


class Residual_block(nn.Module):
    def __init__(self, nb_filts, first=False):
        super().__init__()
        self.first = first
        if not first:
            self.bn1 = nn.BatchNorm2d(nb_filts[0])
        self.conv1 = nn.Conv2d(nb_filts[0], nb_filts[1], kernel_size=(2, 3), padding=(1, 1), stride=1)
        self.bn2 = nn.BatchNorm2d(nb_filts[1])
        self.conv2 = nn.Conv2d(nb_filts[1], nb_filts[1], kernel_size=(2, 3), padding=(0, 1), stride=1)
        self.downsample = nb_filts[0] != nb_filts[1]
        if self.downsample:
            self.conv_downsample = nn.Conv2d(nb_filts[0], nb_filts[1], padding=(0, 1), kernel_size=(1, 3), stride=1)

    def forward(self, x):
        if not self.first:
            F.selu(self.bn1(x))            # computed and discarded by the reference (:414-420); kept for the BN running stats
        out = self.conv2(F.selu(self.bn2(self.conv1(x))))
        return out + (self.conv_downsample(x) if self.downsample else x)


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
        self.attention = nn.Sequential(nn.Conv2d(64, 128, kernel_size=(1, 1)), nn.SELU(), nn.BatchNorm2d(128),
                                       nn.Conv2d(128, 64, kernel_size=(1, 1)))
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
        x = feats.transpose(1, 2).unsqueeze(1)                                     # [B, 1, 128, T]
        x = F.selu(self.first_bn(F.max_pool2d(x, (3, 3))))
        x = F.selu(self.first_bn1(self.encoder(x)))                                # [B, 64, 42, T/3]
        w = self.attention(x)
        e_S = (x * F.softmax(w, dim=-1)).sum(-1).transpose(1, 2) + self.pos_S      # spectral nodes  [B, 42, 64]
        e_T = (x * F.softmax(w, dim=-2)).sum(-2).transpose(1, 2)                   # temporal nodes  [B, T/3, 64]
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
        return self.out_layer(last_hidden), last_hidden

"""The graph module of the AASIST back-end — GAT_layer_S / GAT_layer_T, pool_S / pool_T, the two heterogeneous branches
(HtrgGAT_layer_ST11 -> pools -> ST12, ST21 -> pools -> ST22), drop_way, branch max, read-out, out_layer
(model/wav2vec2_aasist.py:62-374, 545-604) — as ONE autograd node over the kernels of csrc/graph.hip (+ csrc/gat.hip for the pairwise
scores): 13 launches forward, ~20 backward, parameter gradients written into `p.grad` by one reducing launch.

    graph_module(e_S [B, nS, 64], e_T [B, nT, 64], head) -> (logits [B, nclasses], last_hidden [B, 5 * gat_dims[1]])

`head` supplies the parameter / buffer objects under the reference's names (GAT_layer_S, pool_hT1, master1, out_layer, ...), the
dropout probabilities and the training flag.  Dropout masks are counter hashes of (site seed, element), redrawn every call in
training and recomputed in the backward; in eval mode BatchNorm uses the running statistics.
"""
import ctypes
import weakref

import torch

from . import lib as L
from . import ops

_SEED = [0x3C6EF372]


def _next_seed():
    _SEED[0] = (_SEED[0] * 1664525 + 1013904223) & 0x7FFFFFFF
    return _SEED[0]


def seed(value):
    """Re-seed the graph module's dropout stream (per --seed / per data-parallel rank)."""
    _SEED[0] = int(value) & 0x7FFFFFFF


def supported(head, nS, nT):
    try:
        gS, gT = head.GAT_layer_S, head.GAT_layer_T
        d0 = gS.att_proj.weight.shape[1]
        g0 = gS.att_proj.weight.shape[0]
        g1 = head.HtrgGAT_layer_ST11.att_proj.weight.shape[0]
        kS, kT = max(int(nS * head.pool_S.k), 1), max(int(nT * head.pool_T.k), 1)
        return d0 == 64 and g0 == 64 and g1 in (32, 64) and nT <= 80 and nS <= 80 and kS + kT <= 64 and head.out_layer.weight.shape[0] <= 8 and 5 * g1 <= 256
    except AttributeError:
        return False


def _grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Plan:
    def __init__(self, head, B, nS, nT, dev):
        self.key = (B, nS, nT)
        self.dev, self.busy = dev, False
        g0 = head.GAT_layer_S.att_proj.weight.shape[0]
        g1 = head.HtrgGAT_layer_ST11.att_proj.weight.shape[0]
        self.g0, self.g1 = g0, g1
        self.kS, self.kT = max(int(nS * head.pool_S.k), 1), max(int(nT * head.pool_T.k), 1)
        self.kS2, self.kT2 = max(int(self.kS * head.pool_hS1.k), 1), max(int(self.kT * head.pool_hT1.k), 1)
        self.N1, self.N2 = self.kT + self.kS, self.kT2 + self.kS2
        f = lambda *s: torch.zeros(*s, device=dev)
        self.layers = {}
        for name, N, D, Do, master in (("S", nS, 64, g0, False), ("T", nT, 64, g0, False), ("11", self.N1, g0, g1, True), ("21", self.N1, g0, g1, True),
                                       ("12", self.N2, g1, g1, True), ("22", self.N2, g1, g1, True)):
            nb = ops.gat_score_nblocks(N)
            d = dict(N=N, D=D, Do=Do, master=master, xd=f(B, N, D), S=f(B, N, N), g=f(B, N, D), y=f(B, N, Do), dz=f(B, N, Do), dS=f(B, N, N),
                     dxd_a=f(B, N, D), dxd_b=f(B, N, D), stats=f(4 * Do), bstats=f(2 * Do), acc=torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev),
                     ticket=torch.zeros(1, dtype=torch.int32, device=dev), dP=torch.empty(B * N * N * D, device=dev),
                     part=torch.empty(nb * B, Do * D + 4 * Do, device=dev), nb=nb, a3=f(3, Do))
            if master:
                d.update(am=f(B, N), gm=f(B, D), tM=f(B, N, Do), mout=f(B, Do), d_min=f(B, D))
            self.layers[name] = d
        # pool units: (source layer, first row, rows in, rows kept)
        def unit(n_in, K, Dp):
            return dict(h=f(B, n_in, Dp), sc=f(B, n_in), idx=torch.zeros(B, K, dtype=torch.int32, device=dev), pooled=f(B, K, Dp))
        self.units = {"T": unit(nT, self.kT, g0), "S": unit(nS, self.kS, g0),
                      "T1": unit(self.kT, self.kT2, g1), "S1": unit(self.kS, self.kS2, g1), "T2": unit(self.kT, self.kT2, g1), "S2": unit(self.kS, self.kS2, g1)}
        self.fin = [dict(dTp=f(B, self.kT2, g1), dSp=f(B, self.kS2, g1), dm1=f(B, g1), dm2=f(B, g1)) for _ in range(2)]
        self.de = (f(B, nS, 64), f(B, nT, 64))
        self.slab = None
        self.sig, self.fwd_calls, self.bwd_calls = None, None, None


_PLANS = []


def _acquire(head, B, nS, nT, dev, hold):
    key = (B, nS, nT)
    for pl in _PLANS:
        if pl.key == key and pl.dev == dev and not pl.busy:
            pl.busy = hold
            pl.gen = getattr(pl, "gen", 0) + 1
            return pl
    # never hand out a plan whose backward is still pending (its saved maps, statistics, dropout seeds and recorded pointers belong to
    # that node): a further forward of the same shape gets a fresh plan, up to a hard cap.  Plans of nodes that are freed without a
    # backward are released by the finalizer _hold() attaches to the node.
    same = [pl for pl in _PLANS if pl.key == key and pl.dev == dev]
    if len(same) >= MAX_LIVE_PLANS:
        raise RuntimeError("%d forward passes of shape %r are waiting for their backward; run the backwards (or drop the graphs) "
                           "before another forward of this shape" % (len(same), key))
    idle = [pl for pl in _PLANS if not pl.busy]
    if len(_PLANS) >= 8 and idle:
        _PLANS.remove(idle[0])
    pl = _Plan(head, B, nS, nT, dev)
    pl.busy = hold
    pl.gen = 1
    _PLANS.append(pl)
    return pl


MAX_LIVE_PLANS = 16


def _storages(*ts):
    """What a plan keeps of the tensors its in-flight kernels read / write: the STORAGE (the memory stays allocated), never the tensor
    object — an output tensor carries its grad_fn, and a plan that held it would keep its own autograd node (and through ctx.pl itself)
    alive until the next forward of the shape, so a graph dropped without a backward could never hand its plan back."""
    out = []
    for t in ts:
        if isinstance(t, (tuple, list)):
            out.append(_storages(*t))
        elif torch.is_tensor(t):
            out.append(t.untyped_storage())
        else:
            out.append(t)
    return tuple(out)


def _release(pl, gen):
    if pl.gen == gen:
        pl.busy = False


def _hold(ctx, pl):
    """Ties the plan to the autograd node: generation stamp for the backward's ownership check, release when the node dies unused."""
    ctx.pl, ctx.gen = pl, pl.gen
    if pl.busy:
        weakref.finalize(ctx, _release, pl, pl.gen)


def _owned(ctx):
    pl = ctx.pl
    if pl.gen != ctx.gen:
        raise RuntimeError("this node's plan was re-used by a later forward (generation %d, node holds %d): its saved activations are gone" % (pl.gen, ctx.gen))
    return pl


def _ptr(t):
    return None if t is None else t.data_ptr()


def _bn(pl, lay, bnmod, nvalid, training):
    b = L.SclGraphBn()
    b.acc, b.ticket = lay["acc"].data_ptr(), lay["ticket"].data_ptr()
    b.gamma, b.beta, b.run_mean, b.run_var, b.nbt = (bnmod.weight.data_ptr(), bnmod.bias.data_ptr(), bnmod.running_mean.data_ptr(),
                                                     bnmod.running_var.data_ptr(), bnmod.num_batches_tracked.data_ptr())
    b.stats, b.bstats = lay["stats"].data_ptr(), lay["bstats"].data_ptr()
    b.dgamma, b.dbeta = _grad(bnmod.weight).data_ptr(), _grad(bnmod.bias).data_ptr()
    b.nvalid, b.eps, b.momentum, b.training = float(nvalid), float(bnmod.eps), 0.1 if bnmod.momentum is None else float(bnmod.momentum), 1 if training else 0
    return b


class _Slab:
    """Offsets of every parameter the graph kernels produce gradients for, inside one slab row."""

    def __init__(self):
        self.off, self.n, self.params = {}, 0, []

    def add(self, p):
        if id(p) not in self.off:
            self.off[id(p)] = self.n
            self.params.append(p)
            self.n += (p.numel() + 3) // 4 * 4
        return self.off[id(p)]


def _layer_struct(pl, name, mod, B, training, slab, master_in=None, master_bs=0, d_mout=None, d_mout2=None):
    lay = pl.layers[name]
    s = L.SclGraphLayer()
    s.xd, s.S, s.g, s.y = lay["xd"].data_ptr(), lay["S"].data_ptr(), lay["g"].data_ptr(), lay["y"].data_ptr()
    s.Wa, s.ba, s.Wb, s.bb = (mod.proj_with_att.weight.data_ptr(), mod.proj_with_att.bias.data_ptr(), mod.proj_without_att.weight.data_ptr(),
                              mod.proj_without_att.bias.data_ptr())
    s.N, s.D, s.Do, s.has_master, s.inv_temp = lay["N"], lay["D"], lay["Do"], 1 if lay["master"] else 0, 1.0 / float(mod.temp)
    s.bn = _bn(pl, lay, mod.bn, B * lay["N"], training)
    s.dz, s.dS, s.dxd = lay["dz"].data_ptr(), lay["dS"].data_ptr(), lay["dxd_a"].data_ptr()
    s.slab, s.slab_bs = pl.slab.data_ptr(), pl.slab.shape[1]
    s.o_Wa, s.o_ba, s.o_Wb, s.o_bb = (slab.add(mod.proj_with_att.weight), slab.add(mod.proj_with_att.bias), slab.add(mod.proj_without_att.weight),
                                      slab.add(mod.proj_without_att.bias))
    if lay["master"]:
        s.min, s.min_bs = master_in, master_bs
        s.WM, s.bM, s.aM = mod.att_projM.weight.data_ptr(), mod.att_projM.bias.data_ptr(), mod.att_weightM.data_ptr()
        s.WaM, s.baM, s.WbM, s.bbM = (mod.proj_with_attM.weight.data_ptr(), mod.proj_with_attM.bias.data_ptr(), mod.proj_without_attM.weight.data_ptr(),
                                      mod.proj_without_attM.bias.data_ptr())
        s.am, s.gm, s.tM, s.mout, s.d_min = lay["am"].data_ptr(), lay["gm"].data_ptr(), lay["tM"].data_ptr(), lay["mout"].data_ptr(), lay["d_min"].data_ptr()
        s.d_mout, s.d_mout2 = _ptr(d_mout), _ptr(d_mout2)
        s.o_WM, s.o_bM, s.o_aM = slab.add(mod.att_projM.weight), slab.add(mod.att_projM.bias), slab.add(mod.att_weightM)
        s.o_WaM, s.o_baM, s.o_WbM, s.o_bbM = (slab.add(mod.proj_with_attM.weight), slab.add(mod.proj_with_attM.bias), slab.add(mod.proj_without_attM.weight),
                                              slab.add(mod.proj_without_attM.bias))
    return s


def _a3(lay, mod):
    """[3, Do] attention vectors of the pairwise score: (att_weight) or (att_weight11, att_weight22, att_weight12)."""
    a = lay["a3"]
    ws = [mod.att_weight] if hasattr(mod, "att_weight") else [mod.att_weight11, mod.att_weight22, mod.att_weight12]
    Do = a.shape[1]
    p0 = ws[0].data_ptr()
    room = ws[0].untyped_storage().nbytes() - 4 * ws[0].storage_offset() >= 12 * Do      # the kernel stages 3 * Do floats
    if room and (len(ws) == 1 or (ws[1].data_ptr() == p0 + 4 * Do and ws[2].data_ptr() == p0 + 8 * Do)):
        return ws[0]                      # a view into the flat parameter buffer (the three vectors are adjacent there): read in place
    with torch.no_grad():
        for i, w in enumerate(ws):
            a[i].copy_(w.detach().view(-1))
    lay["a3_src"] = ws                    # stand-alone parameter tensors: the copy is refreshed before every replay
    return a


def _score_fwd(pl, name, mod, B):
    lay = pl.layers[name]
    n1 = lay["N"] if not lay["master"] else (pl.kT if name in ("11", "21") else pl.kT2)
    a3 = _a3(lay, mod)
    lay["a3_used"], lay["n1"] = a3, n1
    ops.gat_score_fwd(lay["xd"], mod.att_proj.weight, mod.att_proj.bias, a3, lay["S"], B, lay["N"], lay["D"], lay["Do"], n1)


def _score_bwd(pl, name, mod, B):
    lay = pl.layers[name]
    ops.gat_score_bwd(lay["xd"], mod.att_proj.weight, mod.att_proj.bias, lay["a3_used"], lay["dS"], lay["dP"], lay["part"], lay["dxd_b"], B, lay["N"], lay["D"],
                      lay["Do"], lay["n1"])


def _unit(pl, uname, src_lay, src_bn, row0, pool, Wt_mod, row_out, slab, seeds, training, B, d_res=None):
    un = pl.units[uname]
    u = L.SclGraphPoolUnit()
    n_in, K = un["h"].shape[1], un["idx"].shape[1]
    u.ysrc, u.src_n, u.row0, u.n_in, u.K = src_lay["y"].data_ptr(), src_lay["N"], row0, n_in, K
    u.stats = src_lay["stats"].data_ptr()
    u.pw, u.pb = pool.proj.weight.data_ptr(), pool.proj.bias.data_ptr()
    p = float(pool.drop.p) if (training and isinstance(pool.drop, torch.nn.Dropout)) else 0.0
    u.pool_seed, u.pool_p = seeds["pool_" + uname], p
    u.h, u.sc, u.idx, u.pooled = un["h"].data_ptr(), un["sc"].data_ptr(), un["idx"].data_ptr(), un["pooled"].data_ptr()
    u.Wt, u.bt = Wt_mod.weight.data_ptr(), Wt_mod.bias.data_ptr()
    u.row_out = row_out
    u.d_res, u.dz = _ptr(d_res), src_lay["dz"].data_ptr()
    u.o_pw, u.o_pb, u.o_Wt, u.o_bt = slab.add(pool.proj.weight), slab.add(pool.proj.bias), slab.add(Wt_mod.weight), slab.add(Wt_mod.bias)
    u.bn = src_bn
    return u


def _structs(pl, head, B, training, seeds):
    """Every kernel argument block of one forward + backward (pointers are stable per plan and parameter set)."""
    slab = _Slab()
    if pl.slab is None:
        pl.slab = torch.zeros(B, 1, device=pl.dev)
    H = head
    mods = {"S": H.GAT_layer_S, "T": H.GAT_layer_T, "11": H.HtrgGAT_layer_ST11, "12": H.HtrgGAT_layer_ST12, "21": H.HtrgGAT_layer_ST21, "22": H.HtrgGAT_layer_ST22}
    st = {}
    for rnd in range(2):          # first round only sizes the slab
        slab = _Slab()
        Ls = {}
        Ls["S"] = _layer_struct(pl, "S", mods["S"], B, training, slab)
        Ls["T"] = _layer_struct(pl, "T", mods["T"], B, training, slab)
        Ls["12"] = _layer_struct(pl, "12", mods["12"], B, training, slab, pl.layers["11"]["mout"].data_ptr(), pl.g1, d_mout=pl.fin[0]["dm2"])
        Ls["22"] = _layer_struct(pl, "22", mods["22"], B, training, slab, pl.layers["21"]["mout"].data_ptr(), pl.g1, d_mout=pl.fin[1]["dm2"])
        Ls["11"] = _layer_struct(pl, "11", mods["11"], B, training, slab, H.master1.data_ptr(), 0, d_mout=pl.fin[0]["dm1"], d_mout2=pl.layers["12"]["d_min"])
        Ls["21"] = _layer_struct(pl, "21", mods["21"], B, training, slab, H.master2.data_ptr(), 0, d_mout=pl.fin[1]["dm1"], d_mout2=pl.layers["22"]["d_min"])
        p_in = lambda m: float(m.input_drop.p) if training else 0.0
        pres = {}
        for tag, names, pools, srcs in (("1", ("11", "21"), ((H.pool_T, H.pool_S), (H.pool_T, H.pool_S)), (("T", "S"), ("T", "S"))),
                                        ("2", ("12", "22"), ((H.pool_hT1, H.pool_hS1), (H.pool_hT2, H.pool_hS2)), (("11", "11"), ("21", "21")))):
            arr = (L.SclGraphPre * 2)()
            for i, nm in enumerate(names):
                mod = mods[nm]
                I = arr[i]
                if tag == "1":
                    uT = _unit(pl, "T", pl.layers["T"], Ls["T"].bn, 0, pools[i][0], mod.proj_type1, 0, slab, seeds, training, B)
                    uS = _unit(pl, "S", pl.layers["S"], Ls["S"].bn, 0, pools[i][1], mod.proj_type2, pl.kT, slab, seeds, training, B)
                    I.same_bn, I.store_common = 0, 1 if i == 0 else 0
                else:
                    src = srcs[i][0]
                    uT = _unit(pl, "T" + str(i + 1), pl.layers[src], Ls[src].bn, 0, pools[i][0], mod.proj_type1, 0, slab, seeds, training, B, d_res=pl.fin[i]["dTp"])
                    uS = _unit(pl, "S" + str(i + 1), pl.layers[src], Ls[src].bn, pl.kT, pools[i][1], mod.proj_type2, pl.kT2, slab, seeds, training, B, d_res=pl.fin[i]["dSp"])
                    I.same_bn, I.store_common = 1, 1
                I.u[0], I.u[1] = uT, uS
                lay = pl.layers[nm]
                I.xd, I.dxd_a, I.dxd_b = lay["xd"].data_ptr(), lay["dxd_a"].data_ptr(), lay["dxd_b"].data_ptr()
                I.in_seed, I.in_p = seeds["in_" + nm], p_in(mod)
                I.Dp, I.N = lay["D"], lay["N"]
                I.slab, I.slab_bs = pl.slab.data_ptr(), pl.slab.shape[1]
            pres[tag] = arr
        fin = L.SclGraphFinal()
        for i, (n1, n2) in enumerate((("11", "12"), ("21", "22"))):
            br = fin.br[i]
            l1, l2 = pl.layers[n1], pl.layers[n2]
            br.y2, br.stats2 = l2["y"].data_ptr(), l2["stats"].data_ptr()
            br.Tp, br.Sp = pl.units["T" + str(i + 1)]["pooled"].data_ptr(), pl.units["S" + str(i + 1)]["pooled"].data_ptr()
            br.m1, br.m2 = l1["mout"].data_ptr(), l2["mout"].data_ptr()
            for k, site in enumerate(("T", "S", "M")):
                br.way_seed[k] = seeds["way_%s%d" % (site, i)]
            f = pl.fin[i]
            br.dz2, br.dTp, br.dSp, br.dm1, br.dm2 = l2["dz"].data_ptr(), f["dTp"].data_ptr(), f["dSp"].data_ptr(), f["dm1"].data_ptr(), f["dm2"].data_ptr()
            br.bn = Ls[n2].bn
        fin.Wout, fin.bout = H.out_layer.weight.data_ptr(), H.out_layer.bias.data_ptr()
        fin.slab, fin.slab_bs = pl.slab.data_ptr(), pl.slab.shape[1]
        fin.o_Wout, fin.o_bout = slab.add(H.out_layer.weight), slab.add(H.out_layer.bias)
        fin.KT, fin.KS, fin.D, fin.NC = pl.kT2, pl.kS2, pl.g1, H.out_layer.weight.shape[0]
        fin.way_p = float(H.drop_way.p) if training else 0.0
        fin.drop_p, fin.drop_seed = (float(H.drop.p) if training else 0.0), seeds["drop"]
        if pl.slab.shape[1] >= slab.n:
            st = dict(L=Ls, pre=pres, fin=fin, slab=slab, mods=mods)
            break
        pl.slab = torch.zeros(B, slab.n, device=pl.dev)
    return st


def _eval_stats(pl, mods):
    for nm, mod in mods.items():
        lay, bn = pl.layers[nm], mod.bn
        ops._call("scl_rs_bn_eval_stats", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps),
                  lay["Do"], lay["stats"].data_ptr(), _S())


def _post(st, names, B, fwd):
    arr = (L.SclGraphLayer * 2)(st["L"][names[0]], st["L"][names[1]])
    ops._call("scl_graph_post_fwd" if fwd else "scl_graph_post_bwd", arr, 2, B, _S(), keep=arr)


_SEED_NAMES = ["in_S", "in_T", "in_11", "in_21", "in_12", "in_22", "pool_T", "pool_S", "pool_T1", "pool_S1", "pool_T2", "pool_S2", "drop"] + \
              ["way_%s%d" % (s, i) for s in "TSM" for i in (0, 1)]


def _signature(head, training):
    """What the recorded launch sequence of a plan depends on besides the plan's own buffers: parameter and gradient storage."""
    w, bnw = head.out_layer.weight, head.GAT_layer_S.bn.weight
    return (torch.cuda.current_stream().cuda_stream, id(head), training, w.data_ptr(), _grad(w).data_ptr(), bnw.data_ptr(), _grad(bnw).data_ptr(), head.pool_hT2.proj.weight.data_ptr(),
            _grad(head.master1).data_ptr(), float(head.drop.p), float(head.drop_way.p), float(head.GAT_layer_S.input_drop.p))


def _set_seeds(pl, seeds):
    """Per-call values inside the persistent argument blocks of a recorded program."""
    st = pl.st
    for tag, names in (("1", ("11", "21")), ("2", ("12", "22"))):
        arr = st["pre"][tag]
        for i, nm in enumerate(names):
            arr[i].in_seed = seeds["in_" + nm]
            if tag == "1":
                arr[i].u[0].pool_seed, arr[i].u[1].pool_seed = seeds["pool_T"], seeds["pool_S"]
            else:
                arr[i].u[0].pool_seed, arr[i].u[1].pool_seed = seeds["pool_T%d" % (i + 1)], seeds["pool_S%d" % (i + 1)]
    fin = st["fin"]
    for i in range(2):
        for k, site in enumerate("TSM"):
            fin.br[i].way_seed[k] = seeds["way_%s%d" % (site, i)]
    fin.drop_seed = seeds["drop"]


def _forward(pl, head, e_S, e_T, training):
    """First call of a (plan, parameter storage, mode): run the launch sequence while ops records it; later calls patch the per-call
    values (seeds, input / output pointers) and replay the recorded C calls — the host stays ahead of the ~45 small kernels."""
    B = e_S.shape[0]
    seeds = {n: (_next_seed() if training else 0) for n in _SEED_NAMES}
    pl.seeds_used = seeds
    sig = _signature(head, training)
    fin_out = (torch.empty(B, head.out_layer.weight.shape[0], device=pl.dev), torch.empty(B, 5 * pl.g1, device=pl.dev))
    if getattr(pl, "sig", None) == sig and pl.fwd_calls is not None:
        _set_seeds(pl, seeds)
        for lay in pl.layers.values():
            if lay.get("a3_src") is not None:
                with torch.no_grad():
                    for i, w in enumerate(lay["a3_src"]):
                        lay["a3"][i].copy_(w.detach().view(-1))
        fin = pl.st["fin"]
        fin.logits, fin.hidden = fin_out[0].data_ptr(), fin_out[1].data_ptr()
        a = pl.drop_entry[1]
        a[0], a[3], a[4], a[7] = e_S.data_ptr(), seeds["in_S"], e_T.data_ptr(), seeds["in_T"]
        pl.live = _storages(e_S, e_T, fin_out)
        ops.replay(pl.fwd_calls)
        return fin_out
    st = _structs(pl, head, B, training, seeds)
    pl.st, pl.sig, pl.fwd_calls, pl.bwd_calls = st, sig, None, None
    for lay in pl.layers.values():
        lay["a3_src"] = None
    mods = st["mods"]
    record = ops._rec() is None
    if record:
        ops.start_recording()
    if not training:
        _eval_stats(pl, mods)
    lS, lT = pl.layers["S"], pl.layers["T"]
    p_in = float(mods["S"].input_drop.p) if training else 0.0
    pl.drop_entry = ops._call("scl_graph_drop", e_S.data_ptr(), lS["xd"].data_ptr(), e_S.numel(), seeds["in_S"], e_T.data_ptr(), lT["xd"].data_ptr(), e_T.numel(),
                              seeds["in_T"], p_in, _S())
    pl.p_in0 = p_in
    _score_fwd(pl, "S", mods["S"], B)
    _score_fwd(pl, "T", mods["T"], B)
    _post(st, ("S", "T"), B, True)
    ops._call("scl_graph_pre_fwd", st["pre"]["1"], 1, B, _S(), keep=st["pre"]["1"])
    _score_fwd(pl, "11", mods["11"], B)
    _score_fwd(pl, "21", mods["21"], B)
    _post(st, ("11", "21"), B, True)
    ops._call("scl_graph_pre_fwd", st["pre"]["2"], 0, B, _S(), keep=st["pre"]["2"])
    _score_fwd(pl, "12", mods["12"], B)
    _score_fwd(pl, "22", mods["22"], B)
    _post(st, ("12", "22"), B, True)
    fin = st["fin"]
    fin.logits, fin.hidden = fin_out[0].data_ptr(), fin_out[1].data_ptr()
    ops._call("scl_graph_final_fwd", ctypes.byref(fin), B, _S(), keep=fin)
    pl.live = _storages(e_S, e_T, fin_out)
    if record:
        pl.fwd_calls = ops.stop_recording()
    return fin_out


def _reduce(pl, head, B):
    st = pl.st
    jobs = []
    P = pl.slab.shape[1]
    for p in st["slab"].params:
        jobs.append((pl.slab.data_ptr() + 4 * st["slab"].off[id(p)], _grad(p).data_ptr(), P, p.numel(), B))
    for nm, mparam in (("11", head.master1), ("21", head.master2)):
        jobs.append((pl.layers[nm]["d_min"].data_ptr(), _grad(mparam).data_ptr(), pl.layers[nm]["D"], pl.layers[nm]["D"], B))
    for nm, mod in st["mods"].items():
        lay = pl.layers[nm]
        Do, D = lay["Do"], lay["D"]
        ncol, nparts, base = Do * D + 4 * Do, lay["nb"] * B, lay["part"].data_ptr()
        jobs.append((base, _grad(mod.att_proj.weight).data_ptr(), ncol, Do * D, nparts))
        jobs.append((base + 4 * Do * D, _grad(mod.att_proj.bias).data_ptr(), ncol, Do, nparts))
        ws = [mod.att_weight] if hasattr(mod, "att_weight") else [mod.att_weight11, mod.att_weight22, mod.att_weight12]
        for t, w in enumerate(ws):
            jobs.append((base + 4 * (Do * D + Do + t * Do), _grad(w).data_ptr(), ncol, Do, nparts))
    for lo in range(0, len(jobs), 96):
        chunk = jobs[lo: lo + 96]
        arr = (L.SclGraphReduceJob * len(chunk))()
        for i, (src, dst, stride, n, nparts) in enumerate(chunk):
            arr[i] = L.SclGraphReduceJob(src, dst, stride, n, nparts)
        ops._call("scl_graph_reduce", arr, len(chunk), _S(), keep=arr)


def _backward(pl, head, d_logits, d_hidden, training):
    st = pl.st
    B = pl.key[0]
    mods = st["mods"]
    fin = st["fin"]
    dl = None if d_logits is None else d_logits.contiguous().float()
    dh = None if d_hidden is None else d_hidden.contiguous().float()
    fin.d_logits, fin.d_hidden = _ptr(dl), _ptr(dh)
    deS, deT = torch.empty_like(pl.de[0]), torch.empty_like(pl.de[1])
    sd = pl.seeds_used
    if pl.bwd_calls is not None and pl.sig == _signature(head, training):
        a = pl.dropb_entry[1]
        a[2], a[4], a[7], a[9] = deS.data_ptr(), sd["in_S"], deT.data_ptr(), sd["in_T"]
        pl.live_b = _storages(dl, dh, deS, deT)
        ops.replay(pl.bwd_calls)
        return deS, deT
    record = ops._rec() is None and pl.fwd_calls is not None
    if record:
        ops.start_recording()
    ops._call("scl_graph_final_bwd", ctypes.byref(fin), B, _S(), keep=fin)
    _post(st, ("12", "22"), B, False)
    _score_bwd(pl, "12", mods["12"], B)
    _score_bwd(pl, "22", mods["22"], B)
    ops._call("scl_graph_pre_bwd", st["pre"]["2"], 0, B, _S(), keep=st["pre"]["2"])
    _post(st, ("11", "21"), B, False)
    _score_bwd(pl, "11", mods["11"], B)
    _score_bwd(pl, "21", mods["21"], B)
    ops._call("scl_graph_pre_bwd", st["pre"]["1"], 1, B, _S(), keep=st["pre"]["1"])
    _post(st, ("S", "T"), B, False)
    _score_bwd(pl, "S", mods["S"], B)
    _score_bwd(pl, "T", mods["T"], B)
    lS, lT = pl.layers["S"], pl.layers["T"]
    pl.dropb_entry = ops._call("scl_graph_drop_bwd", lS["dxd_a"].data_ptr(), lS["dxd_b"].data_ptr(), deS.data_ptr(), deS.numel(), sd["in_S"],
                               lT["dxd_a"].data_ptr(), lT["dxd_b"].data_ptr(), deT.data_ptr(), deT.numel(), sd["in_T"], pl.p_in0, _S())
    _reduce(pl, head, B)
    pl.live_b = _storages(dl, dh, deS, deT)
    if record:
        pl.bwd_calls = ops.stop_recording()
    return deS, deT


class _GraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e_S, e_T, head, training, hold, *params):
        pl = _acquire(head, e_S.shape[0], e_S.shape[1], e_T.shape[1], e_S.device, hold)
        logits, hidden = _forward(pl, head, e_S.contiguous().float(), e_T.contiguous().float(), training)
        _hold(ctx, pl)
        ctx.head, ctx.training = head, training
        return logits, hidden

    @staticmethod
    def backward(ctx, d_logits, d_hidden):
        pl = _owned(ctx)
        try:
            deS, deT = _backward(pl, ctx.head, d_logits, d_hidden, ctx.training)
        finally:
            pl.busy = False
        return (deS, deT, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


_NAMES = ("GAT_layer_S", "GAT_layer_T", "HtrgGAT_layer_ST11", "HtrgGAT_layer_ST12", "HtrgGAT_layer_ST21", "HtrgGAT_layer_ST22", "pool_S", "pool_T",
          "pool_hS1", "pool_hT1", "pool_hS2", "pool_hT2", "out_layer")


def graph_module(e_S, e_T, head):
    training = bool(head.GAT_layer_S.training)
    params = [p for n in _NAMES for p in getattr(head, n).parameters()] + [head.master1, head.master2]
    hold = torch.is_grad_enabled() and (e_S.requires_grad or e_T.requires_grad or any(p.requires_grad for p in params))
    return _GraphFn.apply(e_S, e_T, head, training, hold, *params)

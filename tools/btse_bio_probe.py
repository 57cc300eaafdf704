"""The fused bio transformer of the wav2vec2_btse plugin (csrc/btse.hip): kernel time of one forward / backward launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scl_amd import ops
from scl_amd.btse_head import BtseHead, DEFAULT_BTSE, _plan
dev = torch.device("cuda:0")
for B, Lt in ((64, 199), (128, 199), (64, 64), (64, 512)):
    head = BtseHead(DEFAULT_BTSE).to(dev)
    pl = _plan(head, B, 8, Lt, dev)
    pl["bio"].copy_(torch.randint(0, 3, (B, Lt), dtype=torch.int32))
    pl["lens"].fill_(Lt)
    pl["ds"].normal_()
    def t(fn, n=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    f = t(lambda: ops.btse_bio_fwd(pl["desc"])); bw = t(lambda: ops.btse_bio_bwd(pl["desc"]))
    print("B %3d tokens %3d: forward %8.1f us  backward %8.1f us" % (B, Lt, f, bw))

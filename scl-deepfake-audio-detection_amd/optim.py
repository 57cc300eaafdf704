"""AdamW over the model's flat parameter buffer (one fused HIP kernel per step).

Mirror of `torch.optim.AdamW(model.parameters(), lr=args.max_lr, weight_decay=args.weight_decay)`
(main.py:339): same update rule and defaults; parameters without gradients (fairseq's pre-training
heads, the unused BatchNorms) are left untouched, as torch does for `.grad is None`.  It is a real
torch.optim.Optimizer so that `torch.optim.lr_scheduler.CyclicLR(..., cycle_momentum=False)`
(main.py:341) drives `param_groups[0]['lr']` unchanged.
"""
import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_sync=None):
        self.model = model
        self.P = model.P
        params = [p for p in self.P.params.values() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        n = self.P.n_train
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.P.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.P.device)
        self.step_count = 0
        self.grad_sync = grad_sync
        model.grad_sync = grad_sync

    def zero_grad(self, set_to_none=True):
        # gradients live in one flat buffer that every backward overwrites in full; nothing to clear,
        # and the .grad views must stay attached.  A model with torch-composed parts (wav2vec2_aasist's head) clears the
        # slice autograd accumulates into.
        if hasattr(self.model, "zero_torch_grads"):
            self.model.zero_torch_grads()
        return None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        scale = 1.0
        if self.grad_sync is not None:
            scale = self.grad_sync.finish()
        self.step_count += 1
        ops.adamw_flat(self.P.flat, self.P.grad, self.exp_avg, self.exp_avg_sq, self.P.bf16, self.P.n_train, float(g["lr"]),
                       g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count, scale)
        self.model.optimizer_stepped(bf16_fresh=True)
        return None

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v

"""GumbelVectorQuantizer (fairseq/modules/gumbel_vector_quantizer.py; used at speecht5.py:95-107,858-882).

The d -> groups*num_vars projection runs on the HIP GEMM.  The Gumbel-softmax sampling, hard one-hot,
perplexities and the code-book product ([B*T, 200] x [200, 384]) are fp32 torch ops: they are
RNG-dependent bookkeeping on ~0.1 % of the step and are listed in DESIGN.md as not yet
ported to HIP."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn


class GumbelVectorQuantizer(nn.Module):
    def __init__(self, dim, num_vars, temp, groups, combine_groups, vq_dim, time_first, activation=nn.GELU(),
                 weight_proj_depth=1, weight_proj_factor=1):
        super().__init__()
        assert weight_proj_depth == 1 and time_first and not combine_groups
        self.groups = groups
        self.combine_groups = combine_groups
        self.input_dim = dim
        self.num_vars = num_vars
        self.time_first = time_first
        assert vq_dim % groups == 0
        self.vars = nn.Parameter(torch.FloatTensor(1, groups * num_vars, vq_dim // groups))
        nn.init.uniform_(self.vars)
        self.weight_proj = nn.Linear(dim, groups * num_vars)
        nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
        nn.init.zeros_(self.weight_proj.bias)
        if isinstance(temp, str):
            import ast
            temp = ast.literal_eval(temp)
        self.max_temp, self.min_temp, self.temp_decay = temp
        self.curr_temp = self.max_temp

    def set_num_updates(self, num_updates):
        self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    def sample(self, logits):
        """Training-time hard Gumbel-softmax (straight-through); overridable for parity tests."""
        tau = self.curr_temp
        if Fn.static_shapes():   # the temperature decays with the update count: a replayed step reads it from device memory
            tau = Fn.stage_host(lambda: torch.tensor([float(self.curr_temp)], dtype=torch.float32), logits.device)
        return F.gumbel_softmax(logits.float(), tau=tau, hard=True).type_as(logits)

    def forward(self, x, produce_targets=False):
        result = {"num_vars": self.num_vars * self.groups}
        bsz, tsz, fsz = x.shape
        logits = Fn.as_float(Fn.linear(Fn.as_compute(x).reshape(-1, fsz), self.weight_proj.weight, self.weight_proj.bias))
        logits = logits.reshape(bsz * tsz * self.groups, -1)
        _, k = logits.max(-1)
        hard_x = logits.new_zeros(*logits.shape).scatter_(-1, k.view(-1, 1), 1.0).view(bsz * tsz, self.groups, -1)
        hard_probs = torch.mean(hard_x.float(), dim=0)
        result["code_perplexity"] = torch.exp(-torch.sum(hard_probs * torch.log(hard_probs + 1e-7), dim=-1)).sum()
        avg_probs = torch.softmax(logits.view(bsz * tsz, self.groups, -1).float(), dim=-1).mean(dim=0)
        result["prob_perplexity"] = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-7), dim=-1)).sum()
        result["temp"] = self.curr_temp
        sel = self.sample(logits) if self.training else hard_x
        sel = sel.view(bsz * tsz, -1)
        if produce_targets:
            result["targets"] = sel.view(bsz * tsz * self.groups, -1).argmax(dim=-1).view(bsz, tsz, self.groups).detach()
        # sum_v onehot[b,g,v] * vars[g,v,:]  ==  per-group matmul; the reference materialises the [B*T, G*V, D/G]
        # product (1.2 GB at cfg 2) before summing -- same values, 200x less traffic
        q = torch.einsum("bgv,gvd->bgd", sel.view(bsz * tsz, self.groups, self.num_vars),
                         self.vars.view(self.groups, self.num_vars, -1))
        result["x"] = q.reshape(bsz, tsz, -1)
        return result

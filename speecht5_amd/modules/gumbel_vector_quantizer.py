"""GumbelVectorQuantizer (fairseq/modules/gumbel_vector_quantizer.py; used at speecht5.py:95-107,858-882).

The d -> groups*num_vars projection runs on the HIP GEMM; sampling, hard one-hot selection, code-book lookup, both
perplexities and (optionally) the time-wise mix with the encoder states are csrc/vq.hip (functional.GumbelVQFunction);
only the noise draw itself stays a torch RNG call so that its stream matches F.gumbel_softmax."""
import torch
import torch.nn as nn

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

    def gumbel_noise(self, logits):
        """Gumbel(0, 1) noise for the training-time sampling, drawn exactly as F.gumbel_softmax draws it (the parity tests
        replace this hook with the noise recorded from the reference run)."""
        return -torch.empty_like(logits).exponential_().log()

    def forward(self, x, produce_targets=False, mix_w=None):
        """x [B, T, C].  mix_w (optional, [T] fp32 of 0/1): result["x"] is then the time-wise mix w * codes + (1 - w) * x of
        speecht5.py:870-877 instead of the codes alone (same kernel, csrc/vq.hip)."""
        result = {"num_vars": self.num_vars * self.groups}
        bsz, tsz, fsz = x.shape
        G, V = self.groups, self.num_vars
        xc = Fn.as_compute(x).reshape(-1, fsz)
        logits = Fn.as_float(Fn.linear(xc, self.weight_proj.weight, self.weight_proj.bias))      # [B*T, G*V] fp32
        tau = self.curr_temp
        if Fn.static_shapes():   # the temperature decays with the update count: a replayed step reads it from device memory
            tau = Fn.stage_host(lambda: torch.tensor([float(self.curr_temp)], dtype=torch.float32), logits.device)
        gum = self.gumbel_noise(logits).view_as(logits) if self.training else None
        vars_c = Fn.as_compute(self.vars).detach().view(G * V, -1)
        out, code_pp, prob_pp, rows = Fn.GumbelVQFunction.apply(logits, self.vars, vars_c, xc if mix_w is not None else None, gum, mix_w,
                                                                tau, self.training, G, V, tsz, xc.dtype)
        result["code_perplexity"] = code_pp
        result["prob_perplexity"] = prob_pp
        result["temp"] = self.curr_temp
        if produce_targets:
            result["targets"] = (rows.long() - torch.arange(G, device=rows.device) * V).view(bsz, tsz, G)
        result["x"] = out.view(bsz, tsz, -1)
        return result

"""SpeechEncoderPostnet (HuBERT NCE head) mirror of
SpeechT5/speecht5/models/modules/speech_encoder_postnet.py:17-124."""
import torch
import torch.nn as nn

from .. import functional as Fn


class SpeechEncoderPostnet(nn.Module):
    def __init__(self, dictionaries, args):
        super().__init__()
        assert not args.target_glu
        self.target_glu = None
        self.skip_masked = args.skip_masked
        self.skip_nomask = args.skip_nomask
        self.logit_temp = args.logit_temp
        final_dim = args.final_dim if args.final_dim > 0 else args.encoder_embed_dim
        if not any(d is None for d in dictionaries):
            self.num_classes = [len(d) for d in dictionaries]
            self.label_embs_concat = nn.Parameter(torch.FloatTensor(sum(self.num_classes), final_dim))
            nn.init.uniform_(self.label_embs_concat)
        self.untie_final_proj = args.untie_final_proj
        self.final_proj = nn.Linear(args.encoder_embed_dim, final_dim * len(dictionaries) if self.untie_final_proj else final_dim)

    def compute_nce(self, x, i, target):
        """logits[s] = [cos(x_s, e_{t_s}), cos(x_s, e_0), ..., cos(x_s, e_{V-1})] / temp with -inf where e_c == e_{t_s}, for
        target list i (code-book rows [row0, row0 + V) of label_embs_concat): functional.NCELogitsFunction -- row
        normalisations, the [S, V] cosine GEMM and the logit assembly as HIP kernels, fp32 like the reference."""
        row0 = sum(self.num_classes[:i])
        return Fn.NCELogitsFunction.apply(x, self.label_embs_concat, row0, self.num_classes[i], target, self.logit_temp)

    def forward(self, x, padding_mask, mask_indices, target_list):
        """x [B,T,C] (compute dtype).  Boolean-index gathers are torch glue, the projection is the HIP GEMM."""
        label_embs_list = self.label_embs_concat.split(self.num_classes, 0)
        if Fn.static_shapes():
            return self._forward_static(x, padding_mask, mask_indices, target_list, label_embs_list)

        host = getattr(mask_indices, "_st5_host", None)

        def branch(sel, sel_host):
            if sel_host is not None:  # gather indices known on the host: no nonzero() sync on the device
                idx = torch.nonzero(sel_host.reshape(-1)).squeeze(1).to(x.device, non_blocking=True)
                rows = x.reshape(-1, x.shape[-1]).index_select(0, idx)
                sel = idx
            else:
                rows = x[sel]  # [S, C]
            proj = Fn.linear(rows.contiguous(), self.final_proj.weight, self.final_proj.bias)
            projs = proj.chunk(len(target_list), dim=-1) if self.untie_final_proj else [proj] * len(target_list)
            outs = []
            for i, (p, t) in enumerate(zip(projs, target_list)):
                tg = t.reshape(-1).index_select(0, sel) if sel_host is not None else t[sel]
                emb = label_embs_list[i]
                outs.append(self.compute_nce(p.contiguous(), i, tg))
            return outs

        mh = ph = None
        if host is not None:
            mh, ph = host
        if not self.skip_masked:
            logit_m_list = branch(torch.logical_and(~padding_mask, mask_indices), (~ph & mh) if host is not None else None)
        else:
            logit_m_list = [None for _ in target_list]
        if not self.skip_nomask:
            logit_u_list = branch(torch.logical_and(~padding_mask, ~mask_indices), (~ph & ~mh) if host is not None else None)
        else:
            logit_u_list = [None for _ in target_list]
        return {"logit_m_list": logit_m_list, "logit_u_list": logit_u_list, "padding_mask": padding_mask}

    def _forward_static(self, x, padding_mask, mask_indices, target_list, label_embs_list):
        """Fixed-shape form for recorded / replayed steps: the boolean-index gathers above have data-dependent sizes (the span
        mask is redrawn every step), which a captured graph cannot hold.  Here EVERY frame is projected and scored; `sel_m` /
        `sel_u` say which frames each loss runs over (the criterion turns them into ignored targets and into the sample
        size).  Same per-frame logits, same sums."""
        R = x.shape[0] * x.shape[1]
        proj = Fn.linear(x.reshape(R, x.shape[-1]).contiguous(), self.final_proj.weight, self.final_proj.bias)
        projs = proj.chunk(len(target_list), dim=-1) if self.untie_final_proj else [proj] * len(target_list)
        logits = []
        for i, (p, t) in enumerate(zip(projs, target_list)):
            tg = t.reshape(-1)
            emb = label_embs_list[i]
            logits.append(self.compute_nce(p.contiguous(), i, tg))
        valid = ~padding_mask
        out = {"padding_mask": padding_mask,
               "logit_m_list": [None] * len(target_list) if self.skip_masked else logits,
               "logit_u_list": [None] * len(target_list) if self.skip_nomask else logits,
               "sel_m": torch.logical_and(valid, mask_indices).reshape(-1), "sel_u": torch.logical_and(valid, ~mask_indices).reshape(-1)}
        return out

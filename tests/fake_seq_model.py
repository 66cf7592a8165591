"""A stand-in encoder-decoder with the surface the beam-search generator talks to (forward_encoder_torchscript, decoder.forward,
get_normalized_probs, encoder.reorder_encoder_out, max_decoder_positions; the two pre-net attributes mark it as a "T5 structure"),
in plain torch on the CPU.  TEST INFRASTRUCTURE: lets the generator's HOST logic (beam bookkeeping, length / unk penalties,
prefix forcing, n-gram blocking, finalisation order) be compared with the verbatim reference generator without the model
kernels; oracle/make_golden_beam_fake.py runs the reference on it, tests/test_generator_cpu.py the product."""
import torch
import torch.nn as nn


class FakeEncoder(nn.Module):
    def reorder_encoder_out(self, encoder_out, new_order):
        return {"encoder_out": [encoder_out["encoder_out"][0].index_select(1, new_order)],
                "encoder_padding_mask": [encoder_out["encoder_padding_mask"][0].index_select(0, new_order)],
                "encoder_states": [], "src_tokens": [], "decoder_input": [None]}


class FakeDecoder(nn.Module):
    """Stateless: the logits of position t depend on the whole prefix (running mean of the token embeddings) and on the
    sentence's encoder summary, so beams of one sentence diverge and sentences differ."""

    def __init__(self, vocab, dim):
        super().__init__()
        self.emb = nn.Embedding(vocab, dim)
        self.mix = nn.Linear(dim, dim)
        self.out = nn.Linear(dim, vocab)

    def forward(self, tokens, encoder_out=None, incremental_state=None):
        enc = encoder_out["encoder_out"][0]                      # T x B x C
        keep = (~encoder_out["encoder_padding_mask"][0]).to(enc.dtype).t().unsqueeze(-1)   # T x B x 1
        ctx = (enc * keep).sum(0) / keep.sum(0).clamp(min=1.0)     # B x C
        e = self.emb(tokens)                                       # B x t x C
        run = e.cumsum(1) / torch.arange(1, tokens.size(1) + 1, dtype=e.dtype).view(1, -1, 1)
        h = torch.tanh(self.mix(run) + ctx.unsqueeze(1) + 0.5 * e)
        return self.out(h) * 3.0, {"attn": [None], "inner_states": None}


class FakeSeqModel(nn.Module):
    def __init__(self, vocab, dim=16, hop=80, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.hop = hop
        self.text_encoder_prenet = nn.Identity()      # (is_t5_structure: both attributes present)
        self.speech_encoder_prenet = nn.Identity()
        self.encoder = FakeEncoder()
        self.decoder = FakeDecoder(vocab, dim)
        self.proj = nn.Linear(hop, dim)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 if p.dim() > 1 else 0.1))

    def max_decoder_positions(self):
        return 1024

    def forward_encoder_torchscript(self, net_input):
        src, pad = net_input["source"], net_input["padding_mask"]
        B, S = src.shape
        T = S // self.hop
        frames = src[:, : T * self.hop].view(B, T, self.hop)
        fpad = pad[:, : T * self.hop].view(B, T, self.hop).all(-1)
        return {"encoder_out": [torch.tanh(self.proj(frames)).transpose(0, 1)], "encoder_padding_mask": [fpad],
                "encoder_states": [], "src_tokens": [], "decoder_input": [None]}

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        x = net_output[0].float()
        return torch.log_softmax(x, -1) if log_probs else torch.softmax(x, -1)


def fake_sample(B=3, S=1600, seed=5):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, S, generator=g)
    pad = torch.zeros(B, S, dtype=torch.bool)
    for b in range(1, B):                      # ragged lengths
        pad[b, S - 240 * b:] = True
    return {"net_input": {"source": src, "padding_mask": pad}, "id": torch.arange(B)}


CASES = dict(
    beam1=dict(kw=dict(beam_size=1, max_len_b=14)),
    beam4=dict(kw=dict(beam_size=4, max_len_b=14)),
    beam5_lenpen_unnorm=dict(kw=dict(beam_size=5, max_len_a=0.005, max_len_b=6, len_penalty=0.6, normalize_scores=False, unk_penalty=1.5)),
    beam3_minlen=dict(kw=dict(beam_size=3, max_len_b=12, min_len=6)),
    beam4_ngram2=dict(kw=dict(beam_size=4, max_len_b=16, no_repeat_ngram_size=2)),
    beam3_ngram3_temp=dict(kw=dict(beam_size=3, max_len_b=16, no_repeat_ngram_size=3, temperature=1.7)),
    beam3_prefix=dict(kw=dict(beam_size=3, max_len_b=12), prefix=[[7, 9, 11], [12, 2, 1], [5, 6, 1]]),
)

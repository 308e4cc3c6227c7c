"""`ip_adapter.resampler.Resampler` on the MI355X engine (reference ip_adapter/resampler.py:81-147: constructor
arguments, parameter names, forward semantics).  PerceiverAttention / FeedForward are not separate modules here: the whole
resampler is one op schedule (followyourclick_amd.engine.encoders.ResamplerEngine)."""
from __future__ import annotations

import torch
import torch.nn as nn

from followyourclick_amd.encoders import EngineBacked
from followyourclick_amd.engine import encoders as EN


def _node(parent: nn.Module, name: str) -> nn.Module:
    if name not in parent._modules:
        parent.add_module(name, nn.Module())
    return parent._modules[name]


class Resampler(EngineBacked):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024, ff_mult=4,
                 max_seq_len: int = 257, apply_pos_emb: bool = False, num_latents_mean_pooled: int = 0,
                 compute_dtype: torch.dtype = None):
        super().__init__()
        if apply_pos_emb or num_latents_mean_pooled:
            raise NotImplementedError("Resampler on the MI355X engine: apply_pos_emb / num_latents_mean_pooled are not used by "
                                      "MyIPAdapterPlus (reference my_ip_adapter.py:240-250) and not implemented")
        self.engine_config = EN.ResamplerConfig(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_queries=num_queries,
                                                embedding_dim=embedding_dim, output_dim=output_dim, ff_mult=ff_mult)
        inner = dim_head * heads
        g = torch.Generator().manual_seed(0)

        def lin(o, i):
            return nn.Parameter(torch.randn(o, i, generator=g) / i ** 0.5, requires_grad=False)

        def vec(n, v):
            return nn.Parameter(torch.full((n,), v), requires_grad=False)

        def norm(node, n):
            node.register_parameter("weight", vec(n, 1.0))
            node.register_parameter("bias", vec(n, 0.0))

        self.latents = nn.Parameter(torch.randn(1, num_queries, dim, generator=g) / dim ** 0.5, requires_grad=False)
        for name, (o, i) in (("proj_in", (dim, embedding_dim)), ("proj_out", (output_dim, dim))):
            n = _node(self, name)
            n.register_parameter("weight", lin(o, i))
            n.register_parameter("bias", vec(o, 0.0))
        norm(_node(self, "norm_out"), output_dim)
        layers = _node(self, "layers")
        for li in range(depth):
            layer = _node(layers, str(li))
            att, ff = _node(layer, "0"), _node(layer, "1")
            norm(_node(att, "norm1"), dim)
            norm(_node(att, "norm2"), dim)
            _node(att, "to_q").register_parameter("weight", lin(inner, dim))
            _node(att, "to_kv").register_parameter("weight", lin(2 * inner, dim))
            _node(att, "to_out").register_parameter("weight", lin(dim, inner))
            norm(_node(ff, "0"), dim)
            _node(ff, "1").register_parameter("weight", lin(dim * ff_mult, dim))
            _node(ff, "3").register_parameter("weight", lin(dim, dim * ff_mult))
        self._init_engine_state(compute_dtype)

    def _build_engine(self, sd, device):
        return EN.ResamplerEngine(EN.pack_resampler(sd, self.engine_config, self.compute_dtype, device))

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._get_engine().resample(x).to(x.dtype if x.dtype.is_floating_point else torch.float32)

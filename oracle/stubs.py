"""Test doubles for the conditioning front-end (tokenizer + CLIP text encoder).

TEST INFRASTRUCTURE.  No tokenizer vocabulary or CLIP checkpoint exists offline, and the
text encoder sits *before* the hot path: it is treated as an opaque producer of
(B, 77, D) states.  Both the real reference pipeline (in ``make_golden.py``) and the HIP engine's
drop-in pipeline (in tests) are driven with these same doubles.
"""
import types
import zlib

import torch


class FakeTokenizer:
    """Maps a prompt to deterministic ids (crc32 per word); mirrors the subset of the CLIPTokenizer
    call surface that AnimationPipeline._encode_prompt uses (pipeline_animation.py:161-175, 216-222)."""
    model_max_length = 77

    def __init__(self, vocab_size: int = 1000):
        self.vocab_size = vocab_size

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        L = self.model_max_length
        ids = torch.zeros(len(prompts), L, dtype=torch.int64)
        for i, p in enumerate(prompts):
            toks = [1] + [2 + zlib.crc32(w.encode()) % (self.vocab_size - 3) for w in p.split()][: L - 2] + [self.vocab_size - 1]
            ids[i, : len(toks)] = torch.tensor(toks)
        return types.SimpleNamespace(input_ids=ids, attention_mask=(ids != 0).long())

    def batch_decode(self, ids):
        return ["<ids>"] * len(ids)


class StubTextEncoder(torch.nn.Module):
    """ids -> (B, 77, dim) states: seeded embedding table + position table."""

    def __init__(self, dim: int, vocab_size: int = 1000, seed: int = 3000):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.tok = torch.nn.Parameter(torch.randn(vocab_size, dim, generator=g), requires_grad=False)
        self.pos = torch.nn.Parameter(0.5 * torch.randn(77, dim, generator=g), requires_grad=False)
        self.config = types.SimpleNamespace(hidden_size=dim)

    @property
    def dtype(self):
        return self.tok.dtype

    @property
    def device(self):
        return self.tok.device

    def forward(self, input_ids, attention_mask=None):
        return (self.tok[input_ids] + self.pos[None, : input_ids.shape[1]],)

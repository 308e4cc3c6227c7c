"""Torch-CPU emulation of the C ABI ops (TEST INFRASTRUCTURE, never imported by the product).

Two uses:
  * `-m "not gpu"` tests swap this in for `followyourclick_amd.ops.impl` to check the *host
    orchestration* (layouts, weight packing, residual wiring, scheduler tables) of the engine against
    the oracle without a GPU;
  * `-m gpu` tests use it as the per-op specification the HIP kernels are compared with.
It mirrors the buffer layouts of include/fyc.h exactly (flat buffers + leading dimensions).
"""

import torch
import torch.nn.functional as F

PLAIN, CONV, CONV_UP2 = 0, 1, 2
LINEAR, GEGLU, HEADS = 0, 1, 2


def _flat(t):
    return t.reshape(-1)


def _strided(t, size, stride, off=0):
    """as_strided relative to the tensor's own first element (t may be a view into a bigger buffer)"""
    return torch.as_strided(t, size, stride, t.storage_offset() + off)


class EmuOps:
    name = "emu"

    def __init__(self, acc=torch.float32):
        self.acc = acc

    def ensure_init(self, device):
        pass

    def set_tuning(self, key, value):
        pass

    # ------------------------------------------------------------------------------------
    def gemm(self, a, w, out, *, M, N, K, lda, ldw, ldo=0, bias=None, rowbias=None, rows_per_batch=1, residual=None,
             ldr=0, ldrb=0, out_scale=1.0, epilogue=LINEAR, mode=PLAIN, conv=None, batch=1, stride_a=0, stride_w=0, stride_o=0,
             heads=None, tile=0, a2=None, k_split=0, lda2=0, act=0, ln_stats=None, ln_colsum=None, ln_nparts=0, ln_eps=1e-5,
             chan_parts=None, cs_rows=0, row_parts=None, row_nparts=0, workspace=None):
        acc_t = self.acc
        if chan_parts is not None or row_parts is not None:
            assert epilogue == LINEAR and batch == 1
        if chan_parts is not None:
            assert cs_rows % 16 == 0 and M % cs_rows == 0, (cs_rows, M)
        if row_parts is not None:
            assert row_nparts == self.gemm_row_parts(a.dtype, M=M, N=N, K=K, mode=mode)
        af, wf = _flat(a), _flat(w)
        for z in range(batch):
            W = _strided(wf, (N, K), (ldw, 1), z * stride_w).to(acc_t)
            if mode == PLAIN:
                if a2 is None:
                    A = _strided(af, (M, K), (lda, 1), z * stride_a).to(acc_t)
                else:  # dual-source A = [a | a2]
                    A = torch.cat([_strided(af, (M, k_split), (lda, 1), 0).to(acc_t),
                                   _strided(_flat(a2), (M, K - k_split), (lda2, 1), 0).to(acc_t)], dim=1)
                acc = A @ W.t()
            else:
                Hin, Win, Cin, Hout, Wout = conv["Hin"], conv["Win"], conv["Cin"], conv["Hout"], conv["Wout"]
                frames = M // (Hout * Wout)
                x = af[: frames * Hin * Win * Cin].reshape(frames, Hin, Win, Cin).permute(0, 3, 1, 2).to(acc_t)
                if mode == CONV_UP2:
                    x = F.interpolate(x.float(), size=(Hout, Wout), mode="nearest").to(acc_t)
                    stride = 1
                else:
                    stride = conv.get("stride", 1)
                sl = 128 // a.element_size()                      # channels per 128-byte K slab
                w4 = W.reshape(N, Cin // sl, 3, 3, sl).permute(0, 1, 4, 2, 3).reshape(N, Cin, 3, 3)
                if mode == CONV and conv.get("pad", 1) == 0:
                    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w4, None, stride=stride, padding=0)
                else:
                    y = F.conv2d(x, w4, None, stride=stride, padding=1)
                assert y.shape[2] == Hout and y.shape[3] == Wout, (y.shape, Hout, Wout)
                acc = y.permute(0, 2, 3, 1).reshape(M, N)
            if ln_stats is not None:      # folded LayerNorm: rstd * (x W'^T - mean * colsum)
                if ln_nparts > 0:         # partial {sum, sum sq} per row from the producer's epilogue
                    pr = _flat(ln_stats)[: M * ln_nparts * 2].reshape(M, ln_nparts, 2).float().sum(dim=1)
                    mean = pr[:, 0] / K
                    rstd = torch.rsqrt((pr[:, 1] / K - mean * mean).clamp_min(0) + ln_eps)
                    st = torch.stack([mean, rstd], dim=1).to(acc_t)
                else:
                    st = ln_stats.reshape(-1, 2)[:M].to(acc_t)
                acc = st[:, 1:2] * (acc - st[:, 0:1] * ln_colsum.to(acc_t)[None, :N])
            if bias is not None:
                acc = acc + bias.to(acc_t)[None, :N]
            if rowbias is not None:
                nb_rows = (M + rows_per_batch - 1) // rows_per_batch
                rb = _strided(rowbias, (nb_rows, N), (ldrb if ldrb > 0 else N, 1)).to(acc_t)
                acc = acc + rb.repeat_interleave(rows_per_batch, dim=0)[:M]
            if epilogue == GEGLU:
                blk = acc.reshape(M, N // 32, 2, 16)
                y = (blk[:, :, 0] * F.gelu(blk[:, :, 1])).reshape(M, N // 2)
                o = _strided(_flat(out), (M, N // 2), (ldo, 1), z * stride_o)
                o.copy_(y.to(out.dtype))
            elif epilogue == LINEAR:
                if act == 1:
                    acc = F.gelu(acc)
                elif act == 2:
                    acc = acc * torch.sigmoid(1.702 * acc)
                if residual is not None:
                    acc = acc + _strided(_flat(residual), (M, N), (ldr, 1), 0).to(acc_t)
                acc = acc * out_scale
                o = _strided(_flat(out), (M, N), (ldo, 1), z * stride_o)
                o.copy_(acc.to(out.dtype))
                if chan_parts is not None or row_parts is not None:      # statistics of the values as stored
                    v = o.double()
                    if chan_parts is not None:       # [row tiles][slots][N][2], slot = sample - first sample of the tile
                        nt, tr, sl = self.gemm_stat_layout(a.dtype, M=M, N=N, K=K, cs_rows=cs_rows, mode=mode)
                        cp = _flat(chan_parts)[: nt * sl * N * 2].reshape(nt, sl, N, 2)
                        cp.zero_()
                        for t in range(nt):
                            r0, r1 = t * tr, min((t + 1) * tr, M)
                            first = r0 // cs_rows
                            for f in range(first, (r1 - 1) // cs_rows + 1):
                                blk = v[max(r0, f * cs_rows):min(r1, (f + 1) * cs_rows)]
                                cp[t, f - first, :, 0] = blk.sum(dim=0).float()
                                cp[t, f - first, :, 1] = (blk * blk).sum(dim=0).float()
                    if row_parts is not None:
                        rp = _flat(row_parts)[: M * row_nparts * 2].reshape(M, row_nparts, 2)
                        edges = [round(i * N / row_nparts) for i in range(row_nparts + 1)]
                        for i in range(row_nparts):
                            blk = v[:, edges[i]:edges[i + 1]]
                            rp[:, i, 0] = blk.sum(dim=1).float()
                            rp[:, i, 1] = (blk * blk).sum(dim=1).float()
            else:
                acc = acc * out_scale
                sc, H, T = heads["seg_cols"], heads["heads"], heads["tokens"]
                d, Bn = sc // H, M // T
                for s, (t, tr, ld) in enumerate(zip(heads["outs"], heads["transposed"], heads["ld"])):
                    seg = acc[:, s * sc:(s + 1) * sc].reshape(Bn, T, H, d)
                    if not tr:
                        _flat(t)[: Bn * H * T * d].reshape(Bn, H, T, d).copy_(seg.permute(0, 2, 1, 3).to(t.dtype))
                    else:
                        ld = ld if ld > 0 else T
                        v = _strided(_flat(t), (Bn, H, d, T), (H * d * ld, d * ld, ld, 1), 0)
                        v.copy_(seg.permute(0, 2, 3, 1).to(t.dtype))

    def gemm_split_bytes(self, dtype, *, M, N, K, mode=PLAIN):
        return 0

    def gemm_stat_layout(self, dtype, *, M, N, K, cs_rows, mode=PLAIN, batch=1, tile=0):
        """row tiles of 96 rows: neither a divisor nor a multiple of the usual sample sizes, so samples straddle tiles"""
        tr = 96
        return (M + tr - 1) // tr, tr, (tr - 1) // cs_rows + 2

    def chan_stats_reduce(self, parts, cs, *, rows, N, cs_rows, tile_rows, slots, out_rows=0):
        nt = (rows + tile_rows - 1) // tile_rows
        p = _flat(parts)[: nt * slots * N * 2].reshape(nt, slots, N, 2).double()
        out = torch.zeros(rows // cs_rows, N, 2, dtype=torch.float64)
        for t in range(nt):
            first = (t * tile_rows) // cs_rows
            for sl in range(slots):
                if first + sl < out.shape[0]:
                    out[first + sl] += p[t, sl]
        grp = (out_rows or cs_rows) // cs_rows
        out = out.reshape(out.shape[0] // grp, grp, N, 2).sum(dim=1)
        _flat(cs)[: out.numel()].reshape(out.shape).copy_(out)

    def gemm_row_parts(self, dtype, *, M, N, K, mode=PLAIN, batch=1, tile=0):
        """any partition of the columns is valid for the consumer; use several parts so that their summation is exercised"""
        return 3 if N % 3 == 0 else (2 if N % 2 == 0 else 1)

    # ------------------------------------------------------------------------------------
    def attention(self, q, k, vt, o, *, batch, heads, n_q, n_k, d, ldo, ldvt, scale, kv_batch_div=1, accumulate=False,
                  o_scale=1.0):
        acc_t = self.acc
        Q = _flat(q)[: batch * heads * n_q * d].reshape(batch, heads, n_q, d).to(acc_t)
        kvB = (batch + kv_batch_div - 1) // kv_batch_div
        Kt = _flat(k)[: kvB * heads * n_k * d].reshape(kvB, heads, n_k, d).to(acc_t)
        Vt = _strided(_flat(vt), (kvB, heads, d, n_k), (heads * d * ldvt, d * ldvt, ldvt, 1), 0).to(acc_t)
        idx = torch.arange(batch) // kv_batch_div
        S = torch.matmul(Q, Kt[idx].transpose(-1, -2)) * scale
        P = S.softmax(dim=-1)
        if q.dtype != torch.float32:
            P = P.to(q.dtype).to(acc_t)  # the kernel feeds bf16 / f16 probabilities to the MFMA
        O = torch.matmul(P, Vt[idx].transpose(-1, -2))           # b h n d
        O = O.permute(0, 2, 1, 3).reshape(batch * n_q, heads * d)
        dst = _strided(_flat(o), (batch * n_q, heads * d), (ldo, 1), 0)
        if accumulate:
            O = dst.to(acc_t) + o_scale * O
        dst.copy_(O.to(o.dtype))

    def temporal_attention(self, qkv, o, *, clips, frames, pixels, heads, d, scale):
        acc_t = self.acc
        Cc = heads * d
        x = _flat(qkv)[: clips * frames * pixels * 3 * Cc].reshape(clips, frames, pixels, 3, heads, d).to(acc_t)
        q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]          # b f p h d
        q, k, v = (t.permute(0, 2, 3, 1, 4) for t in (q, k, v))          # b p h f d
        P = (torch.matmul(q, k.transpose(-1, -2)) * scale).softmax(dim=-1)
        if qkv.dtype != torch.float32:
            P = P.to(qkv.dtype).to(acc_t)
        O = torch.matmul(P, v).permute(0, 3, 1, 2, 4).reshape(clips * frames * pixels, Cc)
        _flat(o)[: O.numel()].reshape(O.shape).copy_(O.to(o.dtype))

    def temporal_block_supported(self, dtype, *, clips, frames, pixels, heads, d):
        """the shapes libfyc_hip.so's kernel is built for (csrc/temporal_block.hip); `temporal_block` itself is a
        specification for any head layout that fits the operand packing (3 d <= 128, d <= 48)"""
        return dtype in (torch.bfloat16, torch.float16) and heads == 8 and d == 40 and frames == 16 and pixels % 8 == 0

    @staticmethod
    def _tblock_unpack(wstream, C_, heads, d):
        """inverse of the fyc_temporal_block weight stream (include/fyc.h), written from the layout description: returns
        (w [H][3][48][C] q | k | v rows, table [H][16][3][48] f32 bias + positional term per frame, wo [H][C][48])"""
        ks, nb = C_ // 32, C_ // 16
        a_tab, a_n = 6 * ks, 6 * ks + 6
        b_wo, b_tab, b_n = 3 * ks, 3 * ks + 2 * nb, 3 * ks + 2 * nb + 3
        S = _flat(wstream).reshape(2 * heads, max(a_n, b_n), 64, 8)                 # [stage][piece][lane][e]
        lane = torch.arange(64)
        r, gq = lane & 15, lane >> 4

        def block(piece):                                                            # [64][8] fragment -> [16][32] weight block
            blk = torch.zeros(16, 32, dtype=piece.dtype)
            for e in range(8):
                blk[r, 8 * gq + e] = piece[:, e]
            return blk
        w = torch.zeros(heads, 3, 48, C_, dtype=wstream.dtype)
        tab = torch.zeros(heads, 16, 3, 48)
        wo = torch.zeros(heads, C_, 48, dtype=wstream.dtype)
        for h in range(heads):
            A, B = S[2 * h], S[2 * h + 1]
            for s_ in range(ks):
                for b in range(6):
                    w[h, b // 3, 16 * (b % 3): 16 * (b % 3) + 16, 32 * s_: 32 * s_ + 32] = block(A[6 * s_ + b])
                for b in range(3):
                    w[h, 2, 16 * b: 16 * b + 16, 32 * s_: 32 * s_ + 32] = block(B[3 * s_ + b])
            ta = A[a_tab: a_tab + 6].reshape(-1).view(torch.float32).reshape(16, 96)
            tab[h, :, 0], tab[h, :, 1] = ta[:, :48], ta[:, 48:]
            tab[h, :, 2] = B[b_tab: b_tab + 3].reshape(-1).view(torch.float32).reshape(48, 16).t()
            for t in range(2):
                for j in range(nb):
                    blk = block(B[b_wo + t * nb + j])                                # columns = k-slots 8 g + e of k-step t
                    for k in range(32):
                        g_, e = k // 8, k % 8
                        if t == 0:
                            f = 4 * g_ + e if e < 4 else 16 + 4 * g_ + e - 4
                        else:
                            f = 32 + 4 * g_ + e if e < 4 else None
                        if f is None:
                            assert not blk[:, k].float().any(), "unused k-slots of the Wo' fragments must be zero"
                        else:
                            wo[h, 16 * j: 16 * j + 16, f] = blk[:, k]
        return w, tab, wo

    def temporal_block(self, x, out, *, w_qkv, colsum, bias, pe_bias, w_out, b_out, clips, frames, pixels, heads, d, scale, eps=1e-5,
                       wstream=None):
        """out = x + Attn_F(LayerNorm(x) + pe) Wo^T + bo from the per-head operands of engine/weights.py::pack_temporal_block
        (LayerNorm folded: rstd (x W'^T - mean colsum) + bias); q|k|v, the probabilities and the attention output are rounded to
        the storage dtype where the kernel stores them.  With `wstream` (the register-resident kernel's packed weights) the
        operands are taken from the stream and the projections see the normalised tokens (x - mean) rstd rounded to the storage
        dtype, as that kernel computes them"""
        acc_t, T = self.acc, x.dtype
        Cc = heads * d
        X = _flat(x)[: clips * frames * pixels * Cc].reshape(clips, frames, pixels, Cc).to(acc_t)
        if wstream is not None:
            key = ("tblock", wstream.data_ptr(), Cc, heads, d)
            cache = self.__dict__.setdefault("_ff_cache", {})
            if key not in cache:
                cache[key] = (wstream, self._tblock_unpack(wstream, Cc, heads, d))      # (holds the stream: see ff_block)
            w, tab, wo = cache[key][1]
            Xf = X.float()
            xn = ((Xf - Xf.mean(-1, keepdim=True)) * (Xf.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()).to(T).to(acc_t)
            qkv = torch.einsum("bfpc,hsnc->bfphsn", xn, w.to(acc_t)) + tab.to(acc_t).permute(1, 0, 2, 3)[None, :, None]   # b f p h 3 48
            qkv = qkv.to(T).to(acc_t)
            q, k, v = qkv[..., 0, :d], qkv[..., 1, :d], qkv[..., 2, :d]
            S = torch.einsum("bfphd,bgphd->bphfg", q, k) * scale
            P = S.softmax(dim=-1).to(T).to(acc_t)
            O = torch.einsum("bphfg,bgphd->bfphd", P, v).to(T).to(acc_t)
            y = X + torch.einsum("bfphk,hnk->bfpn", O, wo.to(acc_t)[:, :, :d]) + b_out.to(acc_t)
            _flat(out)[: y.numel()].reshape(y.shape).copy_(y.to(out.dtype))
            return
        mean = X.mean(-1, keepdim=True)
        rstd = (X.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
        a = torch.einsum("bfpc,hnc->bfphn", X, w_qkv.to(acc_t))
        qkv = rstd[..., None] * (a - mean[..., None] * colsum.to(acc_t)) + bias.to(acc_t)
        if pe_bias is not None:
            qkv = qkv + pe_bias.to(acc_t)[None, :, None]
        qkv = qkv.to(T).to(acc_t)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:3 * d]                  # b f p h d
        S = torch.einsum("bfphd,bgphd->bphfg", q, k) * scale
        P = S.softmax(dim=-1).to(T).to(acc_t)
        O = torch.einsum("bphfg,bgphd->bfphd", P, v).to(T).to(acc_t)
        y = X + torch.einsum("bfphk,hnk->bfpn", O, w_out.to(acc_t)[:, :, :d]) + b_out.to(acc_t)
        _flat(out)[: y.numel()].reshape(y.shape).copy_(y.to(out.dtype))

    def ff_block_supported(self, dtype, *, rows, C_, hidden, cs_rows=0):
        """the shape libfyc_hip.so's kernel is built for (csrc/ff_block.hip); `ff_block` itself is a specification for any
        C, hidden that are multiples of 32"""
        return dtype in (torch.bfloat16, torch.float16) and C_ == 320 and hidden == 1280 and rows % 128 == 0 and (cs_rows == 0 or (cs_rows % 128 == 0 and rows % cs_rows == 0))

    @staticmethod
    def _ff_unpack(wstream, C_, hidden):
        """inverse of the fyc_ff_block weight stream (include/fyc.h), written from the layout description: returns
        (Wp [C][C], W1 [2 hidden][C] GEGLU-packed with gamma folded in, bias [2 hidden] with beta folded in, W2' [C][hidden])"""
        nb, ks, chunks = C_ // 32, C_ // 16, hidden // 32
        sa = (7 * ks + 9) // 10
        hp = (max(2 * nb, 2 * sa + 1, 2 * (ks - sa) + 2 * nb) + 3) // 4 * 4
        npj = ks // 2
        S = _flat(wstream).reshape(npj + 2 * chunks + 2, hp, 64, 8)                # [half-stage][piece][lane][e]
        lane = torch.arange(64)
        r, kh = lane % 32, lane // 32

        def block(piece):                                                          # [64][8] A operand -> [32][16] weight block
            blk = torch.zeros(32, 16, dtype=piece.dtype)
            for e in range(8):
                blk[r, 8 * kh + e] = piece[:, e]
            return blk
        Wp = torch.zeros(C_, C_, dtype=wstream.dtype)
        for s_ in range(ks):
            for j in range(nb):
                Wp[32 * j: 32 * j + 32, 16 * s_: 16 * s_ + 16] = block(S[s_ // 2, (s_ % 2) * nb + j])
        W1 = torch.zeros(2 * hidden, C_, dtype=wstream.dtype)                      # back in the GEGLU-packed order of engine/weights.py::_ff
        bi = torch.zeros(2 * hidden)
        W2 = torch.zeros(C_, hidden, dtype=wstream.dtype)
        for c in range(chunks):
            ha = npj + 2 * c
            vg = torch.zeros(2, 32, C_, dtype=wstream.dtype)                       # [value | gate][unit][C]
            for s_ in range(ks):
                for v in range(2):
                    piece = S[ha, 2 * s_ + v] if s_ < sa else S[ha + 1, 2 * (s_ - sa) + v]
                    vg[v, :, 16 * s_: 16 * s_ + 16] = block(piece)
            bvg = S[ha + 2, 2 * sa].reshape(-1).view(torch.float32)[:64].reshape(2, 32)
            for half in range(2):                                                  # 16 value rows, their 16 gate rows, ...
                W1[64 * c + 32 * half: 64 * c + 32 * half + 16] = vg[0, 16 * half: 16 * half + 16]
                W1[64 * c + 32 * half + 16: 64 * c + 32 * half + 32] = vg[1, 16 * half: 16 * half + 16]
                bi[64 * c + 32 * half: 64 * c + 32 * half + 16] = bvg[0, 16 * half: 16 * half + 16]
                bi[64 * c + 32 * half + 16: 64 * c + 32 * half + 32] = bvg[1, 16 * half: 16 * half + 16]
            for sg in range(2):
                for j in range(nb):
                    blk = block(S[ha + 3, 2 * (ks - sa) + sg * nb + j])            # columns = k-slots 8 kh + e of k-step sg
                    for k in range(16):
                        kh_, e = k // 8, k % 8
                        unit = 16 * sg + 8 * (e // 4) + 4 * kh_ + e % 4
                        W2[32 * j: 32 * j + 32, 32 * c + unit] = blk[:, k]
        return Wp, W1, bi, W2

    def ff_block(self, x, residual, out, *, wstream, b_out, rows, C_, hidden, eps=1e-5, chan_parts=None, cs_rows=0):
        """out = residual + b_out + [x | GEGLU(LN(x) W1^T + b1)] [Wp | W2']^T with gamma / beta folded into W1 / b1: the kernel
        feeds FF1 the normalised tokens (x - mean) rstd rounded to the storage dtype; the hidden activation is rounded to the
        storage dtype where the kernel packs it into MFMA operands; chan_parts [rows / 128][C][2] = per 128-row tile {sum, sum of
        squares} of the stored output"""
        acc_t, T = self.acc, x.dtype
        # unpacked weights per stream; the entry keeps the stream tensor alive: keyed by address alone, a NEW stream that the
        # allocator put where a freed one had been got the old weights (a test then failed at random, round 3)
        key = (wstream.data_ptr(), C_, hidden)
        cache = self.__dict__.setdefault("_ff_cache", {})
        if key not in cache:
            cache[key] = (wstream, self._ff_unpack(wstream, C_, hidden))
        Wp, W1, bi, W2 = cache[key][1]
        X = _flat(x)[: rows * C_].reshape(rows, C_).to(acc_t)
        Xf = X.float()
        mean = Xf.mean(-1, keepdim=True)
        rstd = (Xf.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
        xn = ((Xf - mean) * rstd).to(T).to(acc_t)
        pre = xn @ W1.to(acc_t).t() + bi.to(acc_t)
        blk = pre.reshape(rows, hidden // 16, 2, 16)
        h = (blk[:, :, 0] * F.gelu(blk[:, :, 1])).reshape(rows, hidden).to(T).to(acc_t)
        y = X @ Wp.to(acc_t).t() + h @ W2.to(acc_t).t() + b_out.to(acc_t)
        if residual is not None:
            y = y + _flat(residual)[: rows * C_].reshape(rows, C_).to(acc_t)
        y = y.to(out.dtype)
        _flat(out)[: rows * C_].reshape(rows, C_).copy_(y)
        if chan_parts is not None:
            assert cs_rows % 128 == 0 and rows % 128 == 0
            t = y.double().reshape(rows // 128, 128, C_)
            _flat(chan_parts)[: (rows // 128) * C_ * 2].reshape(rows // 128, C_, 2).copy_(torch.stack([t.sum(1), (t * t).sum(1)], dim=-1).float())

    def panel_linear_supported(self, dtype, *, rows, N, K, gn_rows_per_sample=0, gn_groups=32):
        """the shapes libfyc_hip.so's kernel is built for (csrc/panel_linear.hip)"""
        return (dtype in (torch.bfloat16, torch.float16) and rows % 128 == 0 and K in (320, 640) and N in (320, 640)
                and (gn_rows_per_sample == 0 or (gn_rows_per_sample % 128 == 0 and rows % gn_rows_per_sample == 0)))

    @staticmethod
    def _panel_unpack(wstream, N, K):
        """inverse of the fyc_panel_linear weight stream (include/fyc.h), written from the layout description"""
        pn = min(N, 320)                                                         # columns per pass (the kernel: 320)
        nb = pn // 16
        S = _flat(wstream).reshape(N // pn, K // 64, 2 * nb, 64, 8)               # [pass][stage][piece][lane][e]
        lane = torch.arange(64)
        r, gq = lane & 15, lane >> 4
        W = torch.zeros(N, K, dtype=wstream.dtype)
        for P in range(N // pn):
            for t in range(K // 64):
                for sk in range(2):
                    for j in range(nb):
                        blk = torch.zeros(16, 32, dtype=wstream.dtype)
                        for e in range(8):
                            blk[r, 8 * gq + e] = S[P, t, sk * nb + j][:, e]
                        W[pn * P + 16 * j: pn * P + 16 * j + 16, 32 * (2 * t + sk): 32 * (2 * t + sk) + 32] = blk
        return W

    def panel_linear(self, x, out, *, wstream, rows, N, K, bias=None, residual=None, gn_cs=None, gn_gamma=None, gn_beta=None,
                     gn_rows_per_sample=0, gn_stat_samples=1, gn_groups=32, gn_eps=1e-6):
        """out = [GroupNorm](x) W^T + bias (+ residual); the normalised operand is rounded to the storage dtype (what
        fyc_gn_apply_cs would have stored); GroupNorm statistics from the f64 channel sums"""
        acc_t, T = self.acc, x.dtype
        key = ("panel", wstream.data_ptr(), N, K)
        cache = self.__dict__.setdefault("_ff_cache", {})
        if key not in cache:
            cache[key] = (wstream, self._panel_unpack(wstream, N, K))      # (holds the stream: see ff_block)
        W = cache[key][1]
        X = _flat(x)[: rows * K].reshape(rows, K).to(acc_t)
        if gn_cs is not None:
            S = rows // gn_rows_per_sample
            cpg = K // gn_groups
            cs = _flat(gn_cs)[: S * gn_stat_samples * K * 2].reshape(S, gn_stat_samples, K, 2).sum(dim=1)
            st = cs.reshape(S, gn_groups, cpg, 2).sum(dim=2)
            cnt = gn_rows_per_sample * cpg
            mean = st[..., 0] / cnt
            rstd = 1.0 / torch.sqrt((st[..., 1] / cnt - mean * mean).clamp_min(0) + gn_eps)
            scale = rstd.float()[:, :, None] * gn_gamma.float().reshape(gn_groups, cpg)[None]            # [S][groups][cpg]
            shift = gn_beta.float().reshape(gn_groups, cpg)[None] - mean.float()[:, :, None] * scale
            Xn = X.float().reshape(S, gn_rows_per_sample, K) * scale.reshape(S, 1, K) + shift.reshape(S, 1, K)
            X = Xn.reshape(rows, K).to(T).to(acc_t)
        y = X @ W.to(acc_t).t()
        if bias is not None:
            y = y + bias.to(acc_t)
        if residual is not None:
            y = y + _flat(residual)[: rows * N].reshape(rows, N).to(acc_t)
        _flat(out)[: rows * N].reshape(rows, N).copy_(y.to(out.dtype))

    # ------------------------------------------------------------------------------------
    def gn_stats(self, x, stats, *, rows, C_, groups, rows_per_sample):
        xs = _flat(x)[: rows * C_].reshape(rows // rows_per_sample, rows_per_sample, groups, C_ // groups).double()
        stats.reshape(-1)[: xs.shape[0] * groups * 2].reshape(xs.shape[0], groups, 2).copy_(
            torch.stack([xs.sum(dim=(1, 3)), (xs * xs).sum(dim=(1, 3))], dim=-1))

    def gn_apply(self, x, stats, gamma, beta, y, *, rows, C_, groups, rows_per_sample, eps, silu):
        S = rows // rows_per_sample
        cpg = C_ // groups
        st = stats.reshape(-1)[: S * groups * 2].reshape(S, groups, 2)
        cnt = rows_per_sample * cpg
        mean = st[..., 0] / cnt
        var = (st[..., 1] / cnt - mean * mean).clamp_min(0)
        rstd = 1.0 / torch.sqrt(var + eps)
        xs = _flat(x)[: rows * C_].reshape(S, rows_per_sample, groups, cpg).float()
        yv = (xs - mean.float()[:, None, :, None]) * rstd.float()[:, None, :, None]
        yv = yv.reshape(S, rows_per_sample, C_) * gamma.float() + beta.float()
        if silu:
            yv = F.silu(yv)
        _flat(y)[: rows * C_].reshape(S, rows_per_sample, C_).copy_(yv.to(y.dtype))

    def gn_apply_cs(self, x1, cs1, gamma, beta, y, *, rows, C1, groups, rows_per_sample, eps, silu, x2=None, cs2=None, C2=0,
                    cs_rows=0):
        S, Cc = rows // rows_per_sample, C1 + C2
        cpg = Cc // groups
        nf = rows_per_sample // (cs_rows or rows_per_sample)      # statistics samples (frames) per GroupNorm sample
        assert rows_per_sample % (cs_rows or rows_per_sample) == 0
        xs = _flat(x1)[: rows * C1].reshape(rows, C1).float()
        cs = _flat(cs1)[: S * nf * C1 * 2].reshape(S, nf, C1, 2).sum(dim=1)
        if x2 is not None:
            xs = torch.cat([xs, _flat(x2)[: rows * C2].reshape(rows, C2).float()], dim=1)
            cs = torch.cat([cs, _flat(cs2)[: S * nf * C2 * 2].reshape(S, nf, C2, 2).sum(dim=1)], dim=1)
        st = cs.reshape(S, groups, cpg, 2).sum(dim=2)
        cnt = rows_per_sample * cpg
        mean = st[..., 0] / cnt
        rstd = 1.0 / torch.sqrt((st[..., 1] / cnt - mean * mean).clamp_min(0) + eps)
        yv = (xs.reshape(S, rows_per_sample, groups, cpg) - mean.float()[:, None, :, None]) * rstd.float()[:, None, :, None]
        yv = yv.reshape(S, rows_per_sample, Cc) * gamma.float() + beta.float()
        if silu:
            yv = F.silu(yv)
        _flat(y)[: rows * Cc].reshape(S, rows_per_sample, Cc).copy_(yv.to(y.dtype))

    def layernorm(self, x, gamma, beta, y, *, rows, C_, eps=1e-5, pe=None, pe_div=1, pe_rows=1):
        xs = _flat(x)[: rows * C_].reshape(rows, C_).float()
        yv = F.layer_norm(xs, (C_,), gamma.float(), beta.float(), eps)
        if pe is not None:
            idx = (torch.arange(rows) // pe_div) % pe_rows
            yv = yv + pe.reshape(-1, C_)[idx].float()
        _flat(y)[: rows * C_].reshape(rows, C_).copy_(yv.to(y.dtype))

    def row_stats(self, x, stats, *, rows, C_, eps=1e-5):
        xs = _flat(x)[: rows * C_].reshape(rows, C_).float()
        mean = xs.mean(dim=1)
        var = ((xs - mean[:, None]) ** 2).mean(dim=1)
        stats.reshape(-1, 2)[:rows].copy_(torch.stack([mean, torch.rsqrt(var + eps)], dim=1))

    def softmax_rows(self, x, *, rows, cols, ld, causal_rows=0):
        v = _strided(_flat(x), (rows, cols), (ld, 1), 0)
        s = v.float()
        if causal_rows > 0:
            q = (torch.arange(rows) % causal_rows)[:, None]
            s = s.masked_fill(torch.arange(cols)[None, :] > q, float("-inf"))
        v.copy_(s.softmax(dim=-1).to(x.dtype))

    def embed_tokens(self, ids, table, pos, out, *, rows, seq, C_):
        pidx = torch.arange(rows) % seq
        _flat(out)[: rows * C_].reshape(rows, C_).copy_((table[ids.reshape(-1)[:rows]] + pos[pidx]).to(out.dtype))

    def patchify(self, image, out, *, B, Cin, H, W, P, ld):
        gh, gw = H // P, W // P
        cols = image.reshape(B, Cin, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, Cin * P * P)
        v = _flat(out)[: B * gh * gw * ld].reshape(B * gh * gw, ld)
        v.zero_()
        v[:, : Cin * P * P].copy_(cols.to(out.dtype))

    # ------------------------------------------------------------------------------------
    def concat_channels(self, a_, b_, y, *, rows, c1, c2):
        _flat(y)[: rows * (c1 + c2)].reshape(rows, c1 + c2).copy_(
            torch.cat([_flat(a_)[: rows * c1].reshape(rows, c1), _flat(b_)[: rows * c2].reshape(rows, c2)], dim=1))

    def silu_f32(self, x, y):
        y.copy_(F.silu(x))

    def cast_from_f32(self, x, y, *, rows, cols, ld):
        v = _flat(y)[: rows * ld].reshape(rows, ld)
        v.zero_()
        v[:, :cols].copy_(_flat(x)[: rows * cols].reshape(rows, cols).to(y.dtype))

    def cast_to_f32(self, x, y, *, rows, cols, ld):
        _flat(y)[: rows * cols].reshape(rows, cols).copy_(_flat(x)[: rows * ld].reshape(rows, ld)[:, :cols].float())

    def unet_input(self, latents, mask, first, x, *, B, F, HW, c_latent, c_pad, cfg_dup, mask_frames=1, mode=0):
        lat = latents.reshape(B, c_latent, F, HW)
        out = torch.zeros(B, F, HW, c_pad)
        out[..., :c_latent] = lat.permute(0, 2, 3, 1)
        if mode == 1:      # use_first_frame_condition_concat (reference unet.py:580-586): first-frame latents beside every frame
            if first is not None:
                out[..., c_latent: 2 * c_latent] = first.reshape(B, 1, c_latent, HW).permute(0, 1, 3, 2)
            out = torch.cat([out] * cfg_dup, dim=0)
            _flat(x)[: out.numel()].reshape(out.shape).copy_(out.to(x.dtype))
            return
        if mask is not None:
            m = mask.reshape(B, mask_frames, HW).clamp(0, 1)
            out[..., c_latent] = m if mask_frames > 1 else m.expand(B, F, HW)
        else:
            out[:, 0, :, c_latent] = 1.0
        if first is not None:
            out[:, 0, :, c_latent + 1: 2 * c_latent + 1] = first.reshape(B, c_latent, HW).permute(0, 2, 1)
        out = torch.cat([out] * cfg_dup, dim=0)
        _flat(x)[: out.numel()].reshape(out.shape).copy_(out.to(x.dtype))

    def cfg_ddim_step(self, pred, latents, coef, *, B, F, HW, c_latent, ld, cfg, guidance, pred_type, clip_sample,
                      pred_single=None, video_scale=0.0, variance_noise=None, sigma=0.0, clipped_model_output=False):
        n = (2 if cfg else 1) * B * F * HW
        p = _flat(pred)[: n * ld].reshape(-1, B, F, HW, ld)[..., :c_latent].float()
        v = p[0] + guidance * (p[1] - p[0]) if cfg else p[0]
        if pred_single is not None:
            s1 = _flat(pred_single)[: B * F * HW * ld].reshape(B, F, HW, ld)[..., :c_latent].float()
            v = s1 + video_scale * (p[0] - s1) + guidance * (p[1] - p[0])
        v = v.permute(0, 3, 1, 2)                                  # B C F HW
        x = latents.reshape(B, c_latent, F, HW)
        sa, sb, sap, sbp = [float(c) for c in coef.reshape(-1)[:4]]
        if pred_type == 1:
            x0, eps = sa * x - sb * v, sa * v + sb * x
        elif pred_type == 0:
            x0, eps = (x - sb * v) / sa, v
        else:
            x0, eps = v, v
        if clip_sample:
            x0 = x0.clamp(-1, 1)
        if clipped_model_output:                                   # scheduling_ddim.py:342-344
            eps = (x - sa * x0) / sb
        nx = sap * x0 + sbp * eps
        if variance_noise is not None:                             # eta > 0 (:346-363)
            nx = nx + sigma * variance_noise.reshape(B, c_latent, F, HW).float()
        x.copy_(nx)

    def nchw_to_nhwc(self, z, x, *, N, C_, HW, c_pad, scale):
        out = torch.zeros(N, HW, c_pad)
        out[..., :C_] = z.reshape(N, C_, HW).permute(0, 2, 1) * scale
        _flat(x)[: out.numel()].reshape(out.shape).copy_(out.to(x.dtype))

    def nhwc_to_nchw(self, x, y, *, N, C_, HW, ld, mul=1.0, add=0.0, lo=-3.0e38, hi=3.0e38):
        v = _flat(x)[: N * HW * ld].reshape(N, HW, ld)[..., :C_].float() * mul + add
        _flat(y)[: N * C_ * HW].reshape(N, C_, HW).copy_(v.clamp(lo, hi).permute(0, 2, 1))

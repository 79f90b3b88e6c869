"""Chunked restatement of the reference's memory read for banks whose affinity cannot be materialised.

TEST INFRASTRUCTURE ONLY (see oracle/stm_oracle.py's header): imported by ``tests/`` and by the parity leg of
``bench.py`` as the checker, never by ``mivos_amd``.

The reference computes the whole ``[T*H*W, H*W]`` affinity, ``torch.topk`` over the memory axis, a softmax over the k
survivors and a dense ``bmm`` with the scattered weights (`model/propagation/prop_net.py:47-63` softmax_w_g_top,
`:81-108` EvalMemoryReader.forward).  Every query column of that computation is independent of the others, so the same
arithmetic can be run on blocks of ``qblock`` queries: affinity block -> topk -> softmax -> weighted sum of the k selected
value rows (the scattered zeros of the dense bmm contribute exact zeros).  That is what this file does, in plain torch on
whatever device the operands live on (CPU, or the GPU inside a `-m gpu` test: at BASELINE config 5's bank depth the
affinity is 160 GB as one tensor and 1.2 M x 256 x 8 B = 2.5 GB per block), in fp32 like the reference or in fp64 as the
arbitration truth.

PIN: ``tests/test_oracle_golden.py::test_chunked_memory_read_equals_the_materialised_oracle`` compares it with
``stm_oracle.memory_read`` (itself bit-identical to the unmodified reference, tests/golden/ops_small.npz) on CPU: same
index sets, readout within fp32 rounding (the dense bmm sums in memory order, the gather in rank order).
"""
import math

import torch


def memory_read_rows(keys, values, qk, top_k, dtype=torch.float32, qblock=256, want_margin=True):
    """keys [K, n_mem, 128], values [K, n_mem, 512] or None, qk [n_q, 128] (the engine's row layout; the reference's
    mk[b, :, m] is keys[b, m]) -> dict(readout [K, n_q, 512] (None without values), idx [K, n_q, k] int64 best first,
    weights [K, n_q, k], margin [K, n_q] = score(rank k) - score(rank k+1) (inf when n_mem == k)).

    prop_net.py:85-88: affinity[m, q] = sum_c mk[c, m] * (qk[c, q] / sqrt(CK)); :54-59: topk, exp(v - v_max) / sum;
    :104-108: mem = mv @ affinity.  top_k=None is the reference's full softmax (prop_net.py:99-102): computed with a running
    (max, denominator, numerator) over memory blocks, idx / weights / margin are then None."""
    K, n_mem, ck = keys.shape
    n_q = qk.shape[0]
    dev = keys.device
    q = qk.to(dtype) / math.sqrt(ck)                                   # the reference divides the query (prop_net.py:86)
    readout = None if values is None else torch.empty((K, n_q, values.shape[2]), dtype=dtype, device=dev)
    if top_k is None:
        for o in range(K):
            ko = keys[o].to(dtype)
            vo = values[o].to(dtype)
            for q0 in range(0, n_q, qblock):
                a = q[q0:q0 + qblock] @ ko.t()                         # [B, n_mem]
                p = torch.softmax(a, dim=1)                            # F.softmax(affinity, dim=1) of the reference, per column
                readout[o, q0:q0 + qblock] = p @ vo
        return dict(readout=readout, idx=None, weights=None, margin=None)
    idx = torch.empty((K, n_q, top_k), dtype=torch.int64, device=dev)
    wgt = torch.empty((K, n_q, top_k), dtype=dtype, device=dev)
    margin = torch.full((K, n_q), float("inf"), dtype=dtype, device=dev)
    kk = min(top_k + 1, n_mem) if want_margin else top_k
    for o in range(K):
        ko = keys[o].to(dtype)
        for q0 in range(0, n_q, qblock):
            a = q[q0:q0 + qblock] @ ko.t()                             # [B, n_mem]
            v, i = torch.topk(a, kk, dim=1)
            if kk > top_k:
                margin[o, q0:q0 + qblock] = v[:, top_k - 1] - v[:, top_k]
            v, i = v[:, :top_k], i[:, :top_k]
            e = torch.exp(v - v[:, :1])
            e = e / e.sum(dim=1, keepdim=True)
            idx[o, q0:q0 + qblock], wgt[o, q0:q0 + qblock] = i, e
            if values is not None:
                rows = values[o][i.reshape(-1)].to(dtype).view(i.shape[0], top_k, -1)
                readout[o, q0:q0 + qblock] = (e.unsqueeze(2) * rows).sum(1)
    return dict(readout=readout, idx=idx, weights=wgt, margin=margin)


def memory_read(mk, mv, qk, top_k, dtype=None, qblock=256):
    """Reference-shaped wrapper: mk [B,CK,T,H,W], mv [B,CV,T,H,W], qk [1,CK,H,W] -> [B,CV,H,W] (stm_oracle.memory_read)."""
    B, CK = mk.shape[:2]
    CV = mv.shape[1]
    H, W = qk.shape[-2:]
    dtype = dtype or mk.dtype
    keys = mk.reshape(B, CK, -1).transpose(1, 2)
    vals = mv.reshape(B, CV, -1).transpose(1, 2)
    q = qk.reshape(CK, H * W).t()
    r = memory_read_rows(keys, vals, q, top_k, dtype=dtype, qblock=qblock, want_margin=False)
    return r["readout"].transpose(1, 2).reshape(B, CV, H, W)

#!/usr/bin/env python3
"""CPU (torch fp32): which fp16 rounding of the two TransReID streams carries DATOR's embedding error?  The fp32 forward of both
streams + the fp32 fusion head (the oracle's), with a .half() round trip switched on per block (all weights and activations of the
block) or per kind of rounding point.  python tools/sim_dator_rounding.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibloc_amd import dator as D  # noqa: E402
from oracle import dator_oracle as do  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402
import bench  # noqa: E402

F = torch.nn.functional
torch.set_num_threads(8)
cfg = D.STREAM_CFG
ALLR = {'w_q', 'w_k', 'w_v', 'w_o', 'w_fc1', 'w_fc2', 'ln1', 'qkv', 'P', 'ao', 'ln2', 'gelu'}
WS = {'w_q', 'w_k', 'w_v', 'w_o', 'w_fc1', 'w_fc2'}


def h(t, on):
    return t.half().float() if on else t


def stream(w, x, Rl, patch):
    B = x.shape[0]
    t = F.conv2d(h(x, patch), h(w['patch.w'], patch), w['patch.b'], stride=16).flatten(2).transpose(1, 2)
    t = torch.cat([w['cls'].reshape(1, 1, -1).expand(B, -1, -1), t], 1) + w['pos'].unsqueeze(0)
    hd = 64
    norms = []
    for l in range(11):
        R = Rl.get(l, set())
        W = lambda n: h(w[n], ('w_' + n.split('.')[1]) in R)
        p = f'l{l}.'
        a = h(F.layer_norm(t, (768,), w[p + 'ln1.g'], w[p + 'ln1.b'], 1e-6), 'ln1' in R)
        q = h(F.linear(a, W(p + 'q.w'), w[p + 'q.b']), 'qkv' in R).view(B, -1, 12, hd).transpose(1, 2)
        k = h(F.linear(a, W(p + 'k.w'), w[p + 'k.b']), 'qkv' in R).view(B, -1, 12, hd).transpose(1, 2)
        v = h(F.linear(a, W(p + 'v.w'), w[p + 'v.b']), 'qkv' in R).view(B, -1, 12, hd).transpose(1, 2)
        P = h(torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1), 'P' in R)
        o = h((P @ v).transpose(1, 2).reshape(B, -1, 768), 'ao' in R)
        n0 = float(t.norm())
        br = F.linear(o, W(p + 'o.w'), w[p + 'o.b'])
        t = t + br
        a = h(F.layer_norm(t, (768,), w[p + 'ln2.g'], w[p + 'ln2.b'], 1e-6), 'ln2' in R)
        g = h(F.gelu(F.linear(a, W(p + 'fc1.w'), w[p + 'fc1.b'])), 'gelu' in R)
        br2 = F.linear(g, W(p + 'fc2.w'), w[p + 'fc2.b'])
        norms.append((round(n0, 1), round(float(br.norm()), 1), round(float(br2.norm()), 1)))
        t = t + br2
    return t, norms


def main():
    rw, dw, hw = D.random_stream_weights(20), D.random_stream_weights(21), D.random_head_weights(22)
    frw = {k: torch.from_numpy(v) for k, v in D.fold_lora(rw).items()}
    fdw = {k: torch.from_numpy(v) for k, v in D.fold_lora(dw).items()}
    crops = bench.Crops("dator", 21)
    rng = np.random.default_rng(0)
    rgb, dep = crops.variants(list(range(8)), rng, "cpu")
    xr = torch.from_numpy(np.stack([do.preprocess_rgb(c) for c in rgb.numpy()]))
    xd = torch.from_numpy(np.stack([do.preprocess_depth(d) for d in dep.numpy()]))
    with torch.no_grad():
        tr0, norms = stream(frw, xr, {}, False)
        td0, _ = stream(fdw, xd, {}, False)
        ref = torch.from_numpy(do.head_forward(hw, tr0.numpy(), td0.numpy()))
        print('rgb stream norms (resid, attn branch, mlp branch):', norms)

        def rel(Rl, patch):
            tr, _ = stream(frw, xr, Rl, patch)
            td, _ = stream(fdw, xd, Rl, patch)
            e = torch.from_numpy(do.head_forward(hw, tr.numpy(), td.numpy()))
            tok = float(((tr - tr0).norm() / tr0.norm() + (td - td0).norm() / td0.norm()) / 2)
            return float((torch.linalg.norm(e - ref, dim=1) / torch.linalg.norm(ref, dim=1)).mean()), tok
        print('patch only (emb rel, token rel)', rel({}, True))
        for l in range(11):
            print('block', l, 'all roundings only', rel({l: set(ALLR)}, False), ' weights only', rel({l: set(WS)}, False))
        full = {l: set(ALLR) for l in range(11)}
        print('all', rel(full, True))
        print('all weights exact', rel({l: ALLR - WS for l in range(11)}, False))
        print('all activations exact', rel({l: set(WS) for l in range(11)}, True))
        for n in sorted(ALLR - WS):
            print('only activation kind', n, 'in every block', rel({l: {n} for l in range(11)}, False))


if __name__ == "__main__":
    main()

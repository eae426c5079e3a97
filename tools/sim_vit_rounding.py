#!/usr/bin/env python3
"""CPU (torch fp32): which fp16 rounding of the HIP ViT forward carries the embedding error?  The fp32 forward of ViT-B/14 with seeded
random-init weights, with a .half() round trip switched on per rounding point (weights per matrix, LayerNorm outputs, q/k/v, softmax
probabilities, attention output, GELU hidden, pixel input) and per block.  Result (DESIGN (c)): the patch embedding and blocks 0-1 carry
~75 % of the error variance because the residual stream is still small there -> ibloc_amd.vit.DEFAULT_PRECISION gives those operands a
second fp16 term.  python tools/sim_vit_rounding.py"""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from ibloc_amd import vit as V
import dataclasses
F=torch.nn.functional
torch.set_num_threads(4)
cfg=dataclasses.replace(V.CONFIGS['dinov2_vitb14'],pos_interp='size')
w={k:torch.from_numpy(v) for k,v in V.random_weights(cfg,20).items()}
import bench
crops=bench.Crops('dinov2_vitb14',21)
rng=np.random.default_rng(0)
from oracle import vit_oracle as vo
from ibloc_amd import preprocess as pp
u8=crops.variants(list(range(6)),rng,'cpu').numpy()
x=torch.from_numpy(np.stack([vo.preprocess_crop(c,pp.RECIPES['dinov2']) for c in u8]))
def h(t,on): return t.half().float() if on else t
ALLR={'in','w_q','w_k','w_v','w_o','w_fc1','w_fc2','ln1','qkv','P','ao','ln2','gelu'}
def fwd2(Rl, patch=True):
    B=x.shape[0]
    t=F.conv2d(h(x,patch),h(w['patch.w'],patch),w['patch.b'],stride=14).flatten(2).transpose(1,2)
    t=torch.cat([w['cls'].reshape(1,1,-1).expand(B,-1,-1),t],1)
    t=t+vo.interpolate_pos(w['pos'],cfg.pos_grid,cfg.grid,'size').unsqueeze(0)
    hd=64
    norms=[]
    for l in range(12):
        R=Rl.get(l,set())
        W=lambda n: h(w[n],('w_'+n.split('.')[1]) in R)
        p=f'l{l}.'
        a=h(F.layer_norm(t,(768,),w[p+'ln1.g'],w[p+'ln1.b'],1e-6),'ln1' in R)
        q=h(F.linear(a,W(p+'q.w'),w[p+'q.b']),'qkv' in R).view(B,-1,12,hd).transpose(1,2)
        k=h(F.linear(a,W(p+'k.w'),w[p+'k.b']),'qkv' in R).view(B,-1,12,hd).transpose(1,2)
        v=h(F.linear(a,W(p+'v.w'),w[p+'v.b']),'qkv' in R).view(B,-1,12,hd).transpose(1,2)
        P=h(torch.softmax(q@k.transpose(-1,-2)*hd**-0.5,-1),'P' in R)
        o=h((P@v).transpose(1,2).reshape(B,-1,768),'ao' in R)
        br=F.linear(o,W(p+'o.w'),w[p+'o.b'])*w[p+'ls1']
        n0=float(t.norm()); t=t+br
        a=h(F.layer_norm(t,(768,),w[p+'ln2.g'],w[p+'ln2.b'],1e-6),'ln2' in R)
        g=h(F.gelu(F.linear(a,W(p+'fc1.w'),w[p+'fc1.b'])),'gelu' in R)
        br2=F.linear(g,W(p+'fc2.w'),w[p+'fc2.b'])*w[p+'ls2']
        norms.append((n0,float(br.norm()),float(br2.norm())))
        t=t+br2
    return F.layer_norm(t[:,0],(768,),w['ln_f.g'],w['ln_f.b'],1e-6), norms
with torch.no_grad():
    ref,norms=fwd2({},False)
    print('norms (resid, attn branch, mlp branch):',[tuple(round(v,1) for v in n) for n in norms])
    def rel2(Rl,patch=True):
        o,_=fwd2(Rl,patch); return float((torch.linalg.norm(o-ref,dim=1)/torch.linalg.norm(ref,dim=1)).mean())
    print('patch only', rel2({}))
    for l in (0,1):
        for n in sorted(ALLR-{'in'}): print('layer',l,'only',n,rel2({l:{n}},False))
    full={l:set(ALLR) for l in range(12)}
    print('all',rel2(full))
    WS={'w_q','w_k','w_v','w_o','w_fc1','w_fc2'}
    def variant(name,mod):
        R={l:set(ALLR) for l in range(12)}
        mod(R); print(name, rel2(R))
    variant('L0 weights exact', lambda R: R.__setitem__(0,R[0]-WS))
    variant('L0,1 weights exact', lambda R: (R.__setitem__(0,R[0]-WS),R.__setitem__(1,R[1]-WS)))
    variant('L0 all exact', lambda R: R.__setitem__(0,set()))
    variant('L0 all exact, L1 weights exact', lambda R: (R.__setitem__(0,set()),R.__setitem__(1,R[1]-WS)))
    variant('L0,1 all exact', lambda R: (R.__setitem__(0,set()),R.__setitem__(1,set())))
    variant('L0 weights + ln1 ln2 gelu ao exact', lambda R: R.__setitem__(0,{'qkv','P'}))
    variant('L0,1 weights + ln1 ln2 gelu ao exact', lambda R: (R.__setitem__(0,{'qkv','P'}),R.__setitem__(1,{'qkv','P'})))
    variant('L0,1,2 weights + ln1 ln2 gelu ao exact', lambda R: (R.__setitem__(0,{'qkv','P'}),R.__setitem__(1,{'qkv','P'}),R.__setitem__(2,{'qkv','P'})))

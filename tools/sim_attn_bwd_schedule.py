"""The launch of the five-product attention backward (attn_bwd_fused_kernel<8>) simulated on the host for the configs[3] AR batches:
workgroups = (batch row, head) x key chunk, dealt in index order to whichever of 256 CUs frees up first (a min-heap); a workgroup's
time = its query tiles x 1.0 with 5..8 active 32-key waves or x 0.6 with <= 4 (the kernel's measured 10.7 vs 6.5 us per tile,
DESIGN 3.5).  Prints, per batch and chunk count: mean load per CU, the makespan in index order and in LPT order (heaviest first),
and the pure matrix work (active waves / 8).  CPU only:  python tools/sim_attn_bwd_schedule.py"""
import heapq, sys
import numpy as np, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from valle2_amd import ConfigValle, synth
cfg = ConfigValle(d_model=512,n_heads=8,dim_feedforward=2048,num_layers=12,dropout=0.0,batch_size=16)
def visit_cost(n_act):
    if n_act<=0: return 0.05   # barrier-only visit
    return 0.6 if n_act<=4 else 1.0
def sim(order, works):
    cu=[0.0]*256; heapq.heapify(cu); end=0
    for i in order:
        t=heapq.heappop(cu)+works[i]; end=max(end,t); heapq.heappush(cu,t)
    return end
tot=[]
for seed in range(100,108):
    b = synth.synth_ar_batch(cfg,16,seed=seed)
    tl=b['tokens_lens'].numpy(); cl=b['codes_lens'].numpy()
    xl=int(tl.max()); T=xl+int(cl.max())
    # kv_len per row: padding mask = pad(codes pad mask, (max tokens,0)) -> valid keys = xl + codes_len
    kvl = xl+cl
    nt=(T+31)//32
    for nc in (4,5,6,8):
        keys=((T+nc-1)//nc+31)//32*32
        works=[];ideal=0
        for c in range(nc):
            kb0=c*keys
            for bb in range(16):
                for h in range(8):
                    w=0;
                    if kb0>=kvl[bb]: works.append(0.02); continue
                    t_first = kb0//32 if kb0>=xl else 0
                    nwk=(min(kb0+keys,T)-kb0+31)//32
                    for qt in range(t_first,nt):
                        n_act=min(nwk,(kvl[bb]-kb0+31)//32)
                        lim=max(xl-1,qt*32+31)-kb0
                        n_act=min(n_act,0 if lim<0 else lim//32+1)
                        w+=visit_cost(n_act); ideal+=max(n_act,0)/8
                    works.append(w)
        n=len(works)
        cur=sim(range(n),works)
        lpt=sim(sorted(range(n),key=lambda i:-works[i]),works)
        print(seed,'T',T,'nc',nc,'keys',keys,'sum/256 %.1f'%(sum(works)/256),'cur %.1f'%cur,'lpt %.1f'%lpt,'mfma-ideal %.1f'%(ideal/256))

"""Slot model of the attention work list (profiles/r03_attention_schedule.txt): makespan of the 8 XCD queues on 64 workgroup slots each,
work item cost = 60 + partner rows, for the round-2 order (size-sorted list cut into equal-cost queues) and the round-3 order
(groups dealt in snake order first).  usage: python profiles/attention_schedule_model.py"""
import sys, heapq
sys.path.insert(0,'/root/repo')
import numpy as np
from equidock_public_amd import synthetic
def items_of(sizes):
    items=[]; nl=sum(a for a,b in sizes); lo=0; ro=nl
    groups=[]
    for a,b in sizes:
        for (a0,a1,o0,o1) in ((lo,lo+a,ro,ro+b),(ro,ro+b,lo,lo+a)):
            g=[]
            for s0 in range(a0,a1,32): g.append((s0,min(s0+32,a1),o0,o1))
            groups.append(g)
        lo+=a; ro+=b
    return groups
def fill(items_sorted):
    total=sum(it[3]-it[2] for it in items_sorted)
    queues=[[] for _ in range(8)]; c=0; acc=0
    for it in items_sorted:
        queues[c].append(it); acc+=it[3]-it[2]
        if c<7 and acc*8>=total*(c+1): c+=1
    return queues
def current(groups):
    items=[it for g in groups for it in g]
    items=sorted(items,key=lambda it:-(it[3]-it[2]))
    return fill(items)
def mixed(groups):
    # groups by decreasing partner size dealt round-robin (snake) into 8 lists, concatenated, equal-cost contiguous fill, big first inside a queue
    gs=sorted(groups,key=lambda g:-(g[0][3]-g[0][2]))
    lists=[[] for _ in range(8)]
    for i,g in enumerate(gs):
        r=i%16; c=r if r<8 else 15-r
        lists[c].append(g)
    items=[it for l in lists for g in l for it in g]
    q=fill(items)
    return [sorted(x,key=lambda it:-(it[3]-it[2])) for x in q]
def makespan(queues, c0=60, slots=64, wg_per_item=2):
    # dq pass WGs of all items first (list order interleaved over XCDs -> per XCD its queue order), then kv pass
    ms=[]
    for q in queues:
        h=[0.0]*slots; heapq.heapify(h)
        for p in range(2):
            for it in q:
                for _ in range(wg_per_item):
                    t=heapq.heappop(h); heapq.heappush(h,t+c0+(it[3]-it[2]))
        ms.append(max(h))
    tot=sum((c0+it[3]-it[2])*2*wg_per_item for q in queues for it in q)
    return max(ms), tot/(8*slots), [round(m) for m in ms]
for name,sizes in (('R',synthetic.realistic_sizes(64,5055)),('C',[(300,300)]*64),('B',[(200,200)]*8),('E',[(2000,2000)]*4),('A',[(200,200)])):
    g=items_of(sizes)
    for nm,f in (('current',current),('mixed',mixed)):
        m,ideal,per=makespan(f(g))
        print(name,nm,'makespan',round(m),'ideal',round(ideal),'eff',round(ideal/m,3),per)

import itertools, sys, random
def run(n, order_seed, extra=0):
    # ring 0->1->...->n-1->0 plus `extra` upstream cells pointing at ring cells
    N=n+extra
    sb=[(i+1)%n for i in range(n)]+[random.Random(order_seed+7).randrange(n) for _ in range(extra)]
    sbext=list(sb)
    # thread state machine
    st=[dict(pc=0,b=None,bb=None,it=0,out=None) for _ in range(N)]
    rng=random.Random(order_seed)
    live=list(range(N))
    while live:
        i=rng.choice(live); t=st[i]
        if t['pc']==0: t['b']=sb[i]; t['pc']=1
        elif t['pc']==1:
            if t['it']>=48: t['pc']=3; continue
            t['bb']=sb[t['b']]; t['pc']=2
        elif t['pc']==2:
            if t['bb']==t['b']: t['pc']=3
            else: sb[i]=t['bb']; t['b']=t['bb']; t['it']+=1; t['pc']=1
        elif t['pc']==3:
            b=t['b']
            t['out']= sbext[b] if sb[b]==b else b
            live.remove(i)
    J=[t['out'] for t in st]
    return J
def roots(J):
    # follow J; detect components: count of terminal structures (self loops or rings)
    N=len(J); term=set()
    for i in range(N):
        seen=[];x=i
        while x not in seen:
            seen.append(x); x=J[x]
        cyc=tuple(sorted(seen[seen.index(x):]))
        term.add(cyc)
    return term
bad=0
for n in (2,3,4,5):
    for seed in range(20000):
        J=run(n,seed,extra=3)
        t=roots(J)
        if len(t)!=1:
            bad+=1
            if bad<10: print("SPLIT n",n,"seed",seed,J,t)
print("bad",bad)

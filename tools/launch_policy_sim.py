#!/usr/bin/env python3
"""Discrete-event model of the level server: R regions in flight, each asking for its next
level (kinds and kernel times as measured at 100 regions in flight), NS launch streams, a stream busy until the
slowest level of its batch is done.  Prints levels per ms, mean wait between request and launch (ms) and mean batch
size for four choices of the next kind, and for more streams.  It reproduces the measured wait (0.29 vs 0.33-0.39 ms)
and says the choice of kind does not matter; only more concurrent launches would, which the hardware queues refuse."""
import heapq, random, collections
def sim(policy, R=100, NS=11, T=3000.0, seed=1, maxb=48):
    rnd=random.Random(seed)
    kinds=[("plain",0.25,0.10),("nb1",0.06,0.80),("nb2",0.43,1.30),("nb3",0.22,1.65),("nb4",0.04,2.05)]
    def next_kind():
        x=rnd.random(); a=0
        for i,(n,p,d) in enumerate(kinds):
            a+=p
            if x<a: return i
        return 2
    ev=[]  # (time, type, data)
    waiting=collections.defaultdict(list)  # kind -> [(t_req, region)]
    streams=[0]*NS  # busy counts
    t=0.0; done=0; wait_sum=0.0; nlaunch=0
    for r in range(R): heapq.heappush(ev,(rnd.random()*2.0,"req",r))
    def try_launch(now):
        nonlocal wait_sum, nlaunch
        while True:
            fs=[i for i in range(NS) if streams[i]==0]
            ks=[k for k in waiting if waiting[k]]
            if not fs or not ks: return
            if policy in ("oldest","anykind"): k=min(ks,key=lambda k: waiting[k][0][0])
            elif policy=="largest": k=max(ks,key=lambda k: len(waiting[k]))
            elif policy=="aged":
                old=[k for k in ks if now-waiting[k][0][0]>0.5]
                k=min(old,key=lambda k: waiting[k][0][0]) if old else max(ks,key=lambda k: len(waiting[k]))
            elif policy=="shortfirst": k=min(ks)   # plain first, then nb1...
            if policy=="anykind":                  # one kernel serves every kind: all waiting levels leave together
                b=[]
                for k2 in ks:
                    b+=[(tr,r,k2) for (tr,r) in waiting[k2]]; waiting[k2]=[]
                b.sort(); rest=b[maxb:]; b=b[:maxb]
                for (tr,r,k2) in rest: waiting[k2].append((tr,r))
            else:
                b=[(tr,r,k) for (tr,r) in waiting[k][:maxb]]; waiting[k]=waiting[k][maxb:]
            s=fs[0]; streams[s]=len(b); nlaunch+=1
            for (tr,r,k2) in b:
                wait_sum+=now-tr
                d=kinds[k2][2]*rnd.uniform(0.85,1.25)+0.03
                heapq.heappush(ev,(now+d,"fin",(r,s)))
    while ev:
        t,typ,data=heapq.heappop(ev)
        if t>T: break
        if typ=="req":
            k=next_kind(); waiting[k].append((t,data)); try_launch(t)
        else:
            r,s=data; streams[s]-=1; done+=1
            heapq.heappush(ev,(t+ (0.25 if rnd.random()<0.25 else 0.02),"req",r))
            try_launch(t)
    return done/T, wait_sum/max(done,1), done/max(nlaunch,1)
for pol in ("oldest","largest","aged","shortfirst","anykind"):
    for ns in (11,):
        print(pol, ns, ["%.3f"%x for x in sim(pol,NS=ns)])
print("streams 16 oldest", ["%.3f"%x for x in sim("oldest",NS=16)])
print("streams 32 oldest", ["%.3f"%x for x in sim("oldest",NS=32)])

"""Why the 300-iteration class cannot stop early (round 5): Cephes incbcf on Hi-C rows of the swapped class (a ~ 3e8, b = count,
x = 1 - prior) - the convergent r = pk/qk keeps moving by ~1e-8 RELATIVE up to the cap (every step cancels pk = pkm1 - ~pkm2), so
scipy.special.bdtrc itself carries that noise and only the same 300 iterations in the same order land within 1e-10 of it.
    python profiles/cf_wander.py"""
import numpy as np, math
MACHEP=1.11022302462515654042e-16
big=4.503599627370496e15; biginv=2.22044604925031308085e-16
def incbcf_trace(a,b,x):
    k1=a;k2=a+b;k3=a;k4=a+1.0;k5=1.0;k6=b-1.0;k7=k4;k8=a+2.0
    pkm2=0.0;qkm2=1.0;pkm1=1.0;qkm1=1.0;ans=1.0;r=1.0;thresh=3.0*MACHEP
    out=[]
    for n in range(300):
        xk=-(x*k1*k2)/(k3*k4)
        pk=pkm1+pkm2*xk; qk=qkm1+qkm2*xk
        pkm2=pkm1;pkm1=pk;qkm2=qkm1;qkm1=qk
        xk=(x*k5*k6)/(k7*k8)
        pk=pkm1+pkm2*xk; qk=qkm1+qkm2*xk
        pkm2=pkm1;pkm1=pk;qkm2=qkm1;qkm1=qk
        if qk!=0: r=pk/qk
        if r!=0:
            t=abs((ans-r)/r); ans=r
        else: t=1.0
        out.append(r)
        if t<thresh: break
        k1+=1.0;k2+=1.0;k3+=2.0;k4+=2.0;k5+=1.0;k6-=1.0;k7+=2.0;k8+=2.0
        if abs(qk)+abs(pk)>big:
            pkm2*=biginv;pkm1*=biginv;qkm2*=biginv;qkm1*=biginv
        if abs(qk)<biginv or abs(pk)<biginv:
            pkm2*=big;pkm1*=big;qkm2*=big;qkm1*=big
    return out
rng=np.random.default_rng(1)
n=3.0e8
worst={}
iters=[]
for trial in range(3000):
    count=int(rng.integers(1,40))
    # expected = n*prior > count (observed < expected): the swapped class
    expected=count*np.exp(rng.uniform(0.05,3.0))
    prior=expected/n
    aa=float(count); bb=n-count+1.0; xx=prior
    # swapped: incbcf(bb, aa, 1-xx)
    tr=incbcf_trace(bb,aa,1.0-xx)
    iters.append(len(tr))
    fin=tr[-1]
    for k in (10,20,30,40,60,80,120,200):
        if len(tr)>k:
            d=abs(tr[k-1]-fin)/abs(fin)
            worst[k]=max(worst.get(k,0),d)
print("iterations: min %d median %d max %d; share at the cap %.3f"%(min(iters),np.median(iters),max(iters),np.mean(np.array(iters)==300)))
for k in sorted(worst): print("relative distance of r after %3d iterations from the final r: max %.3e"%(k,worst[k]))

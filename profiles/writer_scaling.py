import numpy as np, time, os, sys
sys.path.insert(0, os.getcwd())
from fithic_amd import _capi
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
rng=np.random.default_rng(0)
n=24_000_000
names=["chr%d"%i for i in range(1,6)]
c1=rng.integers(0,5,n).astype(np.int32); m1=(rng.integers(0,50000,n)*5000+2500).astype(np.int32); m2=(m1+rng.integers(4,400,n)*5000).astype(np.int32)
cnt=rng.integers(1,60,n).astype(np.int32); p=rng.random(n)**3; q=np.minimum(p*7,1); b1=np.exp(rng.normal(0,.25,n)); b2=np.exp(rng.normal(0,.25,n)); e=rng.random(n)*20
for th in (8,32,64,128,256):
    t=time.time(); w=_capi.host_write_significances("/dev/shm/sig.gz",names,c1,m1,c1,m2,cnt,p,q,b1,b2,e,0,20000,2000000,gzip_level=1,threads=th); dt=time.time()-t
    print("threads",th,"%.2fs"%dt, "%.2f M rows/s"%(w/dt/1e6), flush=True)
os.remove("/dev/shm/sig.gz")

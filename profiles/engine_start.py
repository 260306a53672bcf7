import time, sys
t0=time.time()
sys.path.insert(0,'/root/repo')
from fithic_amd import _capi
t1=time.time()
L=_capi.lib()
t2=time.time()
c=_capi.Context(0)
t3=time.time()
c.close()
c2=_capi.Context(0)
t4=time.time()
print("import %.3f lib %.3f first ctx %.3f second ctx %.3f"%(t1-t0,t2-t1,t3-t2,t4-t3))

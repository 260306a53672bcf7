import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time, os, ctypes
from fithic_amd import _capi
L=_capi.lib()
F64=ctypes.POINTER(ctypes.c_double); I32=ctypes.POINTER(ctypes.c_int32)
def ptr(a,t): return a.ctypes.data_as(ctypes.POINTER(t))
for name in ["C2","C3","C3w","C5"]:
    g=np.load("tests/golden/f14_%s_fit.npz"%name)
    x=np.ascontiguousarray(g["x_sorted"],np.float64); y=np.ascontiguousarray(g["y_sorted"],np.float64)
    s=float(g["spl_s_fp_ier"][0]); m=len(x)
    t=np.zeros(m+8); c=np.zeros(m+8); nk=ctypes.c_int32(0); fp=ctypes.c_double(0); ier=ctypes.c_int32(0); rs=ctypes.c_int32(0)
    reps=200
    t0=time.perf_counter()
    for _ in range(reps):
        rc=L.fhx_host_spline_fit(ptr(x,ctypes.c_double),ptr(y,ctypes.c_double),m,s,ptr(t,ctypes.c_double),ptr(c,ctypes.c_double),ctypes.byref(nk),ctypes.byref(fp),ctypes.byref(ier),ctypes.byref(rs))
    dt=(time.perf_counter()-t0)/reps
    ok=np.array_equal(t[:nk.value], g["spl_t"]) and np.array_equal(c[:len(g["spl_c"])], g["spl_c"])
    # eval + pava
    xs=np.ascontiguousarray(g["splineX"],np.float64); out=np.zeros(len(xs)); out2=np.zeros(len(xs))
    t1=time.perf_counter()
    for _ in range(reps):
        L.fhx_host_spline_eval(ptr(t,ctypes.c_double),ptr(c,ctypes.c_double),nk.value,ptr(xs,ctypes.c_double),len(xs),ptr(out,ctypes.c_double))
        L.fhx_host_pava_decreasing(ptr(out,ctypes.c_double),len(xs),ptr(out2,ctypes.c_double))
    dt2=(time.perf_counter()-t1)/reps
    print("%s: m=%d knots=%d restarted=%d fit %.1f us (bit-identical %s); eval+pava over %d points %.1f us" % (name,m,nk.value,rs.value,dt*1e6,ok,len(xs),dt2*1e6))

import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
from medicalseg_amd._lib import MskConvDesc
from medicalseg_amd.device import Tensor, get_device
dev=get_device(); n,s,ci=2,128,32; vox=n*s**3
x=Tensor(dev,dev.malloc(vox*ci*4),n,s,s,s,ci,ci,None)
dev.h2d(x.ptr,np.random.default_rng(0).standard_normal(vox*ci,dtype=np.float32))
cd=MskConvDesc(5,5,5,1,1,1,2,2,2)
for co in (1,2,3,4):
    y=Tensor(dev,dev.malloc(vox*co*4),n,s,s,s,co,co,None)
    w=dev.malloc(co*ci*125*4); dev.h2d(w,(np.random.default_rng(1).standard_normal(co*ci*125)*0.01).astype(np.float32)); b=dev.small(co)
    f=lambda: dev.call("msk_conv3d_fwd",cd,x.msk(),C.c_void_p(w),C.c_void_p(b),y.msk())
    f(); dev.sync(); dev.timer_start()
    for _ in range(5): f()
    ms=dev.timer_stop()/5
    print(f"32->{co}: {ms:.3f} ms  {2*125*ci*co*vox/ms/1e9:.1f} TFLOP/s")

import sys, numpy as np, ctypes as C
sys.path.insert(0,'.')
from cpu_tsdf_amd import capi
capi.use_test_library()  # knobs / selftest hooks live in libtsdf_hip_test.so (include/tsdf_hip_test.h)
W,H=640,480
img=np.arange(2*W*H,dtype=np.float32).reshape(2,H,W)+1.0
uv=np.array([[0,0],[5,7],[W-1,H-1],[W,0],[W,3],[W+5,3],[-1,3],[3,H],[3,-1],[W,H-1],[2*W,0],[0,H+100],[-1,-1],[W-1,0],[0,H-1]],dtype=np.int32)
for plane in (0,1):
    out=np.zeros(len(uv),np.uint32)
    capi.check(capi.load().tsdf_hip_selftest_struct_oob(capi.as_f32p(img), W,H,2,plane, uv.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(C.POINTER(C.c_uint32)), len(uv)),"oob")
    vals=out.view(np.float32)
    for (u,v),x in zip(uv,vals):
        exp = img[plane,v,u] if 0<=u<W and 0<=v<H else 0.0
        print(plane,(int(u),int(v)),x,"expected",exp,"OK" if x==exp else "MISMATCH")

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, ctypes as C
import raytracing_amd as rt
from oracle_lib import *
earth=rt.earthmap()
sc,co=rt.Scene.example('final_scene',1,earth); cam=rt.Camera(co)
W=H=800; spp=8
rgb,mean=sc.render(cam,rt.ImageOptions(H,W,spp,50),rt.RenderOptions(seed=1),want_mean=True)
bad=np.argwhere(np.isnan(mean).any(axis=2)); print('nan pixels',len(bad))
lib=C.CDLL(TWIN_PATH)
lib.twin_trace_sample.restype=C.c_uint32
lib.twin_trace_sample.argtypes=[C.c_void_p,C.c_void_p,C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint64,C.c_void_p,C.c_uint32]
desc=sc._host.rth_scene_desc(sc._h)
for (row,col) in bad[:6]:
    i=int(col); j=H-1-int(row)
    for s in range(spp):
        out=np.zeros((60,12),np.float32)
        n=lib.twin_trace_sample(desc,C.byref(cam.abi),W,H,i,j,s,50,1,out.ctypes.data,60)
        if np.isnan(out[:n]).any():
            print('pixel',i,j,'sample',s,'segments',n)
            k=np.argwhere(np.isnan(out[:n]).any(axis=1))[0][0]
            np.set_printoptions(precision=9,suppress=False,linewidth=200)
            for kk in range(max(0,k-2),min(n,k+2)): print('  seg',kk,out[kk])
            break
    else:
        print('pixel',i,j,'twin finds no NaN')

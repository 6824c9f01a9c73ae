import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import raytracing_amd as rt
from oracle_lib import *
earth=rt.earthmap()
for name,spp,W,H in [('cornell',64,64,64),('cornell_smoke',64,64,64),('random_spheres',32,96,54),('perlin',32,96,54),('final_scene',32,64,64)]:
    sc,co=rt.Scene.example(name,1,earth if name=='final_scene' else None)
    osc,oco=OracleScene.example(name,1,earth if name=='final_scene' else None)
    cam=rt.Camera(co)
    io=rt.ImageOptions(H,W,spp,50)
    t0=time.time(); rgb,mean=sc.render(cam,io,rt.RenderOptions(seed=1),want_mean=True); t1=time.time()
    orgb,omean,secs,_=osc.render(oco,W,H,spp,50,seed=1,threads=16)
    orgb2,_,_,_=osc.render(oco,W,H,spp,50,seed=2,threads=16)
    tm=twin_render(sc,cam,W,H,spp,50,seed=1,threads=16)
    print(name,'gpu %.3fs'%(t1-t0),'rmse gpu-vs-oracle %.3f floor %.3f'%(rmse(rgb,orgb),rmse(orgb,orgb2)),'gpu==twin mean bits',np.array_equal(mean,tm),'maxdiff',np.nanmax(np.abs(mean-tm)), 'mean',mean.mean(),omean.mean(),flush=True)
    for flags in (2,4):
        rgb2,mean2=sc.render(cam,io,rt.RenderOptions(seed=1,flags=flags),want_mean=True)
        print('   flags',flags,'identical',np.array_equal(mean2,mean),flush=True)
    rgb3,mean3=sc.render(cam,io,rt.RenderOptions(seed=1,samples_per_pass=5),want_mean=True)
    print('   spp/pass 5 identical',np.array_equal(mean3,mean),flush=True)
    # 2-rank emulation
    parts=[sc.render(cam,io,rt.RenderOptions(seed=1,rank=r,world_size=3),want_mean=True)[1] for r in range(3)]
    print('   3-rank sum identical',np.array_equal(parts[0]+parts[1]+parts[2],mean),flush=True)
# bigger perf probe
sc,co=rt.Scene.example('final_scene',1,earth); cam=rt.Camera(co)
for spp in (16,64):
    io=rt.ImageOptions(800,800,spp,50)
    t0=time.time(); rgb=sc.render(cam,io,rt.RenderOptions(seed=1,flags=1)); t1=time.time()
    print('final_scene 800x800 spp',spp,'%.3fs'%(t1-t0),'Msamples/s %.2f'%(800*800*spp/(t1-t0)/1e6), sc.profile(),flush=True)
sc,co=rt.Scene.example('cornell',1); cam=rt.Camera(co)
io=rt.ImageOptions(600,600,256,50)
t0=time.time(); rgb=sc.render(cam,io,rt.RenderOptions(seed=1)); t1=time.time()
print('cornell 600x600 spp 256 %.3fs Msamples/s %.2f'%(t1-t0,600*600*256/(t1-t0)/1e6),flush=True)

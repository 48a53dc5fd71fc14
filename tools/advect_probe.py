import importlib, numpy as np, sys
sys.path.insert(0,'.')
sfl = importlib.import_module("esp32-fluid-simulation_amd")
import bench
n=8192
with sfl.Solver(n,n) as s:
    j,i=np.mgrid[0:n,0:n].astype(np.float32)
    for name,v in (("noise", bench.synthetic_velocity(n,0,n)),
                   ("vortex", np.stack([-(j-n/2)/(n/2)*100, (i-n/2)/(n/2)*100],axis=-1).astype(np.float32)),
                   ("uniform", np.full((n,n,2), 37.5, np.float32)),
                   ("zero", np.zeros((n,n,2),np.float32))):
        s.upload(sfl.capi.FIELD_VELOCITY, v)
        s.upload(sfl.capi.FIELD_COLOR, bench.synthetic_color(n,0,n))
        dt=np.float32(1/30)
        best=[1e9,1e9]
        for r in range(4):
            s.upload(sfl.capi.FIELD_VELOCITY, v)
            s.timer_start(); s.advect_color(dt, False); t1=s.timer_stop()
            s.timer_start(); s.advect_velocity(dt, True); t0=s.timer_stop()
            best=[min(best[0],t0),min(best[1],t1)]
        print(name, "advect_velocity %.1f us  advect_color %.1f us"%(best[0]*1e3,best[1]*1e3))

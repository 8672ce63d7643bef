import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, radfoam
from radfoam_amd import foam, _lib
import bench as B
dev='cuda:0'
fm=foam.make_synthetic_foam(2000000,2,5,cache_dir=foam.default_cache_dir())
cam=B.orbit_camera(1920,1080,0); rays_np=foam.camera_rays(cam); st=foam.nearest_point(fm['points'],cam['position'])
t=lambda a: torch.from_numpy(a).to(dev)
p,a,adj,off=t(fm['points']),t(fm['attributes']),t(fm['point_adjacency']),t(fm['point_adjacency_offsets'])
rays=t(rays_np); start=torch.full(rays.shape[:-1],st,dtype=torch.int64).to(torch.uint32).to(dev)
g=torch.randn(rays.shape[:-1]+(4,),device=dev)
pipe=radfoam.create_pipeline(2)
out=pipe.trace_forward(p,a,adj,off,rays,start)
# monkeypatch: pass a stats buffer into backward through _launch_opts
stats=torch.zeros(8,dtype=torch.int64,device=dev)
orig=pipe._launch_opts
def lo(*args,**kw):
    o=orig(*args,**kw); o.stats=stats.data_ptr(); return o
pipe._launch_opts=lo
res=pipe.trace_backward(p,a,adj,off,rays,start,out['rgba'],g)
torch.cuda.synchronize()
s=stats.cpu().tolist()
print('row flushes',s[0],'values flushed',s[1],'bypassed lane contributions',s[2],'cached lane contributions',s[3])

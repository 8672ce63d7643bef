import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from tests.conftest import small_foam
from tests import helpers as H
import radfoam
DEV='cuda:0'
d=0
fm=small_foam(6000,d,11)
cam,rays,start=H.camera_setup(fm,16,8)
pipe=radfoam.create_pipeline(d)
p,a,adj,off=H.to_torch_foam(fm,DEV)
dt=pipe.build_adjacent_diff(p,adj,off).cpu().numpy().view(np.uint16)
ref=O.build_adjacent_diff(fm['points'],fm['point_adjacency'],fm['point_adjacency_offsets'])
print('diff table equal:',np.array_equal(dt,ref), (dt!=ref).sum())
r=torch.from_numpy(rays).to(DEV); s=torch.full(r.shape[:-1],int(start),dtype=torch.int64).to(torch.uint32).to(DEV)
args=(d,fm['points'],fm['attributes'],fm['point_adjacency'],fm['point_adjacency_offsets'])
for mi in [0,1,2,3,4,6,10,1024]:
    out=pipe.trace_forward(p,a,adj,off,r,s,max_intersections=mi)
    o=O.trace_forward(*args,rays,start,max_intersections=mi)
    g=out['rgba'].cpu().numpy(); n=out['num_intersections'].cpu().numpy()
    print('mi',mi,'n gpu',n[3,5,0],'n ref',o['num_intersections'][3,5,0],'rgba gpu',g[3,5],'ref',o['rgba'][3,5], 'eq', np.array_equal(g.view(np.uint32),o['rgba'].view(np.uint32)))
# flat mapping
out=pipe.trace_forward(p,a,adj,off,r.reshape(-1,6),s.reshape(-1))
print('flat n', out['num_intersections'].cpu().numpy()[:8,0])
# two-cell slab
pts=np.array([[0,0,-0.5],[0,0,0.5]],dtype=np.float32); adj2=np.array([1,0],dtype=np.uint32); off2=np.array([0,1,2],dtype=np.uint32)
attrs=np.array([[0.2,-0.1,0.3,0.7],[-0.3,0.4,0.1,1.3]],dtype=np.float32)
ray=np.array([[0,0,-2.0,0,0,3.0]],dtype=np.float32)
t=lambda x: torch.from_numpy(x).to(DEV)
out=pipe.trace_forward(t(pts),t(attrs),t(adj2),t(off2),t(ray),torch.zeros(1,dtype=torch.uint32,device=DEV))
o=O.trace_forward(0,pts,attrs,adj2,off2,ray,np.uint32(0))
print('slab gpu',out['rgba'].cpu().numpy(),out['num_intersections'].cpu().numpy(),'ref',o['rgba'],o['num_intersections'])

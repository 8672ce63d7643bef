import sys, time, ctypes as C
sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..', '..')))
import numpy as np
from tests.host_harness import star_host as S
from radfoam_amd import foam
n = int(sys.argv[1]); seed = int(sys.argv[2])
fm = foam.make_synthetic_foam(n, 0, seed, cache_dir=foam.default_cache_dir() if n in (200000, 500000) else None)
pts = fm["points"]; tree = S.aabb_tree(pts); depth = S.pow2_round_up(n).bit_length() - 1
import os
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '_build', 'libsplay_host.so'))
rows = np.zeros((n, 64), dtype=np.uint32); deg = np.zeros(n, dtype=np.uint32); stats = np.zeros(8)
t = time.time()
L.splay_host(C.c_void_p(pts.ctypes.data), C.c_uint32(n), C.c_void_p(tree.ctypes.data), C.c_uint32(depth), C.c_uint32(12),
             C.c_void_p(rows.ctypes.data), C.c_int(64), C.c_void_p(deg.ctypes.data), C.c_void_p(stats.ctypes.data))
dt = time.time() - t
off = np.zeros(n + 1, dtype=np.int64); np.cumsum(deg, out=off[1:])
adj = rows[np.arange(64)[None, :] < deg[:, None]]
ok = np.array_equal(off.astype(np.uint32), fm["point_adjacency_offsets"]) and np.array_equal(adj, fm["point_adjacency"])
print("%.1fs rounds %d inserts/pt %.2f tree nodes/pt %.1f inconsistencies/pt %.2f certifications/pt %.2f cert hits %d inbox lost %d failed %d  equal qhull %s" % ((dt,) + tuple(stats[:5]) + (stats[5], stats[6], stats[7], ok)))
if not ok:
    d0 = np.diff(fm["point_adjacency_offsets"].astype(np.int64)); bad = np.nonzero(d0 != deg)[0]
    print("rows with wrong degree:", bad.size, bad[:10])

"""python run_owner.py N SEED [EXTRA_SEEDS]: the owner-certifies experiment (owner_host.cpp) against Qhull."""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))
from radfoam_amd import foam  # noqa: E402
from tests.host_harness import star_host as S  # noqa: E402

n, seed = int(sys.argv[1]), int(sys.argv[2])
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fm = foam.make_synthetic_foam(n, 0, seed, cache_dir=foam.default_cache_dir() if n in (200000, 500000) else None)
pts = fm["points"]
tree = S.aabb_tree(pts)
depth = S.pow2_round_up(n).bit_length() - 1
L = C.CDLL(os.path.join(HERE, "..", "_build", "libowner_host.so"))
rows = np.zeros((n, 64), dtype=np.uint32)
deg = np.zeros(n, dtype=np.uint32)
stats = np.zeros(8)
t = time.time()
L.owner_host(C.c_void_p(pts.ctypes.data), C.c_uint32(n), C.c_void_p(tree.ctypes.data), C.c_uint32(depth), C.c_uint32(12),
             C.c_uint32(extra), C.c_void_p(rows.ctypes.data), C.c_int(64), C.c_void_p(deg.ctypes.data),
             C.c_void_p(stats.ctypes.data))
dt = time.time() - t
off = np.zeros(n + 1, dtype=np.int64)
np.cumsum(deg, out=off[1:])
adj = rows[np.arange(64)[None, :] < deg[:, None]]
ok = np.array_equal(off.astype(np.uint32), fm["point_adjacency_offsets"]) and np.array_equal(adj, fm["point_adjacency"])
print("%.1fs  tree nodes/pt: pass 1 %.0f + pass 2 %.0f + kNN seeding %.0f = %.0f;  insertions/pt %.2f;  pass 2 per point: "
      "%.1f triangles closed by the owner's certificate, %.1f owner had none, %.1f own;  failed %d;  equals Qhull: %s"
      % (dt, stats[0], stats[1], stats[2], stats[0] + stats[1] + stats[2], stats[3], stats[4], stats[5], stats[6], stats[7], ok))

"""CPU only (scipy / Qhull): two facts about Delaunay stars of a random foam that decide what an incremental rebuild can
reuse and what a range query around a point has to cover.

 1. churn -- how many stars change when every point moves by a fraction of the mean spacing (what an optimiser step does
    between two rebuilds): a star "changes" when the set of tetrahedra at its point does.  Keeping the stars that pass
    their local in-sphere tests and rebuilding the rest needs most stars to survive a move.
 2. reach -- per star, the largest circumradius r_max of its tetrahedra and the number of points within 2 r_max of the
    point (every circumball passes through the point, so it lies inside that ball): the candidates ONE range query per
    star would have to offer to the link instead of one tree walk per triangle (rf_star.hpp: star_sweep).

  python scripts/model_star_churn.py [points]      -> gpurun_out/model_star_churn.json"""
import json
import os
import sys

import numpy as np
from scipy.spatial import Delaunay, cKDTree

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(0)
p = rng.random((n, 3)) * 2 - 1
spacing = (8.0 / n) ** (1 / 3)
d0 = Delaunay(p)
out = {"points": n, "spacing": spacing, "churn": {}}


def tets(d):
    return set(map(tuple, np.sort(d.simplices, axis=1)))


s0 = tets(d0)
for frac in (0.003, 0.01, 0.03, 0.1):
    q = p + frac * spacing * rng.standard_normal((n, 3))
    s1 = tets(Delaunay(q))
    touched = np.zeros(n, bool)
    for t in (s0 - s1) | (s1 - s0):
        touched[list(t)] = True
    out["churn"]["%g" % frac] = dict(tets=len(s0), tets_gone=len(s0 - s1), stars_changed=float(touched.mean()))
    print("move %.3f of the spacing: %.1f %% of the tetrahedra gone, %.1f %% of the stars changed" %
          (frac, 100 * len(s0 - s1) / len(s0), 100 * touched.mean()))

T = d0.simplices
A = p[T[:, 0]]
M = np.stack([p[T[:, 1]] - A, p[T[:, 2]] - A, p[T[:, 3]] - A], axis=1)
rhs = 0.5 * (M * M).sum(2)
c = np.linalg.solve(M, rhs[..., None])[..., 0]
r = np.sqrt((c * c).sum(1))
rmax = np.zeros(n)
for k in range(4):
    np.maximum.at(rmax, T[:, k], r)
inner = (np.abs(p) < 0.8).all(1)
idx = np.where(inner)[0][:20000]
cnt = cKDTree(p).query_ball_point(p[idx], 2 * rmax[idx], return_length=True)
deg = np.diff(d0.vertex_neighbor_vertices[0])
out["reach"] = dict(rmax_over_spacing_mean=float((rmax[inner] / spacing).mean()),
                    rmax_over_spacing_p99=float(np.quantile(rmax[inner] / spacing, 0.99)),
                    points_within_2rmax_mean=float(cnt.mean()), points_within_2rmax_p90=float(np.quantile(cnt, 0.9)),
                    points_within_2rmax_p99=float(np.quantile(cnt, 0.99)), neighbours_mean=float(deg[inner].mean()))
print(json.dumps(out["reach"]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/model_star_churn.json", "w"), indent=1)

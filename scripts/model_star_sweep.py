"""CPU-only: what the sweep of rf_star.hpp (star_sweep) does per star and what a wave of 64 stars in lockstep pays for
it -- rounds, points offered, tree nodes of the walk, and the per-triangle queries left over afterwards -- from the
host build of the very same code (tests/host_harness), on a random foam, from scratch and after a move.
  python scripts/model_star_sweep.py [points]"""
import sys, ctypes as C, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.host_harness import star_host as S
from radfoam_amd import foam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
fm = foam.make_synthetic_foam(n, 0, 11, cache_dir=foam.default_cache_dir())
pts = fm["points"]
depth = S.pow2_round_up(n).bit_length() - 1
L = S.lib()
L.star_host_sweep_trace.restype = C.c_int
first, count = 64 * 1000, 64 * 200
out = {}


def run(points, old):
    tree = S.aabb_tree(points)
    rec = np.zeros((count, 8), dtype=np.uint32)
    oa = oo = None
    if old is not None:
        oo, oa = np.ascontiguousarray(old[0], dtype=np.uint32), np.ascontiguousarray(old[1], dtype=np.uint32)
    L.star_host_sweep_trace(C.c_void_p(points.ctypes.data), C.c_uint32(n), C.c_void_p(tree.ctypes.data), C.c_uint32(depth),
                            C.c_uint32(12), C.c_uint32(512), None if oa is None else C.c_void_p(oa.ctypes.data),
                            None if oo is None else C.c_void_p(oo.ctypes.data), C.c_uint32(first), C.c_uint32(count),
                            C.c_void_p(rec.ctypes.data))
    return rec.astype(np.int64)


def report(name, rec):
    W = rec.reshape(-1, 64, 8)
    how, offered, rounds, nodes, lq, lnodes, lmax, ins = [W[:, :, k] for k in range(8)]
    line = dict(
        stars=int(rec.shape[0]), swept=float((how == 0).mean()), out_of_budget=float((how == 3).mean()),
        offered_mean=float(offered.mean()), offered_wave_max=float(offered.max(1).mean()),
        rounds_mean=float(rounds.mean()), rounds_wave_max=float(rounds.max(1).mean()),
        walk_nodes_mean=float(nodes.mean()), walk_nodes_wave_max=float(nodes.max(1).mean()),
        stars_with_leftover_queries=float((lq > 0).mean()), waves_with_leftover_queries=float((lq > 0).any(1).mean()),
        leftover_queries_mean=float(lq.mean()), leftover_queries_wave_max=float(lq.max(1).mean()),
        leftover_nodes_mean=float(lnodes.mean()), leftover_nodes_wave_max=float(lnodes.max(1).mean()),
        insertions_mean=float(ins.mean()))
    out[name] = line
    print(name, json.dumps(line))


S.lib().star_host_set_sweep(1)
report("from_scratch", run(pts, None))
rng = np.random.default_rng(1)
moved = (pts + rng.normal(0, 0.03 * (8.0 / n) ** (1 / 3), size=pts.shape)).astype(np.float32)
report("incremental_3pct", run(moved, (fm["point_adjacency_offsets"], fm["point_adjacency"])))
S.lib().star_host_set_sweep(0)
report("incremental_3pct_no_sweep", run(moved, (fm["point_adjacency_offsets"], fm["point_adjacency"])))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/model_star_sweep.json", "w"), indent=1)

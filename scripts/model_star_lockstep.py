"""CPU-only model of the first pass of the GPU triangulation: per-query tree-node counts of 12,800
stars from the host build of rf_star.hpp (tests/host_harness), and what a wave that runs 64 of them in lockstep --
query k of every lane at the same time, as delaunay_star_kernel does -- pays for it.
  python scripts/model_star_lockstep.py"""
import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
from tests.host_harness import star_host as S
from radfoam_amd import foam
n = 200000
fm = foam.make_synthetic_foam(n, 0, 11, cache_dir=foam.default_cache_dir())
pts = fm["points"]; tree = S.aabb_tree(pts); depth = S.pow2_round_up(n).bit_length() - 1
L = S.lib(); L.star_host_trace.restype = C.c_int; L.star_host_set_sweep(0)   # the kernels compile the sweep out
first, count, cap = 64 * 1000, 64 * 200, 160
out = np.zeros((count, cap), dtype=np.uint32); lens = np.zeros(count, dtype=np.uint32)
L.star_host_trace(C.c_void_p(pts.ctypes.data), C.c_uint32(n), C.c_void_p(tree.ctypes.data), C.c_uint32(depth), C.c_uint32(12), C.c_uint32(512),
                  C.c_uint32(first), C.c_uint32(count), C.c_uint32(cap), C.c_void_p(out.ctypes.data), C.c_void_p(lens.ctypes.data))
nodes = (out & 0x7FFFFFFF).astype(np.int64); found = (out >> 31).astype(bool)
mask = np.arange(cap)[None, :] < lens[:, None]
print("queries/star %.1f (found %.1f) nodes/query mean %.1f median %.0f p90 %.0f p99 %.0f max %d" % (lens.mean(), (found & mask).sum(1).mean(), nodes[mask].mean(), np.median(nodes[mask]), np.percentile(nodes[mask], 90), np.percentile(nodes[mask], 99), nodes[mask].max()))
W = nodes.reshape(-1, 64, cap); M = mask.reshape(-1, 64, cap)
per_lane_total = (W * M).sum(2)                      # nodes per star
lockstep = (W * M).max(1).sum(1)                     # sum over query index of the max over lanes
ideal = per_lane_total.max(1)
print("per wave: mean lane total %.0f, max lane total %.0f, lockstep (query-synchronous) %.0f -> efficiency mean/lockstep %.2f, mean/maxlane %.2f" % (per_lane_total.mean(), ideal.mean(), lockstep.mean(), per_lane_total.mean() / lockstep.mean(), per_lane_total.mean() / ideal.mean()))
# `width` lanes per star (a wave handles 64/width stars at a time, 64 stars take `width` rounds), a query of L nodes
# taking about L / eff + 2 steps of the sub-wave
for width, eff in ((8, 3.0), (16, 4.5)):
    steps = np.ceil(nodes / eff) + 2
    S_ = (steps * mask).reshape(-1, width, 64 // width, cap)    # [wave, round, concurrent star, query]
    cost = S_.max(2).sum(2).sum(1)                              # per round lockstep over concurrent stars, rounds add
    print("sub-wave width %d (assumed %.1f nodes per step): steps per 64 stars %.0f (now %.0f)" % (width, eff, cost.mean(), lockstep.mean()))

# ---- persistent waves: a lane that finishes its star gets the next point of the wave's queue; refills happen when at
# least K lanes are idle (seeding a star is a long stretch only the refilled lanes execute), SEED node-steps each.
def persistent(K, SEED=400, chunk=1024):
    """steps a wave needs for `chunk` stars taken from a queue, per 64 stars"""
    total = 0.0
    stars = [(nodes[k][: lens[k]]) for k in range(count)]
    for w0 in range(0, count - chunk + 1, chunk):
        queue = list(range(w0, w0 + chunk))
        lane_star = [None] * 64          # remaining query lengths of the lane's star
        steps = 0.0
        while True:
            idle = [l for l in range(64) if lane_star[l] is None or len(lane_star[l]) == 0]
            if queue and (len(idle) >= K or len(idle) == 64):
                for l in idle:
                    if not queue:
                        break
                    lane_star[l] = list(stars[queue.pop()])
                steps += SEED
            active = [l for l in range(64) if lane_star[l]]
            if not active:
                if not queue:
                    break
                continue
            steps += max(lane_star[l][0] for l in active)     # one query of every active lane, in lockstep
            for l in active:
                lane_star[l].pop(0)
        total += steps
    return total / ((count // chunk) * chunk / 64.0)


print("today (one star per lane, query-synchronous): %.0f node-steps per 64 stars" % lockstep.mean())
for K in (1, 8, 16, 32):
    print("persistent waves, refill when >= %2d lanes are idle (400 node-steps per refill): %.0f" % (K, persistent(K)))

# ---- a budget of tree nodes per STAR: a lane whose star has used it up stops, its star is redone in a second launch of the
# same kernel over the list of such stars (waves of 64 stars that are all expensive: alike, so lockstep costs little there)
def with_star_budget(cap):
    cum = np.cumsum(nodes * mask, axis=1)
    keep = mask & (cum - nodes * mask < cap)            # queries that start within the budget run (the last one to its end)
    Wk = (nodes * keep).reshape(-1, 64, nodes.shape[1])
    first = Wk.max(1).sum(1).mean()
    over = np.where((nodes * mask).sum(1) > cap)[0]
    again = 0.0
    if over.size:
        pad = (-over.size) % 64
        idx = np.concatenate([over, over[:pad]]) if pad else over
        again = (nodes[idx] * mask[idx]).reshape(-1, 64, nodes.shape[1]).max(1).sum(1).sum() / (count / 64.0)
    return first, again, over.size / count


for cap in (1500, 2000, 2500, 3000, 4000, 1 << 30):
    f, a, frac = with_star_budget(cap)
    print("star budget %10d nodes: first launch %.0f + second launch %.0f = %.0f node-steps per 64 stars (%.1f %% of the stars redone)" % (cap, f, a, f + a, 100 * frac))

# ---- lanes grouped by cost: within chunks of `chunk` consecutive stars (a compact region of the kd-order: what a block's
# waves share of the tree stays shared) the stars are dealt to the waves in the order of a cost proxy known before the
# build -- the number of queries (about twice the previous neighbour count on an incremental rebuild)
def grouped(chunk, key):
    total = 0.0
    for c0 in range(0, count - chunk + 1, chunk):
        idx = np.arange(c0, c0 + chunk)
        idx = idx[np.argsort(key[idx], kind="stable")]
        Wk = (nodes[idx] * mask[idx]).reshape(-1, 64, nodes.shape[1])
        total += Wk.max(1).sum(1).sum()
    return total / ((count // chunk) * chunk / 64.0)


nq = mask.sum(1)
tot = (nodes * mask).sum(1)
for chunk in (256, 1024, 4096):
    print("chunks of %4d stars, waves dealt by the number of queries: %.0f node-steps per 64 stars; by the star's total nodes (an oracle): %.0f (now %.0f)" %
          (chunk, grouped(chunk, nq), grouped(chunk, tot), lockstep.mean()))

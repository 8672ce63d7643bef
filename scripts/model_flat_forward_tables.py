"""CPU model of LDS tables for the flat-batch forward (forward_mode 5 and what could follow it), on sampled 256-slot groups of
the sorted training batch (cell sequences from the oracle, lockstep as scripts/model_train_batch.py):

  cell_direct / cell_2way   lane-level hit rates of the block's cell table, direct-mapped (shipped: 304 / 232 entries) and 2-way
  link_direct[S]            a table keyed by (cell, exit neighbour) holding the link: hits = hops that need no link read
  last-exit prediction      the cell entry remembers the neighbour its last visitor left through: hops predicted for free

  python scripts/model_flat_forward_tables.py        (about a minute; needs the cached 2 M-point foam)
Round 4 (profiles/HISTORY.md): cell 0.64 / 0.69 direct at 232 / 304, 0.69 / 0.75 two-way; link 0.42 / 0.47 / 0.51 / 0.59 at 256 /
384 / 512 / 1024 entries; last exit 0.34 of the hops (0.49 of the cell hits)."""

import sys, os, collections, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import model_train_batch as M
from oracle import oracle as O
from radfoam_amd import foam
import bench
fm = foam.make_synthetic_foam(2_000_000, 3, 5, cache_dir=foam.default_cache_dir())
rays, start = bench.training_batch(fm, 1_000_000, 105)
order = M.ray_order(rays, start)
nblocks = order.size // 256
rng = np.random.default_rng(0)
pick = np.sort(rng.choice(nblocks, size=30, replace=False))
slots = (pick[:, None] * 256 + np.arange(256)[None, :]).reshape(-1)
rr = order[slots]
cells, t1, n = O.trace_paths(3, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"], rays[rr], start[rr], cap=384)
n = np.minimum(n, 384).astype(np.int64)
LS = (256, 384, 512, 1024)
CS = (232, 304)
tot = collections.Counter()
def h(x, k):
    return (x * 2654435761 % (1 << 32)) * k >> 32
for b in range(pick.size):
    c = cells[b*256:(b+1)*256].astype(np.int64); nn = n[b*256:(b+1)*256]
    pos = np.zeros(256, dtype=np.int64)
    ldm = {k: np.full(k, -1, dtype=np.int64) for k in LS}
    cdm = {k: np.full(k, -1, dtype=np.int64) for k in CS}
    c2 = {k: (np.full((k//2, 2), -1, dtype=np.int64), np.zeros(k//2, dtype=np.int64)) for k in CS}
    alive = pos < nn
    while alive.any():
        for w in range(4):
            lanes = np.arange(w*64, (w+1)*64)
            idx = lanes[alive[lanes]]
            if idx.size == 0: continue
            cur = c[idx, pos[idx]]
            has_next = pos[idx] + 1 < nn[idx]
            nxt = np.where(has_next, c[idx, np.minimum(pos[idx] + 1, 383)], -1)
            tot["lane_visits"] += idx.size
            tot["lane_hops"] += int(has_next.sum())
            # cells: per lane hit/miss (lanes with the same cell in a wave-step: first misses, rest ride along -> count lane-level)
            for k in CS:
                t = cdm[k]; sl = h(cur, k)
                hit = t[sl] == cur
                tot[f"cell_direct{k}_lane_hits"] += int(hit.sum())
                t[sl] = cur
                ways, mru = c2[k]; s2 = h(cur, k//2)
                hit2 = (ways[s2, 0] == cur) | (ways[s2, 1] == cur)
                tot[f"cell_2way{k}_lane_hits"] += int(hit2.sum())
                for cc, ss in zip(cur.tolist(), s2.tolist()):
                    if ways[ss, 0] == cc: mru[ss] = 0
                    elif ways[ss, 1] == cc: mru[ss] = 1
                    else:
                        v = 1 - mru[ss]; ways[ss, v] = cc; mru[ss] = v
            key = cur[has_next] * (1 << 22) + (nxt[has_next] & ((1 << 22) - 1))
            for k in LS:
                t = ldm[k]; sl = h(key % (1 << 32), k)
                hit = t[sl] == key
                tot[f"link_direct{k}_lane_hits"] += int(hit.sum())
                t[sl] = key
            pos[idx] += 1
        alive = pos < nn
r = dict(tot)
for k, v in list(r.items()):
    if k.endswith("lane_hits"):
        r[k.replace("lane_hits", "hit_rate")] = round(v / (r["lane_hops"] if k.startswith("link") else r["lane_visits"]), 3)
print(json.dumps(r, indent=1))

# last-exit prediction inside the cell table (304 entries): a lane that hits the entry of its cell finds the neighbour its
# predecessor left through; predicted = same neighbour
tot2 = collections.Counter()
for b in range(pick.size):
    c = cells[b*256:(b+1)*256].astype(np.int64); nn = n[b*256:(b+1)*256]
    pos = np.zeros(256, dtype=np.int64)
    K = 304
    tab = np.full(K, -1, dtype=np.int64); last = np.full(K, -1, dtype=np.int64)
    alive = pos < nn
    while alive.any():
        for w in range(4):
            lanes = np.arange(w*64, (w+1)*64)
            idx = lanes[alive[lanes]]
            if idx.size == 0: continue
            cur = c[idx, pos[idx]]
            has_next = pos[idx] + 1 < nn[idx]
            nxt = np.where(has_next, c[idx, np.minimum(pos[idx] + 1, 383)], -1)
            sl = h(cur, K)
            hit = tab[sl] == cur                      # all lanes probe before anyone fills (as the kernel does)
            pred = hit & (last[sl] == nxt) & has_next
            tot2["hops"] += int(has_next.sum()); tot2["cell_hits"] += int((hit & has_next).sum()); tot2["predicted"] += int(pred.sum())
            tab[sl] = cur; last[sl] = nxt
            pos[idx] += 1
        alive = pos < nn
print("last-exit prediction:", dict(tot2), "cell hit rate", round(tot2["cell_hits"]/tot2["hops"], 3), "predicted", round(tot2["predicted"]/tot2["hops"], 3))

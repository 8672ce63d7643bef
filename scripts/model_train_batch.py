"""CPU model of how the training-shaped batch (bench.py --workload train-batch: 1 M shuffled rays of 8 cameras, 2 M-point
foam) maps onto waves: per-ray cell sequences from the oracle (rfo_trace_paths), the kernels' slot order
(rf_build_ray_order: entry cell, then Morton code of the octahedral direction), and a lockstep simulation of sampled
256-slot blocks under scheduling policies.  What it reports per policy:

  wave_steps          issue-time proxy: steps a wave needs until its last lane is done
  lane_visits         (ray, cell) scans = what the lanes fetch when nothing is shared
  wave_distinct       sum over wave-steps of the distinct cells the wave's active lanes sit in = fetches when only lanes
                      that hit the same cell in the same instruction share (what the hardware coalesces)
  block_lru[K]        misses of a K-entry LRU of cell ids per block = fetches with a block-level software cache in LDS
  block_direct[S]     the same for a direct-mapped table of S entries (what an LDS table without bookkeeping can be)

  python scripts/model_train_batch.py [--blocks 120] [--points 2000000] [--rays 1000000]
"""
import argparse
import collections
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from radfoam_amd import foam  # noqa: E402
import bench  # noqa: E402


def spread16(v):
    v = v.astype(np.uint64)
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def ray_order(rays, start):
    """rf_build_ray_order's key (radfoam_amd/csrc/rf_adjacency.hip: ray_keys_kernel), in numpy."""
    d = rays[:, 3:6].astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    s = np.abs(d).sum(axis=1)
    px, py = d[:, 0] / s, d[:, 1] / s
    neg = d[:, 2] < 0
    ox = (1 - np.abs(py)) * np.where(px >= 0, 1.0, -1.0)
    oy = (1 - np.abs(px)) * np.where(py >= 0, 1.0, -1.0)
    px, py = np.where(neg, ox, px), np.where(neg, oy, py)
    u = np.clip((px * 0.5 + 0.5) * 65535.0, 0, 65535).astype(np.uint32)
    v = np.clip((py * 0.5 + 0.5) * 65535.0, 0, 65535).astype(np.uint32)
    key = (start.astype(np.uint64) << np.uint64(32)) | spread16(u) | (spread16(v) << np.uint64(1))
    return np.argsort(key, kind="stable")


def simulate_writeback(cells, n, sizes=(512, 1024, 2048, 4096), epoch_rows=768, epoch=4):
    """Global atomics per segment for the density gradient (one value per segment) under block-level combining:
    the shipped table (epoch_rows entries, 8 probes, entries untouched for `epoch` steps flushed) against a direct-mapped
    write-back cache of S entries (an entry goes to memory only when another cell takes its slot, or at the end)."""
    pos = np.zeros(256, dtype=np.int64)
    alive = pos < n
    dm = {s: np.full(s, -1, dtype=np.int64) for s in sizes}
    dm_flush = {s: 0 for s in sizes}
    table, touched = {}, set()
    ep_flush = ep_bypass = 0
    step = 0
    segs = 0
    while alive.any():
        idx = np.nonzero(alive)[0]
        cur = cells[idx, pos[idx]].astype(np.int64)
        segs += idx.size
        # lanes of a wave in the same cell are merged in registers first (DPP stages): one request per (wave, cell)
        req = np.unique((idx // 64) * (1 << 32) + cur) & 0xFFFFFFFF
        for s in sizes:
            h = (req * 2654435761 % (1 << 32)) * s >> 32
            t = dm[s]
            for c, slot in zip(req.tolist(), h.tolist()):
                if t[slot] != c:
                    if t[slot] >= 0:
                        dm_flush[s] += 1
                    t[slot] = c
        for c in req.tolist():
            if c in table or len(table) < epoch_rows:
                table[c] = True
                touched.add(c)
            else:
                ep_bypass += 1
        pos[idx] += 1
        alive = pos < n
        step += 1
        if step % epoch == 0:
            for c in [c for c in table if c not in touched]:
                del table[c]
                ep_flush += 1
            touched = set()
    ep_flush += len(table)
    out = {"segments": segs, "epoch_table_atomics": ep_flush + ep_bypass}
    for s in sizes:
        out[f"writeback{s}_atomics"] = dm_flush[s] + int((dm[s] >= 0).sum())
    return out


def simulate_row_cache(cells, n, lit):
    """Colour-row + point-gradient atomics per LIT segment (what backward mode 4 sends straight to memory, one row per
    segment) if rows were first combined in LDS: a block-level LRU of K rows against per-wave direct-mapped tables of S
    rows (no cross-wave sharing, but no LDS atomics needed: after the in-register merge every active lane of a wave has
    its own cell)."""
    pos = np.zeros(256, dtype=np.int64)
    alive = pos < n
    ks, ss = (64, 128, 256), (8, 16, 32, 64, 128)
    lru = {k: collections.OrderedDict() for k in ks}
    miss = {k: 0 for k in ks}
    dm = {s: [np.full(s, -1, dtype=np.int64) for _ in range(4)] for s in ss}
    dmiss = {s: 0 for s in ss}
    segs = merged = 0
    while alive.any():
        idx = np.nonzero(alive)[0]
        cur = cells[idx, pos[idx]].astype(np.int64)
        keep = lit[cur]
        idx, cur = idx[keep], cur[keep]
        segs += idx.size
        wv = idx // 64
        for w in range(4):
            req = np.unique(cur[wv == w])
            merged += req.size
            for k in ks:
                l = lru[k]
                for c in req.tolist():
                    if c in l:
                        l.move_to_end(c)
                    else:
                        miss[k] += 1
                        l[c] = True
                        if len(l) > k:
                            l.popitem(last=False)
            for s in ss:
                t = dm[s][w]
                h = (req * 2654435761 % (1 << 32)) * s >> 32
                for c, slot in zip(req.tolist(), h.tolist()):
                    if t[slot] != c:
                        dmiss[s] += 1
                        t[slot] = c
        pos[np.nonzero(alive)[0]] += 1
        alive = pos < n
    out = {"lit_segments": segs, "rows_after_in_wave_merge": merged}
    out.update({f"block_lru{k}_rows": miss[k] for k in ks})
    out.update({f"per_wave_direct{s}_rows": dmiss[s] for s in ss})
    return out


DIRECT_SIZES = (128, 256, 384, 512, 768, 1024)


def simulate_block(cells, t1, n, policy, delta, lru_sizes):
    """cells/t1: [256, cap]; n: [256] scans per ray.  Lockstep per wave of 64 lanes; the four waves of a block advance
    round-robin one wave-step at a time (the LRU sees their interleaved fetches)."""
    waves = [np.arange(w * 64, (w + 1) * 64) for w in range(4)]
    pos = np.zeros(256, dtype=np.int64)       # next scan index per lane
    t0 = np.zeros(256, dtype=np.float64)
    steps = [0, 0, 0, 0]
    lane_visits = 0
    wave_distinct = 0
    lrus = {k: collections.OrderedDict() for k in lru_sizes}
    miss = {k: 0 for k in lru_sizes}
    dms = {k: np.full(k, -1, dtype=np.int64) for k in DIRECT_SIZES}     # direct-mapped: what an LDS table can be
    dmiss = {k: 0 for k in DIRECT_SIZES}
    alive = pos < n
    while alive.any():
        for w, lanes in enumerate(waves):
            a = alive[lanes]
            if not a.any():
                continue
            act = a.copy()
            if policy == "sync_t":      # only the lanes within delta of the wave's laggard advance
                tmin = t0[lanes][a].min()
                act &= t0[lanes] <= tmin + delta
            idx = lanes[act]
            cur = cells[idx, pos[idx]]
            steps[w] += 1
            lane_visits += idx.size
            uniq = np.unique(cur)
            wave_distinct += uniq.size
            for k in lru_sizes:
                l = lrus[k]
                for c in uniq.tolist():
                    if c in l:
                        l.move_to_end(c)
                    else:
                        miss[k] += 1
                        l[c] = True
                        if len(l) > k:
                            l.popitem(last=False)
            for k in DIRECT_SIZES:
                t = dms[k]
                h = (uniq.astype(np.int64) * 2654435761 % (1 << 32)) * k >> 32
                for c, sl in zip(uniq.tolist(), h.tolist()):
                    if t[sl] != c:
                        dmiss[k] += 1
                        t[sl] = c
            tt = t1[idx, pos[idx]]
            t0[idx] = np.maximum(t0[idx], np.where(np.isfinite(tt), tt, t0[idx]))
            pos[idx] += 1
        alive = pos < n
    return dict(wave_steps=sum(steps), lane_visits=lane_visits, wave_distinct=wave_distinct,
                **{f"block_lru{k}": miss[k] for k in lru_sizes}, **{f"block_direct{k}": dmiss[k] for k in DIRECT_SIZES})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=120)
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--sh", type=int, default=3)
    ap.add_argument("--cap", type=int, default=384)
    ap.add_argument("--image", action="store_true", help="a 1080p frame in 16x16 tiles instead (the dense reference point)")
    ap.add_argument("--all-lit", action="store_true",
                    help="row combining with every segment lit (real training: softplus never returns exactly 0)")
    args = ap.parse_args()
    fm = foam.make_synthetic_foam(args.points, args.sh, args.seed, cache_dir=foam.default_cache_dir())
    if args.image:
        cam = foam.default_camera(1920, 1080)
        r = foam.camera_rays(cam)
        st = foam.nearest_point(fm["points"], cam["position"])
        # slot order of the image path: 16x16 tiles, four 8x8 waves each, Z-order inside a wave
        ty, tx = np.meshgrid(np.arange(0, 1072, 16), np.arange(0, 1920, 16), indexing="ij")
        lane = np.arange(64)
        lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4)
        ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4)
        order = []
        for y0, x0 in zip(ty.reshape(-1), tx.reshape(-1)):
            for w in range(4):
                order.append((y0 + (w >> 1) * 8 + ly) * 1920 + x0 + (w & 1) * 8 + lx)
        order = np.concatenate(order)
        rays, start = r.reshape(-1, 6), np.full(r.shape[0] * r.shape[1], st, dtype=np.uint32)
    else:
        rays, start = bench.training_batch(fm, args.rays, args.seed + 100)
        order = ray_order(rays, start)
    nblocks = order.size // 256
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(nblocks, size=min(args.blocks, nblocks), replace=False))
    slots = (pick[:, None] * 256 + np.arange(256)[None, :]).reshape(-1)
    rr = order[slots]
    cells, t1, n = O.trace_paths(args.sh, fm["points"], fm["attributes"], fm["point_adjacency"],
                                 fm["point_adjacency_offsets"], rays[rr], start[rr], cap=args.cap)
    n = np.minimum(n, args.cap).astype(np.int64)
    dens = fm["attributes"][:, -1]
    spacing = (8.0 / args.points) ** (1.0 / 3.0)
    result = {"points": args.points, "blocks_sampled": int(pick.size), "mean_scans_per_ray": float(n.mean()),
              "cell_spacing": spacing, "workload": "1080p frame, 16x16 tiles" if args.image else "train-batch"}
    lru_sizes = (64, 128, 256, 512)
    policies = [("lockstep", 0.0), ("sync_t", 0.5 * spacing), ("sync_t", 1.0 * spacing), ("sync_t", 2.0 * spacing),
                ("sync_t", 4.0 * spacing)]
    for pol, delta in policies:
        tot = collections.Counter()
        for b in range(pick.size):
            sl = slice(b * 256, (b + 1) * 256)
            tot.update(simulate_block(cells[sl], t1[sl], n[sl], pol, delta, lru_sizes))
        name = pol if pol == "lockstep" else f"sync_t(delta={delta / spacing:.1f} cells)"
        rec = dict(tot)
        rec["lane_utilisation"] = round(rec["lane_visits"] / (64.0 * rec["wave_steps"]), 3)
        rec["fetches_per_visit_wave_coalesced"] = round(rec["wave_distinct"] / rec["lane_visits"], 3)
        for k in lru_sizes:
            rec[f"fetches_per_visit_block_lru{k}"] = round(rec[f"block_lru{k}"] / rec["lane_visits"], 3)
        for k in DIRECT_SIZES:
            rec[f"fetches_per_visit_block_direct{k}"] = round(rec[f"block_direct{k}"] / rec["lane_visits"], 3)
        result[name] = rec
        print(name, json.dumps(rec))
    wb = collections.Counter()
    for b in range(pick.size):
        sl = slice(b * 256, (b + 1) * 256)
        wb.update(simulate_writeback(cells[sl], n[sl]))
    wb = dict(wb)
    for k in list(wb):
        if k != "segments":
            wb[k + "_per_segment"] = round(wb[k] / wb["segments"], 3)
    result["density_gradient_combining"] = wb
    print("density gradient combining", json.dumps(wb))
    rc = collections.Counter()
    lit_mask = np.concatenate([(dens > 1e-6) | args.all_lit, [False]])   # index N (none) never lit
    safe = np.where(cells == O.NONE, args.points, cells).astype(np.int64)
    for b in range(pick.size):
        sl = slice(b * 256, (b + 1) * 256)
        rc.update(simulate_row_cache(safe[sl], n[sl], lit_mask))
    rc = dict(rc)
    for k in list(rc):
        if k != "lit_segments":
            rc[k + "_per_lit_segment"] = round(rc[k] / max(rc["lit_segments"], 1), 3)
    result["row_combining"] = rc
    print("row combining", json.dumps(rc))
    # how much the visits of one block repeat at all (the ceiling of any block-level reuse)
    distinct_per_block = [np.unique(cells[b * 256:(b + 1) * 256][cells[b * 256:(b + 1) * 256] != O.NONE]).size
                          for b in range(pick.size)]
    result["distinct_cells_per_block_over_visits"] = round(float(np.sum(distinct_per_block)) / float(n.sum()), 3)
    lit = dens[cells[cells != O.NONE]] > 1e-6
    result["lit_fraction_of_visits"] = round(float(lit.mean()), 3)
    print(json.dumps({k: v for k, v in result.items() if not isinstance(v, dict)}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(result, open(os.path.join(ROOT, "gpurun_out", "model_train_batch.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

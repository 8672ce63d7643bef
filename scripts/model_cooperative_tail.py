"""CPU model of a COOPERATIVE TAIL for the face scan (VERDICT r5 next #6b): today a wave's block loop runs to its longest
face list (66 % of lane-block slots useful); lanes whose list has ended could take blocks of the lanes that are not done.

Input: the exact per-ray cell sequences of a 1080p frame of the north-star foam from the oracle (rfo_trace_paths), the
image path's slot order (16x16 tiles, four 8x8 waves, Z-order lanes), the padded block count of every cell.  For every
wave-step the model has the block count of each live lane and prices three schedules in VALU wave-instructions:

  today        max over the lanes of the block count x SCAN (67 VALU per block of four faces: profiles/isa_constants.json)
  cooperative  the first P blocks as today (P = 0 .. 6, every lane on its own list), then the leftover blocks of all lanes
               dealt to all 64 lanes: ceil(leftover / 64) rounds x (SCAN + HANDOVER) + SETUP + merge rounds x MERGE, where a
               merge round brings one helper's partial winner back to the owner (rounds = the largest number of leftover
               blocks any one lane has); HANDOVER / SETUP / MERGE are the VALU the scheme adds (its 10 + 4-per-round
               ds_bpermute go to the LDS pipe, which the image forward does not use: counted separately, not priced)
  ideal        ceil(sum of blocks / 64) x SCAN: perfect balance, no overhead (the bound of ANY rebalancing)

plus the hop + composite cost per wave-step that no scan schedule touches (153 VALU) -- so the last column is the share
of the forward's VALU stream each schedule would save.
  python scripts/model_cooperative_tail.py [--tiles 400] [--setup 20 --handover 6 --merge 12]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from radfoam_amd import foam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--tiles", type=int, default=400, help="16x16 tiles sampled from the frame")
    ap.add_argument("--scan", type=int, default=67)
    ap.add_argument("--other", type=int, default=153, help="hop + composite VALU per wave-step")
    ap.add_argument("--setup", type=int, default=20, help="VALU to list the leftover blocks (ballots, prefix counts)")
    ap.add_argument("--handover", type=int, default=6, help="VALU per cooperative round around the 10 ds_bpermute of a hand-over")
    ap.add_argument("--merge", type=int, default=12, help="VALU per merge round (certified cross-compare + select)")
    ap.add_argument("--cap", type=int, default=320)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "model_cooperative_tail.json"))
    args = ap.parse_args()
    fm = foam.make_synthetic_foam(args.points, 2, args.seed, cache_dir=foam.default_cache_dir())
    W, H = 1920, 1080
    cam = foam.default_camera(W, H)
    rays = foam.camera_rays(cam).reshape(-1, 6)
    st = foam.nearest_point(fm["points"], cam["position"])
    off = fm["point_adjacency_offsets"].astype(np.int64)
    blocks_of_cell = ((off[1:] - off[:-1] + 3) // 4).astype(np.int64)
    tiles_x, tiles_y = W // 16, (H + 15) // 16
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(tiles_x * (H // 16), size=min(args.tiles, tiles_x * (H // 16)), replace=False))
    lane = np.arange(64)
    lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4)
    ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4)
    idx = []
    for t in pick:
        ty, tx = divmod(int(t), tiles_x)
        for w in range(4):
            idx.append((ty * 16 + (w >> 1) * 8 + ly) * W + tx * 16 + (w & 1) * 8 + lx)
    idx = np.concatenate(idx)
    cells, _t1, n = O.trace_paths(2, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"],
                                  rays[idx], np.full(idx.size, st, dtype=np.uint32), cap=args.cap)
    n = np.minimum(n, args.cap).astype(np.int64)
    nw = idx.size // 64
    cells = cells.reshape(nw, 64, args.cap)
    n = n.reshape(nw, 64)
    prefix = list(range(0, 7))
    tot = {"today": 0, "ideal": 0, **{f"coop{p}": 0 for p in prefix}}
    lds_ops = {f"coop{p}": 0 for p in prefix}
    wave_steps = 0
    slots = useful = 0
    for h in range(int(n.max())):
        live = n > h                                            # [waves, 64]
        any_live = live.any(axis=1)
        if not any_live.any():
            break
        c = np.where(live, cells[:, :, h], 0).astype(np.int64)
        b = np.where(live, blocks_of_cell[np.minimum(c, blocks_of_cell.size - 1)], 0)[any_live]     # blocks per lane
        wave_steps += b.shape[0]
        mx = b.max(axis=1)
        sm = b.sum(axis=1)
        tot["today"] += int((mx * args.scan).sum())
        tot["ideal"] += int((-(-sm // 64) * args.scan).sum())
        slots += int((mx * 64).sum())
        useful += int(sm.sum())
        for p in prefix:
            left = np.maximum(b - p, 0)
            head = np.minimum(mx, p)
            lsum = left.sum(axis=1)
            lmax = left.max(axis=1)
            rounds = -(-lsum // 64)
            coop = head * args.scan + np.where(lsum > 0, args.setup + rounds * (args.scan + args.handover) + lmax * args.merge, 0)
            # a wave takes the cheaper of the two paths it could take at this step (the choice is one scalar compare)
            tot[f"coop{p}"] += int(np.minimum(coop, mx * args.scan).sum())
            lds_ops[f"coop{p}"] += int(np.where((lsum > 0) & (coop < mx * args.scan), rounds * 10 + lmax * 4, 0).sum())
    other = wave_steps * args.other
    res = {"tiles": int(pick.size), "wave_steps": wave_steps, "lane_block_slot_utilisation": round(useful / slots, 4),
           "valu_per_wave_step_today": round((tot["today"] + other) / wave_steps, 1), "assumptions": vars(args), "schedules": {}}
    for k, v in tot.items():
        res["schedules"][k] = {"scan_valu_per_wave_step": round(v / wave_steps, 1),
                               "forward_valu_saved_frac": round((tot["today"] - v) / (tot["today"] + other), 4)}
        if k in lds_ops:
            res["schedules"][k]["ds_bpermute_per_wave_step"] = round(lds_ops[k] / wave_steps, 2)
        print(k, json.dumps(res["schedules"][k]), flush=True)
    print(json.dumps({k: v for k, v in res.items() if k not in ("schedules", "assumptions")}))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1, default=str)


if __name__ == "__main__":
    main()

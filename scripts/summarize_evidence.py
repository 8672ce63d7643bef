"""Digest of a scripts/gpu_evidence.sh run: per-kernel averages of every PMC counter (-> <dir>/counters_<workload>.json),
and a one-screen summary of the bench lines.  usage: python scripts/summarize_evidence.py gpurun_out/ev"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]


def short(name):
    m = re.search(r"rf::(\w+)(<[^>]*>)?", name)
    base, targs = m.group(1), (m.group(2) or "")
    # the statistics instance of the forward kernel (<DEG, HALF, BENCH, QUANT, STATS, EAGER>: STATS = true) is not a
    # bench kernel
    args = [a.strip() for a in targs.strip("<>").split(",")]
    if base == "forward_kernel" and len(args) >= 5 and args[4] == "true":
        base = "forward_kernel[stats]"
    return base


for pm in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    w = os.path.basename(pm)[4:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(pm, "p*", "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "rf::" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, cs in agg.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches_averaged"] = max(len(v) for v in cs.values())
        e["duration_ns"] = sum(dur[k]) / len(dur[k])
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            f_kb, w_kb = e.get("FETCH_SIZE", 0.0), e.get("WRITE_SIZE", 0.0)
            # counters are in KB; gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md, HBM): the
            # x2 figure is the upper estimate bench.py quotes, the raw one is kept beside it
            e["hbm_bytes_raw"] = (f_kb + w_kb) * 1024
            e["hbm_bytes"] = (2 * f_kb + w_kb) * 1024
        out[k] = e
    json.dump(out, open(os.path.join(d, f"counters_{w}.json"), "w"), indent=1)
    for k in sorted(out):
        if k.startswith(("forward_kernel", "backward")) and "[stats]" not in k:
            e = out[k]
            v = e.get("SQ_ACTIVE_INST_VALU")
            g = e.get("GRBM_GUI_ACTIVE")
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 GRBM instances): per-XCD cycles = g / 8
            print(f"[{w}] {k}: dur {e['duration_ns'] / 1e6:.3f} ms"
                  + (f", VALU issue {4 * v / (1024 * g / 8):.3f}, VALU insts {e.get('SQ_INSTS_VALU', 0):.4g}, "
                     f"eff clock {g / 8 / e['duration_ns']:.2f} GHz" if v and g else "")
                  + (f", HBM {e['hbm_bytes'] / 1e9:.2f} GB (raw {e['hbm_bytes_raw'] / 1e9:.2f})" if "hbm_bytes" in e else ""))

for f in sorted(glob.glob(os.path.join(d, "bench*.json"))):
    try:
        line = [l for l in open(f) if l.startswith("{")][-1]
        b = json.loads(line)
        det = b["detail"]
        print(os.path.basename(f), b["value"], b["unit"], "| ms/step", b["ms_per_step"], "fwd", det.get("forward_ms"), "bwd",
              det.get("backward_ms"), "pack", det.get("foam_pack_ms"), "| n_gpus", b["n_gpus"], b["scaling"])
        if "ms_per_iteration" in det:
            print("   loop:", det["ms_per_iteration"], det.get("rebuilds"), det.get("densification"))
        for name, o in (b.get("other_workloads") or {}).items():
            print("   other:", name, {k: o.get(k) for k in ("value", "forward_ms", "backward_ms", "matches_gpu_bitwise",
                                                            "points_grad_rel_l2", "attr_grad_rel_l2", "seconds", "error")},
                  (o.get("roofline") or {}).get("bound"), (o.get("roofline") or {}).get("frac"))
        r = b.get("roofline")
        if r:
            print("   roofline:", {k: r[k] for k in ("bound", "kernel", "frac", "traffic", "algorithmic_GBps")}, r["hbm"])
        c = b.get("cpu_baseline")
        if c:
            print("   cpu:", {k: v for k, v in c.items() if k != "sample"})
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), "unreadable:", e, open(f).read()[-300:] if os.path.exists(f) else "")
for f in ("pytest_gpu.log", "smoke.log"):
    p = os.path.join(d, f)
    if os.path.exists(p):
        print(f, "|", open(p).read().strip().splitlines()[-1:])

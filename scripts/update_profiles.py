"""Copy the evidence of the last scripts/gpu_full.sh run from gpurun_out/ into profiles/<round>/<tag>_*
and refresh profiles/hbm_traffic.json (the per-launch HBM bytes bench.py quotes as roofline.traffic).
usage: python scripts/update_profiles.py r01 e_dealt_tiles"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")
shutil.copy(os.path.join(g, "prof", "bench2m_kernel_stats.csv"), os.path.join(out, tag + "_kernel_stats.csv"))
line = [l for l in open(os.path.join(g, "bench.json")) if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(out, tag + "_bench.json"), "w"), indent=1)
tr = json.load(open(os.path.join(g, "traffic", "traffic.json")))
json.dump(tr, open(os.path.join(out, tag + "_hbm_traffic_pmc.json"), "w"), indent=1)
w = bench["config"]
kernels = {}
for name, v in tr.items():
    short = name.split("rf::")[1].split("<")[0]
    if "true, true" in name:      # the statistics instance of the forward kernel is not a bench kernel
        continue
    kernels[short] = {"hbm_bytes_per_launch": int(v["hbm_bytes_fetch_x2"]),
                      "fetch_size_kb": v["FETCH_SIZE_KB"], "write_size_kb": v["WRITE_SIZE_KB"]}
json.dump({"workload": {"num_points": w["num_points"], "sh_degree": w["sh_degree"], "width": 1920, "height": 1080,
                        "seed": 5},
           "source": "profiles/%s/%s_hbm_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate "
                     "passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests as "
                     "64 B, MI355X_MICROARCH.md HBM section)" % (rnd, tag),
           "kernels": kernels}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(bench["value"], bench["roofline"], sorted(kernels))

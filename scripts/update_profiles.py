"""Copy the evidence of the last scripts/gpu_evidence.sh run from gpurun_out/ev/ into profiles/<round>/<tag>_* and
refresh profiles/counters.json (the per-launch hardware counters bench.py's roofline quotes).
usage: python scripts/update_profiles.py r02 a_baseline"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
ev = os.path.join(ROOT, "gpurun_out", "ev")
kept = []


def keep(src, name):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(out, f"{tag}_{name}"))
        kept.append(f"{tag}_{name}")


for f in sorted(glob.glob(os.path.join(ev, "bench*.json"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        name = os.path.basename(f)
        json.dump(json.loads(lines[-1]), open(os.path.join(out, f"{tag}_{name}"), "w"), indent=1)
        kept.append(f"{tag}_{name}")
for pdir in sorted(glob.glob(os.path.join(ev, "prof_*"))):
    w = os.path.basename(pdir)[5:]
    for f in glob.glob(os.path.join(pdir, "**", "*kernel_stats.csv"), recursive=True):
        keep(f, f"{w}_kernel_stats.csv")
for pm in sorted(glob.glob(os.path.join(ev, "pmc_*"))):
    w = os.path.basename(pm)[4:]
    for i, f in enumerate(sorted(glob.glob(os.path.join(pm, "p*", "**", "*counter_collection.csv"), recursive=True))):
        # the raw per-dispatch rows of the walk kernels only (the full files are tens of MB with the torch kernels)
        rows = [l for j, l in enumerate(open(f)) if j == 0 or "rf::" in l]
        open(os.path.join(out, f"{tag}_{w}_pmc_pass{i + 1}_counter_collection.csv"), "w").writelines(rows)
        kept.append(f"{tag}_{w}_pmc_pass{i + 1}_counter_collection.csv")
    keep(os.path.join(ev, f"counters_{w}.json"), f"{w}_counters.json")
for f in ("pytest_gpu.log", "smoke.log"):
    keep(os.path.join(ev, f), f)

# profiles/counters.json: workload -> {source, kernels: {kernel: counters}}, merged over rounds / tags
cj = os.path.join(ROOT, "profiles", "counters.json")
allc = json.load(open(cj)) if os.path.exists(cj) else {}
sha_file = os.path.join(ev, "csrc_sha256.txt")
csrc_sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None
for f in sorted(glob.glob(os.path.join(ev, "counters_*.json"))):
    w = os.path.basename(f)[9:-5]
    if w == "list":
        continue
    allc[w] = {
        # sha256 of the kernel sources the passes were taken on (radfoam_amd.build.source_hash() on the GPU box):
        # bench.py quotes these counters only for the same sources
        "csrc_sha256": csrc_sha,
        "source": f"profiles/{rnd}/{tag}_{w}_counters.json = per-launch averages of profiles/{rnd}/{tag}_{w}_pmc_pass*_counter_collection.csv "
                  "(rocprofv3 --kernel-trace --pmc, one pass per counter group: SQ+GRBM, SQ+GRBM, FETCH_SIZE, WRITE_SIZE; "
                  "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE tallies 128-B requests as 64 B, "
                  "MI355X_MICROARCH.md HBM section; hbm_bytes_raw = without the doubling)",
        "kernels": {k: v for k, v in json.load(open(f)).items() if "[stats]" not in k},
    }
json.dump(allc, open(cj, "w"), indent=1)
for extra in ("probe_global_atomics.log",):
    keep(os.path.join(ev, extra), extra)
# the ISA constants bench.py's useful_valu_frac rests on, from the same sources (no GPU needed: hipcc -S)
import subprocess
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_stats.py"), "--constants"], check=False)
print("kept under profiles/%s:" % rnd, *kept, sep="\n  ")

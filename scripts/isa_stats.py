"""Static ISA statistics of the walk kernels without a GPU: compiles rf_kernels.hip for gfx950 to assembly
(device only) and prints, per kernel, the instruction mix, register counts, scratch and LDS.
usage: python scripts/isa_stats.py [extra -D flags ...] [--filter substring] [--dump kernel_substring out.s]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
flt, dump = None, None
if "--filter" in args:
    i = args.index("--filter")
    flt = args[i + 1]
    del args[i:i + 2]
if "--dump" in args:
    i = args.index("--dump")
    dump = (args[i + 1], args[i + 2])
    del args[i:i + 3]
src = os.path.join(ROOT, "radfoam_amd", "csrc", "rf_kernels.hip")
out = "/tmp/rf_kernels_isa.s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
       "--cuda-device-only", "-S", src, "-o", out] + args
subprocess.run(cmd, check=True)
text = open(out).read()
# kernels: from "<name>:" label (a global function symbol) to ".end_amdhsa_kernel" metadata; simpler: split on .globl
funcs = re.split(r"\n\s*\.globl\s+", text)
kernel_meta = text[text.find("amdhsa.kernels"):].split("\n  - ")
rows = []
for f in funcs[1:]:
    name = f.split("\n", 1)[0].strip()
    body = f.split(".Lfunc_end", 1)[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    if dump and dump[0] in dem:
        open(dump[1], "w").write(body)
    ins = [l.strip().split()[0] for l in body.splitlines()
           if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.strip()]
    c = lambda pre: sum(1 for i in ins if i.startswith(pre))
    meta = {}
    entry = next((e for e in kernel_meta if re.search(r"\.name:\s+%s\s" % re.escape(name), e)), "")
    for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
        m = re.search(r"\.%s:\s+(\d+)" % key, entry)
        meta[key] = int(m.group(1)) if m else -1
    rows.append((dem.split("(")[0][:70], len(ins), c("v_"), c("v_pk_"), c("s_"), c("global_") + c("buffer_") + c("flat_"),
                 c("ds_"), c("v_rcp") + c("v_rsq") + c("v_sqrt") + c("v_exp") + c("v_log"), meta))
print(f"{'kernel':70s} {'insts':>6s} {'valu':>6s} {'pk':>5s} {'salu':>6s} {'vmem':>5s} {'lds':>5s} {'trans':>5s}  vgpr sgpr scratch lds_bytes")
for r in rows:
    m = r[8]
    print(f"{r[0]:70s} {r[1]:6d} {r[2]:6d} {r[3]:5d} {r[4]:6d} {r[5]:5d} {r[6]:5d} {r[7]:5d}  {m['vgpr_count']:4d} {m['sgpr_count']:4d} "
          f"{m['private_segment_fixed_size']:7d} {m['group_segment_fixed_size']:9d}")

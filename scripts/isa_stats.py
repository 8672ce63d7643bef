"""Static ISA statistics of the walk kernels without a GPU: compiles rf_kernels.hip for gfx950 to assembly
(device only) and prints, per kernel, the instruction mix, register counts, scratch and LDS.
usage: python scripts/isa_stats.py [extra -D flags ...] [--filter substring] [--dump kernel_substring out.s]
       python scripts/isa_stats.py --constants [out.json]
--constants: the instruction counts bench.py's roofline rests on (SCAN_VALU_PER_4_FACES, HOP_VALU_PER_LANE), read off the
ISA of the shipped forward instance instead of being typed in: VALU instructions of the face-scan loop (the innermost
loop: one block of four faces per trip) and of one wave-step outside it (the rest of the outer loop's body), with the
sha256 of the sources they were compiled from; written to profiles/isa_constants.json by default."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
flt, dump = None, None
constants_out = None
if "--constants" in args:
    i = args.index("--constants")
    constants_out = args[i + 1] if i + 1 < len(args) and not args[i + 1].startswith("-") else \
        os.path.join(ROOT, "profiles", "isa_constants.json")
    del args[i:i + (2 if i + 1 < len(args) and not args[i + 1].startswith("-") else 1)]
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


def loop_counts(body):
    """VALU instructions of the innermost loop and of the rest of the outermost loop's body, from the loop annotations
    the compiler writes behind block labels ('=>This Loop Header: Depth=1', 'Parent Loop BBx_y Depth=1' + 'This Inner
    Loop Header: Depth=2', 'in Loop: Header=BBx_y Depth=1')."""
    lines = body.splitlines()
    is_ins = lambda l: l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))
    starts = [i for i, l in enumerate(lines) if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)] + [len(lines)]
    depth_of = {}
    for a, b in zip(starts[:-1], starts[1:]):
        note = lines[a] + " " + (lines[a + 1] if a + 1 < len(lines) and lines[a + 1].lstrip().startswith(";") else "")
        if "Inner Loop Header: Depth=2" in note or re.search(r"in Loop: Header=\S+ Depth=2", note):
            depth = 2
        elif "Depth=1" in note:
            depth = 1
        else:
            depth = 0
        depth_of[(a, b)] = depth
    ops = lambda a, b: [l.strip().split()[0] for l in lines[a:b] if is_ins(l)]
    valu = lambda o: sum(1 for x in o if x.startswith("v_"))
    # the kernels hold TWO inner loops: the filtered scan (first in program order: one block of four faces per trip) and the
    # dividing scan that resolves contested cells (rare: its trips are not what a wave-step costs)
    header_of = {}
    cur_header = None
    for a, b in zip(starts[:-1], starts[1:]):
        note = lines[a] + " " + (lines[a + 1] if a + 1 < len(lines) and lines[a + 1].lstrip().startswith(";") else "")
        if "Inner Loop Header: Depth=2" in note:
            cur_header = lines[a].split(":")[0]
            header_of[(a, b)] = cur_header
        else:
            m = re.search(r"in Loop: Header=(\S+) Depth=2", note)
            if m:
                header_of[(a, b)] = ".L" + m.group(1) if not m.group(1).startswith(".L") else m.group(1)
    inner = [k for k, d in sorted(depth_of.items()) if d == 2]
    first = header_of.get(inner[0]) if inner else None
    scan = [x for k in inner if header_of.get(k) == first for x in ops(*k)]
    # a wave-step outside the scan: the hop (link, next cell record, trail entry, loop bookkeeping) and -- recognisable by
    # what only they contain -- the compositing blocks: the colour-row gather (>= 3 dwordx4 loads), exp (v_ldexp_f32),
    # the contribution atomic
    hop = comp = 0
    for (a, b), d in depth_of.items():
        if d != 1:
            continue
        o = ops(a, b)
        is_comp = sum(1 for x in o if x.startswith("global_load_dwordx4")) >= 3 or any("ldexp" in x for x in o) or \
            any(x.startswith("global_atomic_add") for x in o)
        if is_comp:
            comp += valu(o)
        else:
            hop += valu(o)
    n = lambda pre: sum(1 for x in scan if x.startswith(pre))
    flop = 4 * n("v_pk_fma_f32") + 2 * (n("v_pk_mul_f32") + n("v_pk_add_f32")) + 2 * (n("v_fma_f32") + n("v_fmac_f32")) + \
        n("v_mul_f32") + n("v_add_f32") + n("v_sub_f32")
    return {"scan_valu_per_4_faces": valu(scan), "scan_flop_per_4_faces": flop, "hop_valu_per_lane": hop,
            "composite_valu_per_lane": comp}


if constants_out:
    import hashlib
    import json
    sys.path.insert(0, ROOT)
    from radfoam_amd import build as hip_build
    want = "void rf::forward_kernel<2, false, false, false, false, 0>"
    found = None
    for f in funcs[1:]:
        name = f.split("\n", 1)[0].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if dem.split("(")[0] == want:
            found = loop_counts(f.split(".Lfunc_end", 1)[0])
    assert found, "forward instance not found in the ISA"
    # The rescans of contested cells (scan_resolve / scan_faces_strict) are inlined into the step: their preambles and
    # epilogues sit in depth-1 blocks and would be counted as "hop" although 97.7 % of the wave-steps never execute them
    # (92 -> 103 "per hop" between rounds 5 and 6 without one instruction more on the hot path).  The hop figure is
    # therefore read off a second compile with the rescans compiled out (-DRF_ISA_NO_CONTESTED_PATH: never shipped); the
    # scan and composite figures stay those of the shipped build (without the rescans the certificate is dead code and the
    # scan would read 60 instead of 67).
    if "-DRF_ISA_NO_CONTESTED_PATH" not in args:
        out2 = "/tmp/rf_kernels_isa_nc.s"
        subprocess.run([c if c != out else out2 for c in cmd] + ["-DRF_ISA_NO_CONTESTED_PATH"], check=True)
        hot = None
        for f in re.split(r"\n\s*\.globl\s+", open(out2).read())[1:]:
            name = f.split("\n", 1)[0].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if dem.split("(")[0] == want:
                hot = loop_counts(f.split(".Lfunc_end", 1)[0])
        assert hot, "forward instance not found in the second ISA"
        found["hop_valu_per_lane_with_inlined_rescans"] = found["hop_valu_per_lane"]
        found["hop_valu_per_lane"] = hot["hop_valu_per_lane"]
    rec = {"csrc_sha256": hip_build.source_hash(), "kernel": want,
           "source": "scripts/isa_stats.py --constants (hipcc -S of radfoam_amd/csrc/rf_kernels.hip, gfx950; hop_valu_per_lane from "
                     "a second compile with the rescans of contested cells compiled out, see the script)", **found}
    json.dump(rec, open(constants_out, "w"), indent=1)
    print(json.dumps(rec))
    sys.exit(0)

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

"""Instruction counts per basic block of one kernel of libptcore: scalar, vector, LDS and memory instructions, and
which blocks end in a backward branch (loops).  For comparing two versions of a hot loop without a GPU.
usage: python tools/asm_blocks.py <substring of the mangled kernel name> [--defs "-DX=1 ..."]"""
import re
import subprocess
import sys
import tempfile

name = sys.argv[1]
defs = sys.argv[sys.argv.index("--defs") + 1].split() if "--defs" in sys.argv else []
with tempfile.TemporaryDirectory() as tmp:
    out = f"{tmp}/ptcore.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           "-Iinclude", "--cuda-device-only", "-S", "-o", out, "pathtracer_amd/csrc/ptcore.hip"] + defs, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l and not l.startswith("\t"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
if "--save" in sys.argv:
    open(sys.argv[sys.argv.index("--save") + 1], "w").write("\n".join(body))
blocks, cur = [], None
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = {"label": m.group(1) if m else "entry", "line": i, "s": 0, "v": 0, "ds": 0, "mem": 0, "br": []}
        blocks.append(cur)
        if m:
            continue
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith("s_"):
        cur["s"] += 1
        if "branch" in op:
            cur["br"].append(t.split()[1])
    elif op.startswith("v_"):
        cur["v"] += 1
    elif op.startswith("ds_"):
        cur["ds"] += 1
    elif op.split("_")[0] in ("global", "buffer", "scratch", "flat"):
        cur["mem"] += 1
idx = {b["label"]: k for k, b in enumerate(blocks)}
print(f"{lines[start].split(':')[0]}: {len(blocks)} blocks, s={sum(b['s'] for b in blocks)} v={sum(b['v'] for b in blocks)} ds={sum(b['ds'] for b in blocks)} mem={sum(b['mem'] for b in blocks)}")
if "--blocks" in sys.argv:
    for k, b in enumerate(blocks):
        back = [t for t in b["br"] if t in idx and idx[t] <= k]
        print(f"{k:3d} {b['label']:12s} L{b['line']:5d} s={b['s']:3d} v={b['v']:3d} ds={b['ds']} mem={b['mem']:2d} {'BACK->' + ','.join(back) if back else ''}")

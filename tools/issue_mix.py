"""Static instruction mix of the closest-hit kernels, priced with the issue costs tools/micro/valu_rate.hip measured on
an MI355X (profiles/r03_valu_issue_rates.txt): what one wave64 vector instruction of these kernels costs a SIMD on
average, and how many scalar instructions go with it.  bench.py weights its `binding` figures with the result.

A wave64 v_add / v_sub / v_mul / v_fma / v_and / v_xor / v_mov issues every 2 cycles per SIMD (32 lanes per clock);
v_min / v_max / v_min3 / v_max3 / v_cmp / shifts / v_bfe / 24-bit and 32-bit integer multiplies, the three-operand
integer forms, v_cndmask on a mask in scalar registers and the packed f32 forms every 4;
v_rcp / v_sqrt / v_rsq / v_exp / v_log / v_sin / v_cos every 8; scalar instructions every 4 per SIMD, because the
four SIMDs of a CU share one scalar unit.  The mix is static (every instruction of the kernel counts once, whatever
the loop it sits in), which is a fair stand-in here: nine tenths of these kernels is their main loop.

usage: python tools/issue_mix.py [out.json]      (compiles pathtracer_amd/csrc/ptcore.hip to assembly: two minutes)"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {
    "k_extend_walk": "_Z13k_extend_walkILb0ELb0ELb0ELb0EE",
    "k_extend_walk<tail>": "_Z13k_extend_walkILb0ELb0ELb0ELb1EE",
    "k_extend_walk<instances>": "_Z13k_extend_walkILb1ELb0ELb0ELb0EE",
    "k_extend_packet": "_Z15k_extend_packetILb0ELb0EE",
    "k_extend_packet<instances>": "_Z15k_extend_packetILb1ELb0EE",
    "k_shadow_walk": "_Z13k_shadow_walkILb0EE",
}
# measured at 4 cycles (tools/micro/valu_rate.hip): min, max, min3, max3, compares, shifts, bfe, lshl_add, add3, mad_u32_u24, mul_lo,
# div_scale, div_fixup, cvt, the packed f32 forms, and v_cndmask with its mask in a scalar register pair (the e64 encoding;
# with vcc straight after the compare that wrote it the pair costs 6.6, i.e. the select 2); not measured and counted
# as 4: the other three-operand integer forms, lane reads and writes, div_fmas
HALF = re.compile(r"^v_(min|max|med3|cmp|cmpx|lshl|lshr|ashr|mad_|mul_lo|mul_hi|mul_u32|mul_i32|bfe|bfi|pk_|add_lshl|add3|and_or|or3|xad|mbcnt|readlane|writelane|readfirstlane|ldexp|frexp|div_scale|div_fmas|div_fixup|cvt|cndmask_b32_e64)")
QUARTER = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)")


def csrc_sha256():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pathtracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())  # bench.py's csrc_sha256
    return h.hexdigest()


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_issue_mix.json")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "ptcore.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm, os.path.join(ROOT, "pathtracer_amd", "csrc", "ptcore.hip")],
                              stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    res = {"about": __doc__.split("\n\n")[0], "costs_cycles_per_simd": {"full": 2, "half": 4, "quarter": 8, "scalar": 4}, "csrc_sha256": csrc_sha256(), "kernels": {}}
    for name, mangled in KERNELS.items():
        start = next((i for i, l in enumerate(lines) if l.startswith(mangled) and ":" in l), None)
        if start is None:
            continue
        end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        n = {"full": 0, "half": 0, "quarter": 0, "scalar": 0, "vmem": 0, "lds": 0}
        for l in lines[start:end]:
            t = l.strip()
            if not t or t[0] in ";.":
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                n["quarter" if QUARTER.match(op) else "half" if HALF.match(op) else "full"] += 1
            elif op.startswith("s_") and not op.startswith(("s_load", "s_buffer")):
                n["scalar"] += 1
            elif op.split("_")[0] in ("global", "buffer", "scratch", "flat"):
                n["vmem"] += 1
            elif op.startswith("ds_"):
                n["lds"] += 1
        valu = n["full"] + n["half"] + n["quarter"]
        n["cycles_per_vector_instruction"] = (2 * n["full"] + 4 * n["half"] + 8 * n["quarter"]) / valu
        n["scalar_per_vector_instruction"] = n["scalar"] / valu
        res["kernels"][name] = n
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()

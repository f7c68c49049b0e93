"""Static VALU instruction mix of a kernel, priced with the measured cost of every instruction class (tools/ubench_valu.hip ->
profiles/r5_ubench_valu.txt): what ONE wave64 VALU instruction of that kernel occupies its SIMD for, on average.  bench.py multiplies it with the
SQ_INSTS_VALU the counters report per kernel class, so that the loop's VALU line is one number instead of the 0.70-0.98 interval the counter
categories leave (their INT32 / "other" buckets mix 2-cycle and 3.2-cycle instructions).

    python tools/isa_histogram.py [--lib tungsten_amd/lib/libtungsten_hip.so] [--json out.json] k_finish_trace_closest_wide k_trace_shadow_fast ...

The code object is the gfx950 member of the library's .hip_fatbin bundle (no GPU needed).  Two mixes per kernel: every VALU instruction of the
kernel, and those of its largest loop (a backward branch's [target, branch] range -- the walk's turn, the shading loop over slots), which is what
runs; the latter is the one priced.  Instructions the micro-benchmark did not time are priced like their nearest relative and listed."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

# cycles per wave64 instruction per SIMD (v_fma_f32 = 2.0), profiles/r5_ubench_valu.txt
FAST = 1.85      # f32 add / sub / mul / fma / fmac, and / or / xor / not, add / sub u32, mov: 1.74-2.00
SLOW = 3.2       # min / max (also 3-operand), shifts, bfe / bfi / perm, and_or / or3 / lshl_add / lshl_or / xad, converts, compares, integer multiplies, bit counts
PACKED = 3.7     # v_pk_*_f32
F64 = 3.8
TRANS = 6.25     # rcp / rsq / sqrt / exp / log / sin / cos
CNDMASK = 1.7    # v_cmp + v_cndmask measured 2.52 per instruction as a pair, v_cmp alone 3.38


def price(op):
    """(cycles, measured?) of VALU opcode `op` (suffixes _e32 / _e64 / _sdwa / _dpp stripped)."""
    o = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if o.startswith("v_pk_"):
        return PACKED, o in ("v_pk_fma_f32",)
    if "f64" in o:
        return F64, o == "v_fma_f64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy|_clamp)?_f(32|16)$", o):
        return TRANS, o in ("v_rcp_f32", "v_sqrt_f32")
    if re.match(r"v_(fma|fmac|mul|add|sub|subrev|mac|mad)_f32$", o) or o in ("v_mul_legacy_f32",):
        return FAST, o in ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32")
    if re.match(r"v_(and|or|xor|xnor|not)_b32$", o) or re.match(r"v_(add|sub|subrev)(_co)?_u32$", o) or re.match(r"v_(addc|subb|subbrev)_co_u32$", o) or o in ("v_mov_b32", "v_add_i32", "v_sub_i32"):
        return FAST, o in ("v_and_b32", "v_add_u32", "v_mov_b32")
    if o == "v_cndmask_b32":
        return CNDMASK, True
    if o.startswith("v_cmp") or o.startswith("v_cmpx"):
        return 3.38, o == "v_cmp_lt_f32"
    if o in ("v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32"):      # spills to AGPRs
        return FAST, False
    measured = ("v_max_f32", "v_max3_f32", "v_med3_f32", "v_min3_f32", "v_lshl_or_b32", "v_or3_b32", "v_lshlrev_b32", "v_lshl_add_u32", "v_bfe_u32", "v_and_or_b32",
                "v_xad_u32", "v_perm_b32", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_bcnt_u32_b32", "v_ffbl_b32", "v_fma_mix_f32", "v_cvt_f32_f16")
    return SLOW, o in measured


def code_objects(lib, tmp):
    """The gfx950 code objects inside `lib`: its .hip_fatbin section is one clang offload bundle per translation unit, back to back."""
    fb = os.path.join(tmp, "x.fb")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, lib, os.path.join(tmp, "x.o")], stderr=subprocess.DEVNULL)
    data = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    out = []
    for i, st in enumerate(starts):
        part, co = os.path.join(tmp, "b%d.fb" % i), os.path.join(tmp, "b%d.co" % i)
        with open(part, "wb") as f:
            f.write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
        if subprocess.call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co],
                           stderr=subprocess.DEVNULL) == 0 and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernel_symbols(co):
    """{demangled name without the parameter list: mangled} of the code object's kernels"""
    out = subprocess.check_output([LLVM + "/llvm-readelf", "--symbols", "--wide", co], text=True)
    mangled = [l.split()[-1] for l in out.splitlines() if " FUNC " in l and " GLOBAL " in l and not l.split()[-1].endswith(".kd")]
    names = subprocess.check_output(["c++filt"] + mangled, text=True).split("\n") if mangled else []
    return {re.sub(r"\(.*", "", n).replace("void ", "").strip(): m for n, m in zip(names, mangled)}


def histogram(co, mangled):
    asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--disassemble-symbols=" + mangled, co], text=True)
    ins = []                                     # (address, opcode, branch target or None)
    for line in asm.splitlines():
        m = re.match(r"\s+((?:v|s|global|ds|buffer|scratch|flat)_[a-z0-9_]+)\b(.*?)//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        tgt = None
        if m.group(1).startswith("s_cbranch") or m.group(1) == "s_branch":
            t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
            t2 = re.search(r"//\s*[0-9A-Fa-f]+:\s*[0-9A-Fa-f]+\s*<[^+>]+\+0x([0-9a-fA-F]+)>", line)
            tgt = (t2 or t)
            tgt = int(tgt.group(1), 16) if tgt else None
        ins.append((int(m.group(3), 16), m.group(1), tgt))
    if not ins:
        return None
    base = ins[0][0]
    loops = [(a - (base + t), base + t, a) for a, op, t in ins if t is not None and base + t < a]
    loop = max(loops)[1:] if loops else (ins[0][0], ins[-1][0])

    def mix(sel):
        c = collections.Counter(re.sub(r"_(e32|e64|sdwa|dpp)$", "", op) for a, op, _ in ins if op.startswith("v_") and sel(a))
        n = sum(c.values())
        cyc = sum(price(op)[0]*k for op, k in c.items())
        unm = {op: k for op, k in c.items() if not price(op)[1]}
        cls = collections.Counter()
        for op, k in c.items():
            cls[{FAST: "fast_1.85", SLOW: "slow_3.2", PACKED: "packed_3.7", F64: "f64_3.8", TRANS: "trans_6.25", CNDMASK: "cndmask_1.7", 3.38: "cmp_3.38"}[price(op)[0]]] += k
        return {"valu_instructions": n, "cycles_per_instruction": round(cyc/n, 3) if n else None,
                "classes": {k: round(v/float(n), 4) for k, v in sorted(cls.items())} if n else {},
                "unmeasured_share": round(sum(unm.values())/float(n), 4) if n else 0.0,
                "unmeasured_top": dict(sorted(unm.items(), key=lambda kv: -kv[1])[:8])}
    return {"instructions": len(ins), "loop_instructions": sum(1 for a, _, _ in ins if loop[0] <= a <= loop[1]),
            "kernel": mix(lambda a: True), "loop": mix(lambda a: loop[0] <= a <= loop[1])}


def kernel_prices(lib, wanted):
    """{kernel name as given: histogram} for the kernels whose demangled names START with one of `wanted` (every instantiation)."""
    with tempfile.TemporaryDirectory() as tmp:
        out = {}
        for co in code_objects(lib, tmp):
            for name, mangled in sorted(kernel_symbols(co).items()):
                if any(name == w or name.startswith(w + "<") for w in wanted):
                    h = histogram(co, mangled)
                    if h:
                        out[name] = h
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "tungsten_amd", "lib", "libtungsten_hip.so"))
    ap.add_argument("--json")
    ap.add_argument("kernels", nargs="*", default=["k_finish_trace_closest_wide", "k_trace_shadow_fast"])
    a = ap.parse_args()
    res = kernel_prices(a.lib, a.kernels)
    for k, h in res.items():
        print("%-64s %6d instr, loop %6d | VALU %6d at %.2f cyc (loop: %6d at %.2f cyc, unmeasured %.1f %%) %s" % (
            k[:64], h["instructions"], h["loop_instructions"], h["kernel"]["valu_instructions"], h["kernel"]["cycles_per_instruction"] or 0,
            h["loop"]["valu_instructions"], h["loop"]["cycles_per_instruction"] or 0, 100*h["loop"]["unmeasured_share"], h["loop"]["classes"]))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

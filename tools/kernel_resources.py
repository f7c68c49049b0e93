"""Registers, scratch, spills, LDS and instruction counts of every kernel in build/obj/*.o -- read from the code objects hipcc embedded, no GPU needed.

    python tools/kernel_resources.py [--json out.json] [--diff old.json] [--filter k_shade]

Each object file carries a clang offload bundle in its .hip_fatbin section; the gfx950 code object inside has the kernels' metadata as an ELF note
(llvm-readelf --notes) and their code (llvm-objdump -d).  `--diff` prints only the kernels whose numbers changed against an earlier `--json` dump:
the check a kernel change gets before it gets GPU time (an occupancy step crossed, a spill that appeared, a path that doubled in length)."""
import argparse
import collections
import glob
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, tmp):
    fb, co = os.path.join(tmp, "x.fb"), os.path.join(tmp, "x.co")
    if subprocess.call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, obj, os.path.join(tmp, "x.o")], stderr=subprocess.DEVNULL) != 0:
        return None
    if subprocess.call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co],
                       stderr=subprocess.DEVNULL) != 0:
        return None
    return co


def kernels_of(obj, tmp):
    co = code_object(obj, tmp)
    if not co:
        return {}
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    out = {}
    for blk in notes.split("- .agpr_count")[1:]:
        def field(name, default=0):
            m = re.search(r"\.%s:\s+(\S+)" % name, blk)
            return m.group(1) if m else default
        name = field("name", "?")
        out[name] = {"vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")), "scratch": int(field("private_segment_fixed_size")),
                     "vgpr_spills": int(field("vgpr_spill_count")), "lds": int(field("group_segment_fixed_size"))}
    demangled = dict(zip(out, subprocess.check_output(["c++filt"] + list(out), text=True).split("\n"))) if out else {}
    asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True)
    for fn in re.split(r"\n(?=[0-9a-f]+ <)", asm):
        m = re.match(r"[0-9a-f]+ <([^>]+)>", fn)
        if not m or m.group(1) not in out:
            continue
        ops = re.findall(r"^\s+((?:v|s|global|ds|buffer|scratch|flat)_[a-z0-9_]+)", fn, re.M)
        c = collections.Counter(ops)
        out[m.group(1)].update(instructions=len(ops), f64=sum(v for k, v in c.items() if "f64" in k), branches=sum(v for k, v in c.items() if k.startswith("s_cbranch")),
                               vmem=sum(v for k, v in c.items() if k.split("_")[0] in ("global", "buffer", "flat", "scratch")))
    return {demangled.get(k, k).replace("(DeviceScene, PathState, PassParams, int)", "").replace("(DeviceScene, PathState, PassParams)", ""): v for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--diff")
    ap.add_argument("--filter", default="")
    ap.add_argument("--objdir", default=os.path.join(ROOT, "build", "obj"), help="directory of the object files (an experiment's build/obj_<variant>)")
    a = ap.parse_args()
    table = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(a.objdir, "*.o"))):
            for k, v in kernels_of(obj, tmp).items():
                table[os.path.basename(obj) + ": " + k] = v
    if a.json:
        with open(a.json, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
    old = json.load(open(a.diff)) if a.diff else None
    cols = ("vgpr", "sgpr", "scratch", "vgpr_spills", "lds", "instructions", "f64", "branches", "vmem")
    print("%-88s " % "kernel" + " ".join("%9s" % c for c in cols))
    for k in sorted(table):
        if a.filter not in k:
            continue
        v = table[k]
        if old is not None:
            o = old.get(k)
            if o == v:
                continue
            print("%-88s " % k[:88] + " ".join("%9s" % ("%s>%s" % (o.get(c, "-"), v.get(c, "-")) if o and o.get(c) != v.get(c) else v.get(c, "-")) for c in cols))
        else:
            print("%-88s " % k[:88] + " ".join("%9s" % v.get(c, "-") for c in cols))


if __name__ == "__main__":
    main()

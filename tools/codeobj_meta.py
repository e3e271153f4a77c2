"""Per-kernel resource usage of the gfx950 code object inside libdsim_hip.so (or any HIP shared library):
VGPRs, SGPRs, spilled registers, scratch bytes, static LDS - read from the code object's metadata note with the
ROCm LLVM tools (no GPU needed).  `python tools/codeobj_meta.py [lib.so] > profiles/rNN_codeobj_meta.txt`;
tests/test_codeobj.py uses kernels() to fail the CPU suite when the headline kernel spills."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "disimpy_amd", "csrc", "libdsim_hip.so")

FIELDS = (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".agpr_count")


def kernels(lib=DEFAULT_LIB):
    """[{name, vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, ...}]"""
    tmp = tempfile.mkdtemp(prefix="codeobj_")
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
        if not objs:
            raise RuntimeError("no gfx950 code object in %s" % lib)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", objs[0]], check=True, capture_output=True,
                               text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == ".agpr_count" or (key == ".args" and cur is None) or (line.lstrip().startswith("- ") and key in FIELDS + (".args",)):
            pass
        if line.lstrip().startswith("- .") and key in (".agpr_count", ".args"):
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        if key == ".name":
            cur["name"] = val.strip("'\"")
        elif key in FIELDS:
            cur[key[1:]] = int(val)
    return [k for k in out if "name" in k and "vgpr_count" in k]


def disassembly(lib=DEFAULT_LIB):
    """llvm-objdump -d of the gfx950 code object inside the library (text)"""
    tmp = tempfile.mkdtemp(prefix="codeobj_")
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
        if not objs:
            raise RuntimeError("no gfx950 code object in %s" % lib)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", objs[0]], check=True, capture_output=True,
                              text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _regs(operand):
    """VGPR numbers an operand names: v5 -> {5}, v[4:5] -> {4, 5}"""
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return {int(m.group(1))} if m else set()


def dpp_hazards(lib=DEFAULT_LIB):
    """DPP instructions whose DPP source (src0) is written by a vector instruction fewer than two wait states earlier
    (the hardware needs 2 between a VALU write of a VGPR and its read through DPP; hand-written `*_dpp` in inline
    assembly is invisible to the compiler's hazard recogniser - csrc/dsim_kernels.h row_bcast / row_fmac).  Returns a
    list of (function, instruction, offending earlier instruction)."""
    bad, window, func = [], [], "?"
    for line in disassembly(lib).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            func, window = m.group(1), []
            continue
        ins = line.split("//")[0].strip()
        if not ins or ins.endswith(":"):
            window = [] if ins.endswith(":") else window  # a label: control may arrive from elsewhere - start afresh
            continue
        parts = ins.replace(",", " ").split()
        op = parts[0]
        if "_dpp" in op and len(parts) >= 3:
            src = _regs(parts[2])
            states = 0
            for prev in reversed(window):
                if states >= 2:
                    break
                pp = prev.replace(",", " ").split()
                if pp[0] == "s_nop":
                    states += int(pp[1], 0) + 1
                    continue
                if pp[0].startswith("v_") and len(pp) > 1 and not pp[0].startswith("v_cmp") and (_regs(pp[1]) & src):
                    bad.append((func, ins, prev))
                    break
                states += 1
        window = (window + [ins])[-4:]
    return bad


def demangle(names):
    r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],
                       input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    ks = kernels(lib)
    names = demangle([k["name"] for k in ks])
    print("%-78s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "sgpr", "v-spill", "s-spill", "scratch", "LDS"))
    for k, n in sorted(zip(ks, names), key=lambda t: t[1]):
        n = re.sub(r"\(.*$", "", n.replace("void ", ""))
        print("%-78s %5d %5d %7d %7d %8d %7d" % (n[:78], k["vgpr_count"], k["sgpr_count"], k.get("vgpr_spill_count", 0),
                                               k.get("sgpr_spill_count", 0), k.get("private_segment_fixed_size", 0),
                                               k.get("group_segment_fixed_size", 0)))


if __name__ == "__main__":
    main()

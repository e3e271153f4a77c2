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

"""Kernel metadata (.vgpr_count / .agpr_count / spills / LDS / scratch) of every gfx950 kernel in libmimrl_hip.so.

The .so carries one clang offload bundle per translation unit; this extracts the gfx950 code objects and reads their
NT_AMDGPU_METADATA notes with llvm-readelf.  Used by tests/test_codeobj.py (register-budget tripwire) and as a CLI:
    python tools/codeobj_meta.py [pattern]      # table of kernels whose demangled name contains `pattern`
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mimrl_amd", "libmimrl_hip.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".max_flat_workgroup_size")


def code_objects(lib=LIB):
    b = open(lib, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", b):
        o = m.start()
        nb = struct.unpack_from("<Q", b, o + 24)[0]
        p = o + 32
        for _ in range(nb):
            off, sz, ts = struct.unpack_from("<QQQ", b, p)
            p += 24
            triple = b[p:p + ts].decode()
            p += ts
            if "gfx950" in triple and sz:
                out.append(b[o + off:o + off + sz])
    return out


def kernels(lib=LIB):
    """-> {demangled kernel name: {field: int}}"""
    meta = {}
    with tempfile.TemporaryDirectory() as d:
        for i, co in enumerate(code_objects(lib)):
            path = os.path.join(d, f"k{i}.co")
            open(path, "wb").write(co)
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", path], capture_output=True, text=True, check=True).stdout
            cur = {}
            for line in txt.splitlines():
                s = line.strip()
                if s.startswith("- .") or line.startswith("  - "):       # a new kernel entry of amdhsa.kernels
                    if line.startswith("  - "):
                        if "name" in cur:
                            meta[cur["name"]] = cur
                        cur = {}
                    s = s[2:]
                mm = re.match(r"(\.[a-z_]+):\s+(\S+)$", s)
                if not mm:
                    continue
                if mm.group(1) == ".name" and line.startswith("    .name"):
                    cur["name"] = mm.group(2)
                elif mm.group(1) in FIELDS and (line.startswith("    .") or line.startswith("  - .")):
                    cur[mm.group(1)[1:]] = int(mm.group(2))
            if "name" in cur:
                meta[cur["name"]] = cur
    names = list(meta)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
    out = {}
    for n, dn in zip(names, dem):
        dn = re.sub(r"\(anonymous namespace\)::", "", dn).replace("void mimrl::", "").replace("mimrl::", "")
        out[dn.split("(")[0]] = {k: v for k, v in meta[n].items() if k != "name"}
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    ks = kernels()
    print("%-64s %5s %5s %5s %6s %7s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratch", "lds"))
    for k in sorted(ks):
        if pat in k:
            v = ks[k]
            print("%-64s %5d %5d %5d %6d %7d %7d" % (k[:64], v["vgpr_count"], v["agpr_count"], v["sgpr_count"], v["vgpr_spill_count"],
                                                    v["private_segment_fixed_size"], v["group_segment_fixed_size"]))

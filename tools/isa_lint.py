"""ISA lint for two patterns that cost this code base real time (rules 1 and 2):
  * SERIALIZED LOADS: a vector-memory load whose next memory-relevant instruction is `s_waitcnt vmcnt(0)` -- what a guarded load
    (`cond ? *p : 0`) compiles to: one full round trip per load (the fused CubeMLP forward's set-up had twelve in a row);
  * branch density: s_cbranch per 100 instructions (a per-element `switch (act)` inside an unrolled epilogue).
usage: python tools/isa_lint.py [kernel-name-substring ...]   (disassembles the in-tree libmimrl_hip.so; no GPU needed)"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_meta as cm


def main():
    want = sys.argv[1:]
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for i, co in enumerate(cm.code_objects()):
            path = os.path.join(d, f"k{i}.co")
            open(path, "wb").write(co)
            txt = subprocess.run([os.path.join(cm.LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
                name, body = m.group(1), m.group(2)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"^void ", "", dem).replace("mimrl::(anonymous namespace)::", "").split("(")[0]
                if want and not any(w in dem for w in want):
                    continue
                ins = [l.split("//")[0].strip() for l in body.splitlines() if l.strip() and not l.strip().endswith(":")]
                ins = [l for l in ins if l and not l.startswith("s_nop") and not l.startswith(".")]
                if len(ins) < 50:
                    continue
                serial = 0
                for j, l in enumerate(ins):
                    if l.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                        for k in range(j + 1, min(j + 6, len(ins))):
                            t = ins[k]
                            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                                serial += 1
                                break
                            if t.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_", "v_mfma", "s_barrier")):
                                break
                br = sum(1 for l in ins if l.startswith("s_cbranch"))
                rows.append((serial, br * 100.0 / len(ins), len(ins), dem))
    rows.sort(reverse=True)
    print(f"{'serialized loads':>16s} {'branches/100':>12s} {'instructions':>12s}  kernel")
    for s, b, n, dem in rows[:60]:
        print(f"{s:16d} {b:12.1f} {n:12d}  {dem[:110]}")


if __name__ == "__main__":
    main()

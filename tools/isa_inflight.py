"""ISA check for hand-counted memory waits (gru.hip vm_wait<N>, gemm.hip fast_wait<N>): NO instruction may touch the destination register
of a vector-memory load while that load can still be in flight.

Why.  The bf16 BPTT kernel and the fast GEMM loaders issue their operand loads through inline asm -- invisible to the compiler's waitcnt
pass by design -- and wait for them with ONE explicit `s_waitcnt vmcnt(N)` whose count is exact because every step issues the same
sequence of loads and stores.  Nothing stops the register allocator from re-homing such a loop-carried destination with a `v_mov` placed
in FRONT of the wait, or from reusing a "dead" destination while its load is still outstanding: round 3 shipped exactly that in the odd-T
tail of gru_bwd_kernel<true, true> (stale gates in the last cell step, gradients off by ~10 % and different from run to run; found in round
4 by the odd-T case of test_gradients_reproducible).  For loads the compiler issued itself this check holds by construction, which
validates the analysis.

Method: forward dataflow over the kernel's control-flow graph.  State = {vgpr: fewest vector-memory operations issued behind the youngest
outstanding load into it}; a load starts at 0, every later load / store / atomic adds one, `s_waitcnt vmcnt(n)` retires every entry with
>= n younger operations (vmcnt retires in order on gfx9; 63 younger operations retire it too: the counter has 6 bits), joins take the union with the smaller count.  Any instruction that names an
in-flight register -- as a source (stale read, copies included) or as a destination (the load would land on top of the new value) -- is
reported.   usage: python tools/isa_inflight.py [--ring] [kernel-name-substring ...]      (--ring: excuse the register-ring pattern, see analyse)"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_meta as cm  # noqa: E402

VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store", "flat_atomic",
        "scratch_load", "scratch_store")
LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load")
# LDS-DMA (global_load_lds_* / buffer_load ... lds, round 5: gemm_tall.hip): the data lands in LDS, there is NO register destination -- the
# first operand is the ADDRESS, which the hardware has read at issue.  They still count on vmcnt like every other VMEM operation.
def _is_reg_load(mn, line):
    return mn.startswith(LOADS) and not mn.startswith("global_load_lds") and " lds" not in line.split("//")[0]
CAP = 63   # vmcnt is a 6-bit counter: the 64th outstanding operation cannot issue before the oldest has retired


def _regs(tok):
    """vgprs named by one operand token: v12 -> {12}; v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(body):
    """-> [(addr, mnemonic, [operand tokens], branch target addr or None)]"""
    ins = []
    for line in body.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", ops) if t.strip()] if ops else []
        toks = [t.split()[0] for t in toks]                       # drop modifiers ("off", "offset:256", "op_sel:[..]")
        tgt = None
        if mn.startswith(("s_cbranch", "s_branch")):
            mt = re.search(r"<[^>]*\+0x([0-9A-Fa-f]+)>", line)
            tgt = mt.group(1) if mt else None
        ins.append([addr, mn, toks, tgt, line])
    return ins


def analyse(body, base_addr, ring_excuse=False):
    ins = parse(body)
    if not ins:
        return []
    index = {a: i for i, (a, *_rest) in enumerate(ins)}
    for it in ins:                                                # branch targets are symbol-relative offsets
        if it[3] is not None:
            it[3] = index.get(base_addr + int(it[3], 16))
    n = len(ins)
    state = [None] * n                                            # state at the ENTRY of instruction i
    state[0] = {}
    work = [0]
    bad = {}
    while work:
        i = work.pop()
        st = dict(state[i])
        addr, mn, toks, tgt, line = ins[i]
        is_vmem = mn.startswith(VMEM)
        named = set()
        for t in toks:
            named |= _regs(t)
        hit = named & set(st)
        origin = {r: st[r][1] for r in hit}
        if hit and _is_reg_load(mn, line) and toks:               # a second load into a register whose first load is still outstanding is
            hit -= _regs(toks[0]) - set().union(*[_regs(t) for t in toks[1:]] or [set()])   # fine (in-order return); its ADDRESS must not be in flight
        if hit:
            bad[i] = (addr, line.strip().split("//")[0].strip(), sorted(hit), sorted({origin[r] for r in hit}))
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", " ".join(toks) + " " + line)
            if m:
                k = int(m.group(1))
                st = {r: v for r, v in st.items() if v[0] < k}
        if is_vmem:
            st = {r: (v[0] + 1, v[1]) for r, v in st.items() if v[0] + 1 < CAP}
            if _is_reg_load(mn, line) and toks:
                for r in _regs(toks[0]):
                    st[r] = (0, addr)
        succ = []
        if mn == "s_endpgm":
            succ = []
        elif mn.startswith("s_branch"):
            succ = [tgt] if tgt is not None else []
        else:
            if i + 1 < n:
                succ.append(i + 1)
            if mn.startswith("s_cbranch") and tgt is not None:
                succ.append(tgt)
        for j in succ:
            if state[j] is None:
                state[j] = dict(st)
                work.append(j)
            else:
                merged = dict(state[j])
                changed = False
                for r, v in st.items():
                    if r not in merged or v[0] < merged[r][0]:
                        merged[r] = v
                        changed = True
                if changed:
                    state[j] = merged
                    work.append(j)
    if not ring_excuse:
        return [bad[i] for i in sorted(bad)]
    # Register-ring loops (gemm.hip fast_body): a slot re-requests the register set it published one pass earlier, and every feasible path
    # from that request back to the slot runs through the other slots' counted waits, which retire it.  The compiler, however, merges the
    # "k-tiles ran out" path and the normal path in front of the loop latch, so the control-flow graph has an infeasible edge from the end
    # of a slot straight back to its own start, and this path-insensitive analysis reports the slot's address arithmetic for the registers it
    # is about to reload.  Excused: a hit all of whose origin loads lie LATER in the same straight-line block (no branch target in between).
    # A copy or reuse of the OTHER set -- the one legitimately in flight, the round-3 BPTT bug class -- has its origin in another block.
    targets = {it[3] for it in ins if it[3] is not None}
    block_end = {}
    for i in range(n - 1, -1, -1):                                # last instruction index of the straight-line block containing i
        mn = ins[i][1]
        if i == n - 1 or mn.startswith(("s_cbranch", "s_branch", "s_endpgm")) or (i + 1) in targets:
            block_end[i] = i
        else:
            block_end[i] = block_end[i + 1]
    out = []
    for i in sorted(bad):
        addr, text, regs, origins = bad[i]
        if all(o in index and i <= index[o] <= block_end[i] for o in origins):
            continue
        out.append(bad[i])
    return out


def kernels_matching(want, ring_excuse=False):
    """-> {demangled name: [hazards]} for every kernel whose name contains one of `want`"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for i, co in enumerate(cm.code_objects()):
            path = os.path.join(d, f"k{i}.co")
            open(path, "wb").write(co)
            txt = subprocess.run([os.path.join(cm.LLVM_BIN, "llvm-objdump"), "-d", path], capture_output=True, text=True).stdout
            for m in re.finditer(r"^([0-9a-f]+) <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
                base, name, body = int(m.group(1), 16), m.group(2), m.group(3)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"^void ", "", dem).replace("mimrl::(anonymous namespace)::", "").split("(")[0]
                if want and not any(w in dem for w in want):
                    continue
                out[dem] = analyse(body, base, ring_excuse)
    return out


if __name__ == "__main__":
    ring = "--ring" in sys.argv
    res = kernels_matching([a for a in sys.argv[1:] if a != "--ring"], ring)
    for k in sorted(res):
        print(f"{len(res[k]):4d} in-flight register hazards  {k}")
        for addr, text, regs, org in res[k][:12]:
            print(f"       {addr:08x}  {text[:100]}   <- v{regs} loaded at {[hex(o) for o in org]}")

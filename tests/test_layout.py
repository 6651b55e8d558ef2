"""CPU tier: the C-ABI library loads, exports every symbol include/mimrl.h declares, and its parameter layout
agrees with the Python mirror (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from mimrl_amd import _lib, layout
from tests.golden.configs import CONFIGS, make_opt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mimrl.h")).read()
    declared = sorted(set(re.findall(r"\b(mimrl_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/mimrl.h but not exported"
    assert set(_lib.EXPORTS) == set(declared)
    assert lib.mimrl_abi_version() == 6


@pytest.mark.parametrize("name", list(CONFIGS))
def test_native_layout_matches_python(name):
    opt = make_opt(CONFIGS[name])
    cfg = _lib.make_cfg(opt, 768, 74, 35, bank_capacity=CONFIGS[name]["N"])
    entries, sizes = _lib.layout_entries(cfg)
    py_entries, py_sizes = layout.build_layout(opt, 768, 74, 35)
    assert len(entries) == len(py_entries)
    for (n, g, off, shape), e in zip(entries, py_entries):
        assert (n, off, tuple(shape)) == (e.name, e.offset, e.shape)
        assert g == (1 if e.group == "critic" else 0)
    assert sizes == (py_sizes["main"], py_sizes["critic"])


def test_parameter_counts_match_survey():
    """SURVEY.md Appendix B [probe of the reference]: main 1,083,479 / vmi 1,975,040 (concat 988,165) / vcmi 1,383,948."""
    opt = make_opt(CONFIGS["cfg1_sep"])
    ents, _ = layout.build_layout(opt, 768, 74, 35)
    main = sum(e.numel for e in ents if e.group == "main")
    vmi = sum(e.numel for e in ents if e.name.startswith("vmi"))
    vcmi = sum(e.numel for e in ents if e.name.startswith("vcmi"))
    assert (main, vmi, vcmi) == (1083479, 1975040, 1383948)
    opt = make_opt(CONFIGS["cfg1_cat"])
    ents, _ = layout.build_layout(opt, 768, 74, 35)
    assert sum(e.numel for e in ents if e.name.startswith("vmi")) == 988165


def test_bad_config_is_rejected_loudly():
    opt = make_opt(CONFIGS["tiny_sep"])
    opt.d_common = 256            # the reference itself cannot run this
    cfg = _lib.make_cfg(opt, 768, 74, 35)
    with pytest.raises(_lib.MimrlError):
        _lib.layout_entries(cfg)
    opt = make_opt(CONFIGS["tiny_sep"])
    opt.critic_type = "joint"     # VMI.py:44-45 raises NotImplementedError
    with pytest.raises(NotImplementedError):
        _lib.make_cfg(opt, 768, 74, 35)


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mimrl_amd.engine import HipEngine
    with pytest.raises(_lib.MimrlError):
        HipEngine(make_opt(CONFIGS["tiny_sep"]), 768, 74, 35)
    assert _lib.load().mimrl_device_check() < 0


def test_every_knob_is_in_the_table():
    """csrc/knobs.cpp is the ONE table of the native library's environment knobs: every knob("MIMRL_...") /
    knob_on / knob_int site names a registered knob, no raw getenv("MIMRL_...") is left outside the debug-knob accessor of common.h,
    and every registered name is read somewhere."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mimrl_amd", "csrc")
    table_src = open(os.path.join(csrc, "knobs.cpp")).read()
    table = set(re.findall(r'^\s*\{"(MIMRL_[A-Z0-9_]+)",', table_src, re.M))
    assert 20 <= len(table) <= 40     # (99 in round 5)
    used = set()
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.h")):
        if f.endswith("knobs.cpp"):
            continue
        s = open(f).read()
        assert 'getenv("MIMRL_' not in s, f
        used |= set(re.findall(r'knob(?:_on|_int)?\("(MIMRL_[A-Z0-9_]+)"', s))
    assert used <= table, sorted(used - table)
    assert table - used <= {"MIMRL_KNOBS"}, sorted(table - used)


def test_det_flush_rule_matches_the_sources():
    """Deterministic build (csrc/det.h): the accumulation table is flushed only behind launches that may have called acc_add.  The rule
    (det.hip: det_launch_accumulates) knows the kernels that never do BY NAME; this test derives, from the sources, every __global__ kernel
    that reaches acc_add -- directly or through a __device__ helper -- and checks that none of them matches a 'safe' substring."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mimrl_amd", "csrc")
    det = open(os.path.join(csrc, "det.hip")).read()
    safe = re.findall(r'"([^"]+)"', re.search(r"kSafe\[\] = \{(.*?)\};", det, re.S).group(1))
    assert len(safe) > 20
    accumulating = []
    for f in glob.glob(os.path.join(csrc, "*.hip")):
        src = open(f).read()
        src += "".join(open(os.path.join(csrc, h)).read() for h in re.findall(r'#include "([a-z_]+\.h)"', src) if h not in ("common.h", "det.h"))
        src = re.sub(r"__launch_bounds__\([^)]*\)", "", src)
        # top-level function bodies: "name(args) {" at the start of a definition ... up to the next line that is just "}"
        funcs = {}
        for m in re.finditer(r"(?m)^(?:template[^\n]*\n)?((?:__global__|__device__|static|inline)[^\n;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*(?:__attribute__\(\([^)]*\)\)\s*)?\{)", src):
            end = src.find("\n}\n", m.end())
            funcs[m.group(2)] = (m.group(1), src[m.end():end if end > 0 else len(src)])
        acc = {n for n, (_, body) in funcs.items() if "acc_add(" in body}
        changed = True
        while changed:
            changed = False
            for n, (_, body) in funcs.items():
                if n not in acc and any(re.search(r"\b%s\s*(?:<[^;{}]*>)?\s*\(" % re.escape(h), body) for h in acc):
                    acc.add(n); changed = True
        accumulating += [n for n in acc if "__global__" in funcs[n][0]]
    assert len(accumulating) >= 20, accumulating
    bad = [n for n in accumulating if any(s in n for s in safe)]
    assert not bad, bad

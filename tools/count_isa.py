#!/usr/bin/env python3
"""Count the gfx950 instructions of a kernel's loops (no GPU needed).

    python tools/count_isa.py                      # the phi pass at C2's shape, both sides
    python tools/count_isa.py 'row_sweep_kernel<16, 7>'
    python tools/count_isa.py --json out.json 'phi_pass_kernel<double, 8, 7, 2, 1>'

Compiles hgaprec_amd/csrc/hpf_capi.hip to gfx950 assembly (device only), finds
the kernels whose demangled name contains the given text, cuts each body into
basic blocks and reports, for every loop (a backward branch to a label), the
instruction mix of the blocks between the label and the branch.  The "batch
loop" of a phi pass is the innermost loop that carries the row gathers
(global_load_dwordx4); VERDICT r2 #1 set <= 90 VALU per batch for it.
"""
from __future__ import annotations

import argparse
import json
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "hgaprec_amd" / "csrc" / "hpf_capi.hip"


def assemble(src: Path = SRC, arch: str = "gfx950") -> str:
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        cmd = ["/opt/rocm/bin/hipcc", f"--offload-arch={arch}", "-O3", "-std=c++17", "-fno-gpu-rdc",
               "-S", "--cuda-device-only", "-o", str(out), str(src)]
        subprocess.run(cmd, check=True, cwd=src.parent)
        return out.read_text()


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                       stdout=subprocess.PIPE, check=True)
    return dict(zip(names, p.stdout.splitlines()))


def kernels(asm: str):
    """name -> list of lines of the function body"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                out[name] = cur
                cur = None
            else:
                cur.append(line)
    return out


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "other"


def analyse(body):
    # instruction stream with labels
    insts, labels = [], {}
    for line in body:
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) and not re.match(r"^\.LBB\d+_\d+:", s):
            if not re.match(r"^\.LBB\d+_\d+:", s):
                continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        insts.append((op, s))
    loops = []
    for pos, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= pos:
                loops.append((labels[tgt], pos, tgt))
    res = []
    for a, b, tgt in loops:
        ops = Counter(op for op, _ in insts[a:b + 1])
        cls = Counter()
        for op, c in ops.items():
            cls[classify(op)] += c
        inner = not any((a2 > a or b2 < b) and a2 >= a and b2 <= b for a2, b2, _ in loops if (a2, b2) != (a, b))
        f64 = sum(c for op, c in ops.items() if re.match(r"v_(fma|mul|add|fmac|max|min)_f64", op))
        res.append({
            "label": tgt, "instructions": b - a + 1, "innermost": inner, "classes": dict(cls),
            "valu": cls["VALU"], "valu_f64_math": f64,
            "v_cndmask": sum(c for op, c in ops.items() if op.startswith("v_cndmask")),
            "v_mov": sum(c for op, c in ops.items() if op.startswith("v_mov") or op.startswith("v_accvgpr")),
            "row_gathers": sum(c for op, c in ops.items() if op in ("global_load_dwordx4", "global_load_dwordx2",
                                                                    "global_load_lds_dwordx4", "buffer_load_dwordx4")),
            "top": ops.most_common(14),
        })
    return res, len(insts)


def meta(body):
    txt = "\n".join(body)
    out = {}
    for key in ("NumVgprs", "NumAgprs", "NumSgprs", "Occupancy", "ScratchSize", "LDSByteSize"):
        m = re.search(rf"; {key}: (\d+)", txt)
        if m:
            out[key] = int(m.group(1))
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", nargs="*", default=["phi_pass_kernel<double, 8, 7, 2, 0>", "phi_pass_kernel<double, 8, 7, 2, 1>"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--asm", default=None, help="use this .s instead of compiling")
    ap.add_argument("--src", default=None, help="compile this .hip instead of hpf_capi.hip (a file that instantiates a few kernels)")
    ap.add_argument("--save", default=None, help="keep the generated assembly here")
    a = ap.parse_args()
    asm = Path(a.asm).read_text() if a.asm else assemble(Path(a.src).resolve() if a.src else SRC)
    if a.save:
        Path(a.save).write_text(asm)
    ks = kernels(asm)
    dm = demangle(list(ks))
    # the resource summary follows .Lfunc_end: look it up in the whole text
    report = {}
    for mangled, body in ks.items():
        name = dm[mangled]
        if not any(p in name for p in a.pattern):
            continue
        loops, n = analyse(body)
        tail = asm[asm.find(".Lfunc_end", asm.find("\n" + mangled + ":")):]
        info = meta(tail[:6000].splitlines())
        report[name] = {"instructions": n, "resources": info, "loops": loops}
        print(f"== {name}\n   {n} instructions; {info}")
        for lp in loops:
            tag = "batch loop" if lp["innermost"] and lp["row_gathers"] else ("innermost" if lp["innermost"] else "outer")
            print(f"   loop {lp['label']:>10} [{tag}]: {lp['instructions']} instr  VALU {lp['valu']} "
                  f"(f64 math {lp['valu_f64_math']}, cndmask {lp['v_cndmask']}, mov {lp['v_mov']})  "
                  f"gathers {lp['row_gathers']}  {lp['classes']}")
    if a.json:
        Path(a.json).write_text(json.dumps(report, indent=1))
    return 0 if report else 1


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Static instruction table of one kernel instantiation, per phase, read from the gfx950 ISA.

Compiles a translation unit of gnn-builder_amd/csrc to device assembly (hipcc -S, --cuda-device-only) with
-DGNNB_ZF_MARK, which turns the kernel's phase stamps (ZF_PT(i) in k_stack_zf.hip) into `; ZFMARK i` comments, cuts the
kernel's text at those comments and at its labels, and counts per region and per basic block: MFMA, other VALU, SALU,
LDS, VMEM, SMEM, waits / barriers, branches.  Loops are reported with their back edge so that a region's dynamic count
can be formed as  sum(block count x trips).  No GPU needed.  ("build the per-phase static
instruction table from the ISA".)

    python tools/isa_table.py                      # k_gcn2_zf<RELU, 1, 8, 16 waves, 11 units, fp32, h1 == h0>, the BASELINE config 2 kernel
    python tools/isa_table.py --blocks             # + every basic block
    python tools/isa_table.py --json profiles/r06_c2_gcn2_isa_table.json
"""
from __future__ import annotations

import argparse
import json
import re
import subprocess
import sys
import tempfile
from collections import Counter, OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "gnn-builder_amd" / "csrc"
CLASSES = ("MFMA", "VALU", "SALU", "LDS", "VMEM", "SMEM", "SYNC", "BR")


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep", "s_endpgm")):
        return "SYNC"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "BR"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def compile_asm(unit: str, extra: list[str]) -> str:
    out = Path(tempfile.mkdtemp(prefix="isa_")) / (unit + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", f"-I{ROOT / 'include'}",
           f"-I{CSRC}", "--cuda-device-only", "-S", "-o", str(out), str(CSRC / (unit + ".hip"))] + extra
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.exit(proc.stderr)
    return out.read_text()


def kernel_text(asm: str, pattern: str) -> list[str]:
    lines = asm.split("\n")
    rx = re.compile(pattern)
    starts = [i for i, l in enumerate(lines) if l and not l.startswith(("\t", " ", ".", ";")) and l.split(":")[0] and rx.search(l.split(":")[0]) and l.rstrip().split(";")[0].rstrip().endswith(":")]
    if not starts:
        sys.exit(f"no kernel matches {pattern}")
    s = starts[0]
    e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[s:e + 1]


def parse(text: list[str]):
    """-> list of blocks: {label, mark (the last ZFMARK seen when the block starts), counts, ops, branches}"""
    blocks, cur, mark = [], None, "pre"

    def new(label):
        nonlocal cur
        cur = {"label": label, "mark": mark, "counts": Counter(), "targets": [], "n": 0, "ops": []}
        blocks.append(cur)

    new("entry")
    for l in text[1:]:
        t = l.strip()
        if not t:
            continue
        m = re.match(r";\s*ZFMARK\s+(\S+)", t)
        if m:
            mark = m.group(1)
            new(cur["label"] + "+" + mark)
            continue
        if t.startswith((";", "//")):
            continue
        t = t.split(";")[0].strip()
        if t.startswith(".") and not t.endswith(":"):
            continue
        if re.match(r"^[.\w$]+:$", t):
            new(t.split(":")[0])
            continue
        op = t.split()[0]
        c = classify(op)
        cur["counts"][c] += 1
        cur["n"] += 1
        cur["ops"].append(t.split(";")[0].strip())
        if c == "BR":
            cur["targets"].append(t.split()[1])
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="k_stack_zf")
    ap.add_argument("--kernel", default=r"k_gcn2_zfILi0ELi1ELi8ELi16ELi11ELi0ELb1E")
    ap.add_argument("--define", action="append", default=["GNNB_DEV_FAST", "GNNB_ZF_MARK"])
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--dump", help="write the kernel's assembly text here")
    ap.add_argument("--json")
    a = ap.parse_args()
    asm = compile_asm(a.unit, ["-D" + d for d in a.define])
    text = kernel_text(asm, a.kernel)
    if a.dump:
        Path(a.dump).write_text("\n".join(text))
    blocks = parse(text)
    index = {b["label"]: i for i, b in enumerate(blocks)}
    regions: "OrderedDict[str, Counter]" = OrderedDict()
    for b in blocks:
        regions.setdefault(b["mark"], Counter()).update(b["counts"])
    total = Counter()
    print(f"{'region':>8} " + " ".join(f"{c:>6}" for c in CLASSES))
    for r, c in regions.items():
        total.update(c)
        print(f"{r:>8} " + " ".join(f"{c[k]:>6}" for k in CLASSES))
    print(f"{'total':>8} " + " ".join(f"{total[k]:>6}" for k in CLASSES))
    loops = []
    for i, b in enumerate(blocks):
        for t in b["targets"]:
            j = index.get(t)
            if j is not None and j <= i:
                body = Counter()
                for k in range(j, i + 1):
                    body.update(blocks[k]["counts"])
                loops.append({"head": t, "tail": b["label"], "mark": blocks[j]["mark"], "blocks": i - j + 1,
                              "counts": {k: body[k] for k in CLASSES}})
    print("\nloops (back edges; counts = every block between head and tail, i.e. an upper bound for one trip):")
    for lp in loops:
        print(f"  [{lp['mark']:>4}] {lp['head']:>12} <- {lp['tail']:<12} blocks {lp['blocks']:>3}  " +
              " ".join(f"{k}={v}" for k, v in lp["counts"].items() if v))
    if a.blocks:
        print()
        for b in blocks:
            if b["n"]:
                print(f"  [{b['mark']:>4}] {b['label']:<16} " + " ".join(f"{k}={b['counts'][k]}" for k in CLASSES if b["counts"][k]) +
                      ("  -> " + ",".join(b["targets"]) if b["targets"] else ""))
    if a.json:
        Path(a.json).write_text(json.dumps({
            "kernel": a.kernel, "unit": a.unit, "defines": a.define,
            "regions": {r: {k: c[k] for k in CLASSES} for r, c in regions.items()},
            "total": {k: total[k] for k in CLASSES}, "loops": loops}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Loop structure of one kernel in an llvm-objdump listing: for every backward branch, the instruction mix of the
loop body (VALU, v_readlane / v_writelane = SGPR spills living in VGPR lanes, scratch_ = VGPR spills, DS, SALU).

    tools/kernel_resources.sh obj.o dev.co; llvm-objdump -d --no-show-raw-insn dev.co > dev.s
    python tools/isa_loops.py dev.s <mangled-name-substring> [min_lines]
"""
import re
import sys


def main():
    text = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    start = next(i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <", l) and want in l)
    end = next((i for i in range(start + 1, len(text)) if re.match(r"^[0-9a-f]+ <", text[i])), len(text))
    lines = text[start:end]
    base = int(lines[0].split()[0], 16)
    addr = {}
    for i, l in enumerate(lines):
        m = re.search(r"// ([0-9A-F]{12}):", l)
        if m:
            addr[int(m.group(1), 16) - base] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\S*\s+\d+\s.*\+0x([0-9a-f]+)>", l)
        if m:
            ti = addr.get(int(m.group(1), 16))
            if ti is not None and ti < i:
                loops.append((ti, i))

    def count(a, b, pat):
        return sum(1 for l in lines[a:b + 1] if re.search(pat, l))

    pats = {"valu": r"\tv_", "readlane": "v_readlane", "writelane": "v_writelane", "scratch": "scratch_", "ds": r"\tds_",
            "salu": r"\ts_", "vmem": r"\t(global|buffer|flat)_"}
    print(lines[0][:120])
    print("whole kernel:", len(lines), "lines", {k: count(0, len(lines) - 1, p) for k, p in pats.items()})
    for a, b in sorted(loops):
        if b - a + 1 >= least:
            print(f"loop {a}-{b} ({b - a + 1} lines)", {k: count(a, b, p) for k, p in pats.items()})


if __name__ == "__main__":
    main()

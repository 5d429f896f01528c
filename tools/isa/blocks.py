#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (timing aid for the sweep kernel).

usage: blocks.py listing.s kernel-substring [--min N]
Prints, per basic block: line, label, VALU f64 / VALU other / lane moves (v_readlane, v_writelane = SGPR spill
traffic) / transcendental / SALU / VMEM / LDS / SMEM counts, the s_waitcnt count and the branch target at its end.
"""
import re
import sys


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "v64" if "f64" in op or "_b64" in op or "lshl_add_u64" in op else "v"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_"):
        return "s"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    inside, blocks, cur = False, [], None
    for ln, line in enumerate(open(path), 1):
        if not inside:
            if re.match(r"^_Z\S*:", line) and key in line:
                inside = True
                cur = {"line": ln, "label": "entry", "n": {}, "br": []}
            continue
        m = re.match(r"^(\.LBB\S+):", line)
        if m:
            blocks.append(cur)
            cur = {"line": ln, "label": m.group(1), "n": {}, "br": []}
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if c == "br":
            cur["br"].append(t.split()[-1])
        if op == "s_endpgm":
            blocks.append(cur)
            break
    cols = ["v64", "v", "lane", "trans", "s", "vmem", "lds", "smem", "wait"]
    print(f"{'line':>6} {'label':<14}" + "".join(f"{c:>6}" for c in cols) + "  branches")
    tot = {c: 0 for c in cols}
    for b in blocks:
        n = sum(b["n"].get(c, 0) for c in cols)
        for c in cols:
            tot[c] += b["n"].get(c, 0)
        if n >= mn:
            print(f"{b['line']:>6} {b['label']:<14}" + "".join(f"{b['n'].get(c, 0):>6}" for c in cols) + "  " + " ".join(b["br"]))
    print(f"{'':>6} {'total':<14}" + "".join(f"{tot[c]:>6}" for c in cols))


if __name__ == "__main__":
    main()

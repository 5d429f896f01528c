"""pass-2 blocks (two v_rcp_f64 or a group of list stores) of one queue-kernel instantiation: instruction counts by kind
usage: python tools/isa/p2.py /tmp/isa/cur.s [f32x|f32|f64|none]"""
import re, sys
KEY = {"f32": "queue_kernelILb1ELi2ELb1ELb0E", "f32x": "queue_kernelILb1ELi3ELb1ELb0E", "f64": "queue_kernelILb1ELi1ELb1ELb0E",
       "none": "queue_kernelILb1ELi0ELb1ELb0E"}
txt = open(sys.argv[1]).read().split("\n")
key = KEY.get(sys.argv[2] if len(sys.argv) > 2 else "f32x")
start = [i for i, l in enumerate(txt) if re.match(r"^_Z\S*" + key + r"\S*:", l)][0]
end = [i for i in range(start, len(txt)) if txt[i].strip().startswith("s_endpgm")][0]
blocks = []; cur = ["entry", [], start]; blocks.append(cur)
for i in range(start, end):
    l = txt[i]
    if re.match(r"^\.LBB\S+:", l):
        cur = [l.split(":")[0], [], i]; blocks.append(cur); continue
    if l.strip() and not l.strip().startswith((";", ".")): cur[1].append(l.strip())
tot = 0
for b in blocks:
    ins = b[1]
    n_rcp = sum(1 for x in ins if x.startswith("v_rcp_f64"))
    n_st = sum(1 for x in ins if x.startswith("global_store"))
    if n_rcp >= 2 or (n_st >= 2 and len(ins) < 40):
        v = sum(1 for x in ins if x.startswith("v_")); s_ = sum(1 for x in ins if x.startswith("s_") and not x.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")))
        mov = sum(1 for x in ins if x.startswith("v_mov"))
        print(f"{b[2]:6d} {b[0]:12s} n={len(ins):3d} valu={v:3d} (mov {mov}) salu={s_:3d} rcp={n_rcp} st={n_st}")

# tools/step_chain.py <dir>: the kernel chain of single planning steps out of a rocprofv3 --kernel-trace of tools/rules_step_bench.py
# (median step, slowest step, 95th percentile: start, duration and the gap in front of every kernel)
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:28]
steps, cur = [], []
for s, e, n in rows:
    k = short(n)
    if k.startswith("fo_rays_kernel") and cur:
        steps.append(cur); cur = []
    cur.append((s, e, k))
steps.append(cur)
steps = [st for st in steps if any(k.startswith("fo_reduce") for _, _, k in st) and len(st) < 14]
span = lambda st: (st[-1][1] - st[0][0]) / 1e3
steps.sort(key=span)
print(len(steps), "steps; span p50 %.1f us, max %.1f us" % (span(steps[len(steps) // 2]), span(steps[-1])))
for st in (steps[len(steps) // 2], steps[-1], steps[-len(steps) // 20]):
    t0 = st[0][0]
    print("step span %.1f us:" % span(st))
    prev = t0
    for s, e, k in st:
        print("   %-28s start %6.1f  dur %6.1f  gap before %5.1f" % (k, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
        prev = e

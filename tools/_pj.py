import json, sys
d = json.loads(sys.stdin.readline())
c = d.get("config", {})
print(sys.argv[1], "step", round(d["ms_per_step"], 4), "kernel", round(d["roofline"]["kernel_ms"], 4),
      "small", round(c["small_batch"]["ms_per_step"], 4) if "small_batch" in c else None,
      "rules", (round(c["rules_step"]["ms_per_step"], 4), round(c["rules_step"]["ms_per_step_p50"], 4), round(c["rules_step"]["ms_per_step_max"], 4)) if "rules_step" in c else None,
      "parity", (d.get("parity") or {}).get("ok"))

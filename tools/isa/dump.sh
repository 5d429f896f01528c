#!/bin/bash
# hipcc -S of fo_sweep.hip into /tmp/isa/$1.s + register summary of the queue-kernel instantiations
set -e
mkdir -p /tmp/isa
out=/tmp/isa/${1:-cur}.s
shift || true
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I/root/repo/include -I/root/repo/frenetix-occlusion_amd/csrc \
  -S --cuda-device-only "$@" -o $out /root/repo/frenetix-occlusion_amd/csrc/fo_sweep.hip 2>/dev/null
python3 - $out <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.sgpr_spill_count: (\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count: (\d+)", txt, re.S):
    if "queue" in m.group(2):
        name = re.sub(r".*queue_kernelI(.*)EEvNS.*", r"\1", m.group(2))
        print(f"{name:<16} lds {m.group(1):>6}  sgpr_spill {m.group(3):>4}  vgpr {m.group(4):>4}  vgpr_spill {m.group(5):>3}")
PY

for v in "$@"; do
export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so
timeout 300 python -m pytest tests/test_sweep_gpu.py -m gpu -q -x 2>&1 | tail -2
python - <<'PY'
import sys, numpy as np
sys.path.insert(0,'frenetix-occlusion_amd'); sys.path.insert(0,'.')
import torch
from frenetix_occlusion import synthetic as S
from frenetix_occlusion.sweep import MetricSweep
from oracle import fo_oracle as O
traj, ag = S.make_batch(2000, 32, config_id=2)
sw = MetricSweep(S.VEHICLE_BMW320I, 0.1)
sw.set_agents(*[ag[k] for k in ("pos","yaw","v","cov","shape","raw_dims","type","len")])
out = sw.run(traj["x"],traj["y"],traj["theta"],traj["v"],mode="full"); torch.cuda.synchronize()
ref = O.sweep(traj, ag, S.VEHICLE_BMW320I, 0.1, nthreads=8)
l = out.lists.permute(3,1,0,2).cpu().numpy(); f=np.isfinite(ref["lists"])
print("max |lists - oracle| =", np.abs(l[f]-ref["lists"][f]).max())
PY
done

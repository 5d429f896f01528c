#!/usr/bin/env python3
"""Branch statistics of fo_sweep_queue_kernel on the bench workload, emulated in numpy (no GPU).

Follows the kernel's control flow per (tile of 64 trajectories, agent, timestep) and counts how often each wave-level
branch is taken -- the weights that turn the per-block instruction counts of tools/isa/blocks.py into a cost model.
Needs gpurun_out/bench_agents.npz (tools/dump_bench_batch.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np
from frenetix_occlusion import synthetic as S

M, T = 10000, 31
ag = np.load(os.path.join(ROOT, "gpurun_out", "bench_agents.npz"))
ego = ag["ego"]
order = sys.argv[1] if len(sys.argv) > 1 else "sampler"
traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 3, ego_pos=ego[:2], ego_yaw=float(ego[2]), order=order)
hlA, hwA, wb = S.VEHICLE_BMW320I[0] / 2, S.VEHICLE_BMW320I[1] / 2, S.VEHICLE_BMW320I[2]
x, y, th = traj["x"], traj["y"], traj["theta"]
ec, es = np.cos(th), np.sin(th)
nt = (M + 63) // 64
pad = nt * 64 - M


def tiles_any(b):          # [M] bool -> [nt] any over the 64 lanes of a tile
    return np.concatenate((b, np.zeros(pad, bool))).reshape(nt, 64).any(1)


def pt_box2(px, py, hl, hw):
    qx, qy = np.maximum(np.abs(px) - hl, 0), np.maximum(np.abs(py) - hw, 0)
    return qx * qx + qy * qy


def sat(ex, ey, c, s, px, py, pc, ps, hlB, hwB):
    cr, sr = pc * c + ps * s, ps * c - pc * s
    dx, dy = px - (ex + wb * c), py - (ey + wb * s)
    ax, ay = c * dx + s * dy, c * dy - s * dx
    ux, uy, wx, wy = hlB * cr, hlB * sr, -hwB * sr, hwB * cr
    bx, by = -(pc * dx + ps * dy), -(pc * dy - ps * dx)
    vx, vy, zx, zy = hlA * cr, -hlA * sr, hwA * sr, hwA * cr
    s1 = np.abs(ax) - (hlA + np.abs(ux) + np.abs(wx)); s2 = np.abs(ay) - (hwA + np.abs(uy) + np.abs(wy))
    s3 = np.abs(bx) - (hlB + np.abs(vx) + np.abs(zx)); s4 = np.abs(by) - (hwB + np.abs(vy) + np.abs(zy))
    lb = np.maximum(np.maximum(s1, s2), np.maximum(s3, s4))
    d2 = np.minimum.reduce([pt_box2(ax + a * ux + b * wx, ay + a * uy + b * wy, hlA, hwA) for a in (1, -1) for b in (1, -1)] +
                           [pt_box2(bx + a * vx + b * zx, by + a * vy + b * zy, hlB, hwB) for a in (1, -1) for b in (1, -1)])
    mm = np.where(lb > 0, np.rint(np.sqrt(d2) * 1000.0), 0.0)
    return lb, mm, dx * dx + dy * dy


cnt = dict(wave_t=0, near=0, need=0, exact=0, gate_t=0, gate_far=0, gate_any=0, ingate=0, lane_near=0, lane_need=0,
           lane_t=0, nonzero_cp=0, harm_t=0, lr4s_t=0)
A = len(ag["len"])
for k in range(A):
    L = int(ag["len"][k])
    if L <= 0:
        continue
    pos, yaw = ag["pos"][k], ag["yaw"][k]
    pc, ps = np.cos(yaw), np.sin(yaw)
    hlB, hwB = ag["raw_dims"][k] / 2
    hdev = ag["shape"][k, 0] / 2
    Rsum = np.hypot(hlA, hwA) + np.hypot(hlB, hwB)
    # probe
    c2 = (pos[None, :L, 0] - x[:, :L]) ** 2 + (pos[None, :L, 1] - y[:, :L]) ** 2
    tb = c2.argmin(1)
    r = np.arange(M)
    _, dce, _ = sat(x[r, tb], y[r, tb], ec[r, tb], es[r, tb], pos[tb, 0], pos[tb, 1], pc[tb], ps[tb], hlB, hwB)
    tdce = tb.copy()
    thr = (dce + 0.51) * 1e-3
    thr2, thrR2 = thr * thr, (thr + Rsum) ** 2
    gate_rows = np.zeros((nt, T), np.int32)       # in-gate lanes per (tile, iteration)
    for t in range(min(L, T)):
        lb, mm, cd2 = sat(x[:, t], y[:, t], ec[:, t], es[:, t], pos[t, 0], pos[t, 1], pc[t], ps[t], hlB, hwB)
        near = ~((dce == 0) & (t > tdce)) & (cd2 < thrR2)
        cnt["wave_t"] += nt; cnt["lane_t"] += M
        wn = tiles_any(near); cnt["near"] += wn.sum(); cnt["lane_near"] += near.sum()
        overlap = ~(lb > 0)
        need = near & (overlap | (lb * lb < thr2))
        cnt["need"] += tiles_any(need).sum(); cnt["lane_need"] += need.sum()
        cnt["exact"] += tiles_any(need & ~overlap).sum()
        upd = need & ((mm < dce) | ((mm == dce) & (t < tdce)))
        dce = np.where(upd, mm, dce); tdce = np.where(upd, t, tdce)
        thr = (dce + 0.51) * 1e-3
        thr2, thrR2 = thr * thr, (thr + Rsum) ** 2
        if t >= 1:
            rx, ry = x[:, t] - pos[t - 1, 0], y[:, t] - pos[t - 1, 1]
            d0 = rx * rx + ry * ry
            far = d0 <= (5.0 + hdev + 1e-6) ** 2
            cnt["gate_t"] += nt; cnt["gate_far"] += tiles_any(far).sum()
            dvx, dvy = pc[t] * hdev, ps[t] * hdev
            m2 = np.minimum(d0, np.minimum((rx - dvx) ** 2 + (ry - dvy) ** 2, (rx + dvx) ** 2 + (ry + dvy) ** 2))
            ing = far & (m2 <= 25.0)
            cnt["gate_any"] += tiles_any(ing).sum(); cnt["ingate"] += ing.sum()
            gate_rows[:, t] = np.concatenate((ing, np.zeros(pad, bool))).reshape(nt, 64).sum(1)
            # how many in-gate samples have every box product exactly zero (|u| >= 6 on one axis for all 9 combos)?
            sg = np.sqrt(ag["cov"][k, t - 1, 0, 0] if ag["cov"][k, t - 1, 0, 0] > 0 else 0.1) * np.sqrt(2.0)
            sgy = np.sqrt(ag["cov"][k, t - 1, 1, 1] if ag["cov"][k, t - 1, 1, 1] > 0 else 0.1) * np.sqrt(2.0)
            len3, offx, offy = 2 * hlA / 3, hlA / 3, hwA
            bxs, bys = np.abs(len3 * ec[:, t]), np.abs(len3 * es[:, t])
            zx = (np.abs(rx) - abs(dvx) - bxs - offx) >= 6 * sg
            zy = (np.abs(ry) - abs(dvy) - bys - offy) >= 6 * sgy
            cnt["nonzero_cp"] += (ing & ~(zx | zy)).sum()
    cnt["harm_t"] += nt * min(T - 1, L)
    for TCv in (8, 12, 16, 31):
        calls = 0
        for t0 in range(0, T, TCv):
            n = gate_rows[:, t0:t0 + TCv].sum(1)
            calls += ((n + 63) // 64).sum()
        cnt[f"process_calls_TC{TCv}"] = cnt.get(f"process_calls_TC{TCv}", 0) + int(calls)
    if ag["type"][k] in (0, 1, 2, 5, 6, 7, 9):
        cnt["lr4s_t"] += nt * min(T - 1, L)
w = cnt["wave_t"]
print({k_: int(v) for k_, v in cnt.items()})
for TCv in (8, 12, 16, 31):
    print(f"TC={TCv}: process() calls per (wave, t) {cnt[f'process_calls_TC{TCv}'] / cnt['wave_t']:.4f}  (full queues only: {cnt['ingate'] / 64 / cnt['wave_t']:.4f})")
print(f"order={order}  per (wave, t):  near {cnt['near'] / w:.3f}  need {cnt['need'] / w:.3f}  exact {cnt['exact'] / w:.3f}  "
      f"gate_far {cnt['gate_far'] / cnt['gate_t']:.3f}  gate_any {cnt['gate_any'] / cnt['gate_t']:.3f}  "
      f"in-gate lanes {cnt['ingate'] / cnt['lane_t']:.4f} (process calls per wave-t {cnt['ingate'] / 64 / w:.4f})  "
      f"lane near {cnt['lane_near'] / cnt['lane_t']:.3f} need {cnt['lane_need'] / cnt['lane_t']:.3f}  "
      f"in-gate with a non-saturated box {cnt['nonzero_cp'] / max(cnt['ingate'], 1):.3f}  lr4s share {cnt['lr4s_t'] / cnt['harm_t']:.3f}")

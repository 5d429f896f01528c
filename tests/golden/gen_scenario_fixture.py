#!/usr/bin/env python3
"""Extracts the geometry the hot path needs from the reference's example scenarios into small .npz fixtures
(tests/golden/scenario{1,2,3}_geometry.npz): lanelet bounds, obstacle rectangles + state lists, ego initial state.
Run in the build container (reads /root/reference/example_scenarios/*.xml with the stdlib-XML loader of this repo):

    python tests/golden/gen_scenario_fixture.py

The fixtures are data (numbers out of the CommonRoad XML files), not reference source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
from frenetix_occlusion import scenario as S  # noqa: E402

REF = "/root/reference/example_scenarios"


def pack(sc):
    d = {"dt": np.float64(sc.dt), "ego_initial": sc.ego_initial, "benchmark_id": np.array(sc.benchmark_id)}
    off = [0]
    left, right = [], []
    for ll in sc.lanelets:
        assert len(ll.left) == len(ll.right)
        left.append(ll.left)
        right.append(ll.right)
        off.append(off[-1] + len(ll.left))
    d["lanelet_id"] = np.array([ll.lanelet_id for ll in sc.lanelets], dtype=np.int64)
    d["lanelet_off"] = np.array(off, dtype=np.int64)
    d["lanelet_left"] = np.concatenate(left)
    d["lanelet_right"] = np.concatenate(right)
    nmax = max([len(ll.successors) for ll in sc.lanelets] + [len(ll.predecessors) for ll in sc.lanelets] + [1])
    suc = np.full((len(sc.lanelets), nmax), -1, dtype=np.int64)
    pre = np.full((len(sc.lanelets), nmax), -1, dtype=np.int64)
    adj = np.full((len(sc.lanelets), 4), -1, dtype=np.int64)   # adj_left, same_dir, adj_right, same_dir
    for i, ll in enumerate(sc.lanelets):
        suc[i, :len(ll.successors)] = ll.successors
        pre[i, :len(ll.predecessors)] = ll.predecessors
        if ll.adj_left is not None:
            adj[i, 0], adj[i, 1] = ll.adj_left, int(bool(ll.adj_left_same_direction))
        if ll.adj_right is not None:
            adj[i, 2], adj[i, 3] = ll.adj_right, int(bool(ll.adj_right_same_direction))
    d["lanelet_successors"], d["lanelet_predecessors"], d["lanelet_adjacent"] = suc, pre, adj
    soff = [0]
    states = []
    for ob in sc.obstacles:
        states.append(ob.states.reshape(-1, 4))
        soff.append(soff[-1] + len(ob.states))
    d["obstacle_id"] = np.array([ob.obstacle_id for ob in sc.obstacles], dtype=np.int64)
    d["obstacle_role"] = np.array([ob.role for ob in sc.obstacles])
    d["obstacle_type"] = np.array([ob.obstacle_type for ob in sc.obstacles])
    d["obstacle_dims"] = np.array([[ob.length, ob.width] for ob in sc.obstacles]).reshape(-1, 2)
    d["obstacle_t0"] = np.array([ob.initial_time_step for ob in sc.obstacles], dtype=np.int64)
    d["obstacle_initial"] = np.array([ob.initial for ob in sc.obstacles]).reshape(-1, 4)
    d["obstacle_state_off"] = np.array(soff, dtype=np.int64)
    d["obstacle_states"] = np.concatenate(states) if states else np.zeros((0, 4))
    inc = []
    for it in sc.intersections:
        for k, i in enumerate(it["incomings"]):
            for key, code in (("incoming", 0), ("right", 1), ("straight", 2), ("left", 3)):
                for lid in i[key]:
                    inc.append((it["id"], k, code, lid))
    d["intersection_rows"] = np.array(inc, dtype=np.int64).reshape(-1, 4)
    return d


if __name__ == "__main__":
    for k in (1, 2, 3):
        sc = S.load_commonroad_xml(os.path.join(REF, f"scenario{k}.xml"))
        out = os.path.join(HERE, f"scenario{k}_geometry.npz")
        np.savez_compressed(out, **pack(sc))
        print(out, os.path.getsize(out), "bytes;", len(sc.lanelets), "lanelets,", len(sc.obstacles), "obstacles")

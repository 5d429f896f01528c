"""ctypes binding of libfo_hip.so (C ABI: include/fo_hip.h).

The HIP library is the product path.  There is NO CPU fallback: if the library is missing, fails to load, or no
GPU is visible, every entry point raises.  PyTorch is used only as the owner of device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FO_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libfo_hip.so")  # env: tuning builds

FO_OK, FO_E_ARG, FO_E_UNSUPPORTED_COV, FO_E_HIP, FO_E_NOMEM, FO_E_STATE = 0, -1, -2, -3, -4, -5
NPF, NPI, NL, NC = 12, 4, 5, 16
LISTS_F64, LISTS_F32, LISTS_F32_EXACT = 0, 1, 2
LIST_FORMAT = {"f64": LISTS_F64, "f32": LISTS_F32, "f32x": LISTS_F32_EXACT}   # f32x: float64 arithmetic, float32 store

PF = {"dce": 0, "ttc": 1, "ttce": 2, "max_ego_risk": 3, "max_obst_risk": 4, "max_obst_harm_with_cp": 5,
      "max_ego_harm": 6, "max_obst_harm": 7, "max_collision_probability": 8, "be_decel": 9, "be_btn": 10}
PI = {"time_dce": 0, "max_obst_risk_index": 1, "cp_argmax": 2, "hr_valid": 3}
LST = {"cp": 0, "ego_harm": 1, "obst_harm": 2, "ego_risk": 3, "obst_risk": 4}
COST = {"wttc": 0, "min_dce": 1, "max_ego_risk_all": 2, "max_obst_risk_all": 3, "max_ego_harm_all": 4,
        "max_obst_harm_all": 5, "max_collision_probability_all": 6, "max_obst_harm_with_cp_all": 7, "min_ttce": 8,
        "argmin_dce": 9, "argmin_ttc": 10, "argmax_risk": 11, "safe": 12, "max_btn": 13}
METRIC_BITS = {"dce": 1, "cp": 2, "ttc": 4, "ttce": 8, "wttc": 16, "be": 32, "hr": 64}
TYPE_CODES = {"car": 0, "truck": 1, "bus": 2, "bicycle": 3, "pedestrian": 4, "priorityvehicle": 5,
              "parkedvehicle": 6, "train": 7, "motorcycle": 8, "taxi": 9, "unknown": 10}

# every symbol include/fo_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "fo_abi_version", "fo_build_id", "fo_create", "fo_destroy", "fo_last_error",
    "fo_sweep_configure", "fo_sweep_reserve", "fo_sweep_set_list_format", "fo_sweep_set_agents", "fo_sweep_run", "fo_sweep_check",
    "fo_sweep_last_launch", "fo_sweep_timing", "fo_sweep_timing_read", "fo_sweep_timing_read_each",
    "fo_scene_set_map", "fo_scene_share_map", "fo_scene_set_edge_lines", "fo_scene_set_routes", "fo_scene_map_info", "fo_scene_copy_raster", "fo_scene_fan", "fo_scene_visibility", "fo_scene_future_visibility", "fo_scene_spawn",
    "fo_scene_candidate_count", "fo_scene_set_topology", "fo_scene_spawn_rules", "fo_step_run", "fo_step_mirror_wait",
    "fo_scene_set_centerlines", "fo_scene_spawn_rule_agents", "fo_sweep_autotune", "fo_scene_set_shadow_length",
]


class Vehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("length", "width", "wb_rear_axle", "mass", "a_max")]


class HarmCoeff(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lr4s_const", "lr4s_speed", "lr4s_side", "lr4s_rear", "lr1s_const",
                                          "lr1s_speed", "ped_const", "ped_speed")]


class Thresholds(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("harm", "risk", "be", "cp", "ttc", "dce")]


class SpawnRuleParams(C.Structure):       # fo_spawn_rule_params_t
    _fields_ = ([(n, C.c_double) for n in ("ego_x", "ego_y", "ego_yaw", "ego_s", "ego_d", "s_threshold", "ped_width", "ped_length")] +
                [(n, C.c_int32) for n in ("intention", "win_i0", "win_i1", "behind_static", "behind_turn", "behind_dynamic",
                                          "max_static", "max_dynamic", "n_dynamic_plus1", "reserved_")])


class RuleAgentTypes(C.Structure):       # fo_rule_agent_types_t: index 0 Car, 1 Bicycle, 2 Pedestrian
    _fields_ = [(n, C.c_double * 3) for n in ("speed", "raw_l", "raw_w", "infl_l", "infl_w")]


SPAWN_CELLS, SPAWN_RULES, SPAWN_BOTH = 0, 1, 2
SPAWN_MODE = {"cells": SPAWN_CELLS, "rules": SPAWN_RULES, "both": SPAWN_BOTH}


class Step(C.Structure):       # fo_step_t (include/fo_hip.h): the arguments of one planning step's five stage calls
    _fields_ = [("n_rays", C.c_int32), ("polygon_footprint", C.c_int32), ("ego_yaw", C.c_double), ("fov_deg", C.c_double),
                ("r", C.c_double), ("d_dirs", C.c_void_p), ("d_rmax", C.c_void_p), ("d_half", C.c_void_p),
                ("ego_x", C.c_double), ("ego_y", C.c_double), ("head_x", C.c_double), ("head_y", C.c_double),
                ("full_circle", C.c_int32), ("exact_cells", C.c_int32), ("O", C.c_int32), ("d_edge_skip", C.c_void_p),
                ("d_ocorn", C.c_void_p), ("d_ocen", C.c_void_p), ("d_oflags", C.c_void_p),
                ("win_ix0", C.c_int32), ("win_iy0", C.c_int32), ("win_nx", C.c_int32), ("win_ny", C.c_int32),
                ("d_range", C.c_void_p), ("d_hit_id", C.c_void_p), ("d_ring", C.c_void_p), ("d_obst_vis", C.c_void_p),
                ("d_cls", C.c_void_p), ("d_occ_idx", C.c_void_p), ("d_n_occ", C.c_void_p),
                ("min_ahead", C.c_double), ("max_dist", C.c_double), ("all_occluded", C.c_int32), ("max_agents", C.c_int32),
                ("routes", C.c_int32), ("n_path", C.c_int32), ("T_agents", C.c_int32), ("type4", C.c_int32 * 4),
                ("speed4", C.c_double * 4), ("raw_l4", C.c_double * 4), ("raw_w4", C.c_double * 4),
                ("infl_l4", C.c_double * 4), ("infl_w4", C.c_double * 4), ("d_path", C.c_void_p), ("dt", C.c_double),
                ("var0", C.c_double), ("var_factor", C.c_double), ("d_cell", C.c_void_p), ("d_pos0", C.c_void_p),
                ("d_yaw0", C.c_void_p), ("d_n", C.c_void_p), ("d_pos", C.c_void_p), ("d_yaw", C.c_void_p), ("d_v", C.c_void_p),
                ("d_cov", C.c_void_p), ("d_shape", C.c_void_p), ("d_raw_dims", C.c_void_p), ("d_type", C.c_void_p),
                ("d_len", C.c_void_p), ("M", C.c_int32), ("T", C.c_int32), ("d_x", C.c_void_p), ("d_y", C.c_void_p),
                ("d_theta", C.c_void_p), ("d_vel", C.c_void_p), ("d_acc", C.c_void_p), ("d_cost", C.c_void_p),
                ("d_safe", C.c_void_p), ("d_pair_f", C.c_void_p), ("d_pair_i", C.c_void_p), ("d_lists", C.c_void_p),
                ("list_format", C.c_int32), ("spawn_mode", C.c_int32), ("n_path6", C.c_int32), ("max_rule_points", C.c_int32),
                ("d_path6", C.c_void_p), ("d_oyaw", C.c_void_p), ("d_odims", C.c_void_p), ("rule", SpawnRuleParams),
                ("rule_types", RuleAgentTypes), ("d_rule_points", C.c_void_p), ("d_n_rule_points", C.c_void_p),
                # ABI 12: the step's own host transfers (obstacle rows in, hit ids + visibility flags out)
                ("h_obstacles", C.c_void_p), ("d_obstacles", C.c_void_p), ("obstacles_bytes", C.c_int64),
                ("h_mirror", C.c_void_p), ("d_mirror", C.c_void_p), ("mirror_bytes", C.c_int64)]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfo_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libfo_hip.so or raise (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch ships its own libamdhip64; the process must end up with ONE HIP runtime, the one torch allocates with.
    # Loaded after torch, libfo_hip.so binds to that copy (same soname); loaded first it would pull in /opt/rocm's, torch
    # would then bring a second runtime, and fo_create would fail on a device the other runtime owns.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p
    lib.fo_abi_version.restype = C.c_int
    lib.fo_build_id.restype = C.c_char_p
    lib.fo_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.fo_destroy.argtypes = [vp]
    lib.fo_destroy.restype = None
    lib.fo_last_error.argtypes = [vp]
    lib.fo_last_error.restype = C.c_char_p
    lib.fo_sweep_configure.argtypes = [vp, C.POINTER(Vehicle), C.POINTER(HarmCoeff), C.POINTER(Thresholds),
                                       C.c_uint32, C.c_double]
    lib.fo_sweep_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.fo_sweep_set_list_format.argtypes = [vp, C.c_int]
    lib.fo_sweep_set_agents.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, ip, ip, vp]
    lib.fo_sweep_run.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, dp, vp]
    lib.fo_sweep_check.argtypes = [vp, vp]
    lib.fo_sweep_autotune.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, dp, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_double), vp]
    lib.fo_sweep_timing.argtypes = [vp, C.c_int]
    lib.fo_sweep_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.fo_sweep_timing_read_each.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    lib.fo_sweep_last_launch.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    D = C.c_double
    lib.fo_scene_set_map.argtypes = [vp, C.c_int, ip, dp, C.c_int, dp, D, D, dp, dp, ip]
    lib.fo_scene_share_map.argtypes = [vp, vp]
    lib.fo_scene_set_routes.argtypes = [vp, C.c_int, C.c_int, ip, ip, C.c_int, dp, dp, ip]
    lib.fo_scene_map_info.argtypes = [vp] + [C.POINTER(D)] * 3 + [C.POINTER(C.c_int)] * 3
    lib.fo_scene_copy_raster.argtypes = [vp, vp]
    lib.fo_scene_set_edge_lines.argtypes = [vp, C.c_int, ip]
    lib.fo_scene_fan.argtypes = [vp, C.c_int, D, D, D, C.c_int, dp, dp, dp, vp]
    lib.fo_scene_visibility.argtypes = ([vp, D, D, D, D, D, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, dp, dp, dp]
                                        + [C.c_int] * 4 + [dp] * 7 + [vp])
    lib.fo_scene_future_visibility.argtypes = ([vp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, dp, D, C.c_int, dp, dp, ip, ip]
                                               + [C.c_int] * 3 + [ip, dp, vp])
    lib.fo_scene_spawn.argtypes = ([vp, dp] + [C.c_int] * 4 + [D] * 6 + [C.c_int] * 3 + [ip] + [dp] * 5 + [C.c_int, dp, C.c_int]
                                   + [D] * 3 + [dp] * 12 + [vp])
    lib.fo_scene_candidate_count.argtypes = [vp, ip, vp]
    lib.fo_step_run.argtypes = [vp, C.POINTER(Step), vp]
    lib.fo_step_mirror_wait.argtypes = [vp]
    lib.fo_scene_set_topology.argtypes = [vp, C.c_int, dp, ip, ip, C.c_int, ip, ip, dp]
    lib.fo_scene_spawn_rules.argtypes = ([vp, dp] + [C.c_int] * 5 + [dp, C.c_int] + [dp] * 6 + [C.POINTER(SpawnRuleParams), C.c_int,
                                                                                          dp, ip, vp])
    lib.fo_scene_set_centerlines.argtypes = [vp, C.c_int, ip, dp]
    lib.fo_scene_set_shadow_length.argtypes = [vp, C.c_double]
    lib.fo_scene_spawn_rule_agents.argtypes = ([vp, C.c_int, dp, ip, C.c_int, C.POINTER(RuleAgentTypes), C.c_int, dp, C.c_int]
                                               + [D] * 3 + [dp] * 8 + [ip, ip, vp])
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("fo_destroy", "fo_last_error", "fo_build_id"):
            fn.restype = C.c_int
    _lib = lib
    return lib


_pyhost = False


def pyhost():
    """the CPython helper module ``_fo_pyhost`` (csrc/fo_pyhost.c: packs trajectory OBJECTS into the pinned staging buffer
    with the C API), or None when it has not been built -- the callers then use their numpy statement of the same loop.
    (Host glue between Python objects and host memory, not part of the C ABI and not a compute path.)"""
    global _pyhost
    if _pyhost is False:
        _pyhost = None
        try:
            import importlib.machinery
            import importlib.util
            import sysconfig
            path = os.path.join(os.path.dirname(_HERE), "lib", "_fo_pyhost" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
            if os.path.exists(path) and not os.environ.get("FO_NO_PYHOST"):
                loader = importlib.machinery.ExtensionFileLoader("_fo_pyhost", path)
                spec = importlib.util.spec_from_file_location("_fo_pyhost", path, loader=loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _pyhost = mod
        except Exception:       # a module built for another interpreter: the numpy path serves
            _pyhost = None
    return _pyhost


def build_id():
    """hash of the sources the loaded library was built from (fo_build_id)"""
    return load().fo_build_id().decode()


def current_stream(device_index=None):
    """raw HIP stream handle of torch's current stream.  ``torch.cuda.current_stream().cuda_stream`` walks several
    Python layers (device lookup, Stream object): ~9 us, five times per planning step; the private getter behind it takes
    a fraction of a microsecond."""
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device()
    try:
        return torch._C._cuda_getCurrentRawStream(device_index)
    except AttributeError:      # other torch build: the public, slower way
        return torch.cuda.current_stream(device_index).cuda_stream


def metric_mask(names):
    m = 0
    for n in names:
        if n not in METRIC_BITS:
            raise ValueError(f"unknown metric '{n}'")
        m |= METRIC_BITS[n]
    return m


def make_thresholds(d=None):
    d = d or {}
    return Thresholds(*[float("nan") if d.get(k) is None else float(d[k])
                        for k in ("harm", "risk", "be", "cp", "ttc", "dce")])


class Context:
    """Owns one fo_ctx (one ego vehicle on one GPU)."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.fo_create(C.byref(self._h), int(device))
        if rc != FO_OK:
            raise NativeError(rc, "fo_create failed (is a gfx950 GPU visible? there is no CPU fallback)")
        self.device = int(device)
        self.list_format = "f64"      # element type of the per-timestep lists the context writes (fo_sweep_set_list_format)

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.fo_destroy(h)
            h.value = None        # (no module globals here: __del__ may run while the interpreter shuts down)

    __del__ = close

    def _check(self, rc):
        if rc != FO_OK:
            raise NativeError(rc, self._lib.fo_last_error(self._h).decode())

    def call(self, name, *args):
        self._check(getattr(self._lib, name)(self._h, *args))

    def timing(self, enable=True, every=1):
        """HIP-event timing of the sweep kernel; every = k > 1 times every k-th launch only"""
        self.call("fo_sweep_timing", (max(int(every), 1) if enable else 0))

    def timing_read(self):
        ms, n = C.c_double(), C.c_int()
        self._check(self._lib.fo_sweep_timing_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_read_each(self, cap=1024):
        """durations (ms) of the timed launches, one by one"""
        buf, n = (C.c_double * cap)(), C.c_int()
        self._check(self._lib.fo_sweep_timing_read_each(self._h, buf, cap, C.byref(n)))
        return list(buf[:min(n.value, cap)])

    def last_launch(self):
        g, b, a = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.fo_sweep_last_launch(self._h, C.byref(g), C.byref(b), C.byref(a)))
        return {"grid": g.value, "block": b.value, "agents_per_wave": a.value}

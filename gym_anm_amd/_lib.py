"""ctypes binding of the C ABI (include/anm_mi355x.h) and the loader of the per-topology library.

The only library this module ever loads on its own is the hipcc-built
``gym_anm_amd/_build/libanm_<topology>.so``.  If it is missing and cannot be built, or if no
MI355X is visible when a model is created, a :class:`~gym_anm_amd.errors.HipExtensionError` is
raised: there is no CPU code path in the product.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import codegen
from . import errors as E

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class NetworkDesc(C.Structure):
    _fields_ = [
        ("n_bus", C.c_int32), ("n_dev", C.c_int32), ("n_branch", C.c_int32),
        ("base_mva", C.c_double), ("delta_t", C.c_double), ("lamb", C.c_double),
        ("bus_vmin", c_double_p), ("bus_vmax", c_double_p),
        ("br_from", c_int32_p), ("br_to", c_int32_p),
        ("br_series", c_double_p), ("br_shunt", c_double_p), ("br_tap", c_double_p), ("br_rate", c_double_p),
        ("dev_type", c_int32_p), ("dev_bus", c_int32_p),
        ("dev_pmin", c_double_p), ("dev_pmax", c_double_p), ("dev_qmin", c_double_p), ("dev_qmax", c_double_p),
        ("dev_qp", c_double_p), ("dev_tau", c_double_p), ("dev_rho", c_double_p),
        ("dev_soc_min", c_double_p), ("dev_soc_max", c_double_p), ("dev_eff", c_double_p),
    ]  # fmt: skip


class Dims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ["n_bus", "n_dev", "n_branch", "n_load", "n_gen", "n_des", "action_dim", "state_base_dim",
                 "full_dim", "const_doubles"]]  # fmt: skip


class EnvConfig(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("gamma", C.c_double), ("clip_e_loss", C.c_double), ("clip_penalty", C.c_double),
        ("obs_low", c_double_p), ("obs_high", c_double_p), ("series", c_double_p), ("period", C.c_int32),
    ]  # fmt: skip


class StepWs(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("n_doubles", C.c_int64), ("iter_cap", C.c_int32), ("mid_cap", C.c_int32)]


class BatchView(C.Structure):
    _fields_ = [("env_index", C.c_void_p)] + [(k, C.c_int32) for k in ["w_load", "w_gen", "w_set", "w_des", "w_action", "w_state",
                                                                      "w_exo", "w_aux", "w_full", "w_obs"]]  # fmt: skip


class MpcDims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ["planning_steps", "n_load", "n_gen", "n_des", "n_branch", "n_ctrl",
                                         "n_stage_vars", "n_stage_rows", "table_doubles", "angle_rows"]] + \
               [("angle_bound", C.c_double)]  # fmt: skip


class MpcOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int32), ("trace", C.c_void_p), ("angle_rows", C.c_int32)]


class SolverOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int32), ("precision", C.c_int32), ("handoff_after", C.c_int32)]


FULL_FIELDS = ["bus_p", "bus_q", "bus_v_magn", "bus_v_ang", "bus_i_magn", "bus_i_ang", "dev_p", "dev_q", "des_soc",
               "gen_p_max", "branch_p", "branch_q", "branch_s", "branch_i_magn", "branch_i_ang"]  # fmt: skip


class FullLayout(C.Structure):
    _fields_ = [(k, C.c_int32) for k in FULL_FIELDS + ["size"]]


SOLVE_F64, SOLVE_F32 = 0, 1
IMPL_THREAD, IMPL_RADIAL, IMPL_MESH = 0, 1, 2
HANDOFF_NEVER, HANDOFF_AUTO = -1, -2

_P = C.c_void_p  # raw device (or, for the test double, host) pointers are passed as integers

# name -> (restype, argtypes): every symbol include/anm_mi355x.h declares
ABI = {
    "anm_last_error": (C.c_char_p, []),
    "anm_topology_name": (C.c_char_p, []),
    "anm_topology_signature": (C.c_char_p, []),
    "anm_device_count": (C.c_int, []),
    "anm_model_create": (C.c_int, [C.POINTER(NetworkDesc), C.POINTER(C.c_void_p)]),
    "anm_model_destroy": (None, [C.c_void_p]),
    "anm_model_dims": (C.c_int, [C.c_void_p, C.POINTER(Dims)]),
    "anm_model_set_env": (C.c_int, [C.c_void_p, C.POINTER(EnvConfig)]),
    "anm_model_get_ybus": (C.c_int, [C.c_void_p, c_double_p]),
    "anm_step_ws_record_doubles": (C.c_int, []),
    "anm_model_set_impl": (C.c_int, [C.c_void_p, C.c_int32]),
    "anm_model_get_impl": (C.c_int, [C.c_void_p]),
    "anm_model_lanes_per_env": (C.c_int, [C.c_void_p]),
    "anm_model_full_layout": (C.c_int, [C.c_void_p, C.POINTER(FullLayout)]),
    "anm_transition_f64": (C.c_int, [C.c_void_p, C.c_int64] + [_P] * 11 + [C.POINTER(SolverOpts), _P]),
    "anm_reset_f64": (C.c_int, [C.c_void_p, C.c_int64, _P, _P, C.c_uint64, C.c_uint64] + [_P] * 10
                      + [C.POINTER(SolverOpts), _P]),
    "anm_sample_init_state_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    "anm_step_f64": (C.c_int, [C.c_void_p, C.c_int64] + [_P] * 13 + [C.c_int32, C.c_uint64, C.c_uint64, _P, _P,
                                                                      C.POINTER(StepWs), C.POINTER(SolverOpts), _P]),
    "anm_model_set_classes": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(NetworkDesc))]),
    "anm_model_set_class_obs_bounds": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_double_p]),
    "anm_model_bind_env_classes": (C.c_int, [C.c_void_p, _P, C.c_int64]),
    "anm_model_bind_state_same": (C.c_int, [C.c_void_p, _P]),
    "anm_model_bind_nr_diff": (C.c_int, [C.c_void_p, _P]),
    "anm_model_bind_nr_start": (C.c_int, [C.c_void_p, _P]),
    "anm_model_bind_view": (C.c_int, [C.c_void_p, C.POINTER(BatchView)]),
    "anm_model_obs_fusable": (C.c_int, [C.c_void_p]),
    "anm_model_set_obs": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p]),
    "anm_mpc_create": (C.c_int, [C.POINTER(NetworkDesc), C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_void_p)]),
    "anm_mpc_destroy": (None, [C.c_void_p]),
    "anm_mpc_dims_of": (C.c_int, [C.c_void_p, C.POINTER(MpcDims)]),
    "anm_mpc_get_tables": (C.c_int, [C.c_void_p, c_double_p]),
    "anm_mpc_solve_f64": (C.c_int, [C.c_void_p, C.c_int64] + [_P] * 8 + [C.POINTER(MpcOpts), _P]),
    "anm_mpc_act_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, C.c_int32] + [_P] * 8
                        + [C.POINTER(MpcOpts), _P]),
    "anm_gather_obs_f64": (C.c_int, [C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P, _P,
                                     _P]),
    "anm_time_step_launches": (C.c_int, [C.c_void_p, C.c_int64] + [_P] * 9 + [C.c_int32, C.c_uint64, C.c_uint64, _P, _P,
                                                                              C.POINTER(StepWs), C.POINTER(SolverOpts), _P,
                                                                              C.c_int32,
                                                                              C.POINTER(C.c_float)]),
}  # fmt: skip


def bind(cdll):
    """Attach argtypes / restypes; raises AttributeError if the library lacks a declared symbol."""
    for name, (res, args) in ABI.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


class Backend:
    """A bound per-topology library plus what the host layer needs to know about it."""

    def __init__(self, cdll, device_type, path):
        self.lib = bind(cdll)
        self.device_type = device_type  # "cuda" for the product; the test double says "cpu"
        self.path = path
        self.generic = False  # True: library compiled for another topology, lane-group kernel only
        self.size_class = None  # (MPC size-class libraries only: MpcBackend)

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.anm_last_error()
            raise E.HipExtensionError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def signature(self) -> str:
        return self.lib.anm_topology_signature().decode()


_CACHE = {}


def _is_tree(topo) -> bool:
    n_bus, branches, devices = topo
    if len(branches) != n_bus - 1 or n_bus - 1 > 64 or len(devices) > 64:
        return False
    adj = {i: [] for i in range(n_bus)}
    for f, t in branches:
        adj[f].append(t)
        adj[t].append(f)
    seen, todo = {0}, [0]
    while todo:
        for v in adj[todo.pop()]:
            if v not in seen:
                seen.add(v)
                todo.append(v)
    return len(seen) == n_bus


def load_for_topology(topo, impl=None) -> Backend:
    """Load the gfx950 library that serves ``topo``.

    * a library specialised for this topology (thread-per-environment kernels + the generic
      lane-group kernel) when it exists or when it is worth building (networks up to 12 buses);
    * otherwise, for networks that fit a wavefront (radial: impl "radial") or a workgroup of 512 lanes (any
      topology: impl "mesh"), any
      already-built library in *generic* mode: the lane-group kernels are table-driven and need no
      per-topology compilation;
    * otherwise build the specialised library with hipcc."""
    name = codegen.topology_name(topo)
    if name in _CACHE:
        return _CACHE[name]
    path = codegen.lib_path(name)
    fits_group = topo[0] - 1 <= 512 and len(topo[1]) <= 1024 and len(topo[2]) <= 512   # mesh::fits (anm_mesh.hpp)
    generic = (_is_tree(topo) and (impl == "radial" or (impl is None and topo[0] > 12))) or \
              (fits_group and (impl == "mesh" or (impl is None and topo[0] > 12)))
    if not os.path.exists(path) and generic:
        for cand in sorted(os.listdir(codegen.BUILD_DIR)) if os.path.isdir(codegen.BUILD_DIR) else []:
            # (kernel-tuning variants, ANM_BUILD_TAG: libanm_<topology>.<tag>.so, only among themselves)
            if cand.startswith("libanm_") and cand.endswith(codegen._tag() + ".so") and cand.count(".") == 1 + codegen._tag().count("."):
                gpath = os.path.join(codegen.BUILD_DIR, cand)
                if not codegen.library_is_fresh(gpath):  # built from other sources / another ABI revision
                    continue
                try:
                    be = Backend(C.CDLL(gpath), "cuda", gpath)
                except OSError:
                    continue
                be.generic = True
                return be
    if not os.path.exists(path) and topo[0] > 12 and impl is None and not generic:
        # compiling the thread-per-environment kernels for a network of this size takes hipcc minutes to hours and
        # yields a spilling kernel: only on explicit request
        raise E.UnsupportedNetworkError(
            "a %d-bus network with %d branches and %d devices fits neither lane-group kernel (at most 513 buses, "
            "1024 branches, 512 devices) and is too large for the thread-per-environment kernels; pass "
            "impl='thread' to compile them anyway" % (topo[0], len(topo[1]), len(topo[2])))
    # build_library returns at once when the library's content stamp matches this tree's sources; a
    # stale library (edited kernels, changed C ABI) is rebuilt, or refused when hipcc is unavailable:
    # calling an old binary through the new ctypes signatures would be silent garbage
    path = codegen.build_library(topo)
    try:
        cdll = C.CDLL(path)
    except OSError as ex:
        raise E.HipExtensionError("cannot load the gfx950 library %s: %s" % (path, ex)) from ex
    be = Backend(cdll, "cuda", path)
    if be.signature() != codegen.topology_signature(topo):
        raise E.HipExtensionError("library %s was built for another topology (%s)" % (path, be.signature()))
    _CACHE[name] = be
    return be


MPC_ABI = ("anm_last_error", "anm_topology_signature", "anm_mpc_create", "anm_mpc_destroy", "anm_mpc_dims_of", "anm_mpc_get_tables",
           "anm_mpc_solve_f64", "anm_mpc_act_f64")
MPC_FORECAST_CONSTANT, MPC_FORECAST_PERFECT = 1, 2


class MpcBackend:
    """The MPC-only library of a topology (``libmpc_<topology>.so``): the ``anm_mpc_*`` entry points alone."""

    def __init__(self, cdll, path):
        for name in MPC_ABI:
            res, args = ABI.get(name, (C.c_char_p, []))
            fn = getattr(cdll, name)
            fn.restype, fn.argtypes = res, args
        self.lib, self.path, self.device_type = cdll, path, "cuda"
        self.size_class = None   # name of the MPC size class this library was compiled for (codegen.MPC_CLASSES), if any

    check = Backend.check
    signature = Backend.signature


def _topology_counts(topo):
    n_bus, branches, devices = topo
    types = [t for t, _ in devices]
    return (types.count(codegen.LOAD), types.count(codegen.CLASSICAL) + types.count(codegen.RENEWABLE), types.count(codegen.STORAGE),
            n_bus, len(branches))


def load_mpc_for_topology(topo, size_class=None) -> MpcBackend:
    """The MPC kernel for a network whose step runs in generic mode (on the lane-group kernels of a library compiled for
    another topology).  ``k_mpc`` depends on the network through its SIZES only, so:

    * a library already built for exactly this topology (``libmpc_<topology>.so``) is used when it is there;
    * else the smallest precompiled SIZE CLASS that takes the network padded (``libmpc_class_<name>.so``,
      ``codegen.MPC_CLASSES``: up to 24 loads, 4 generators, 2 storage units, 42 buses, 44 branches) -- no compile;
    * else ``libmpc_<topology>.so`` is built (one hipcc run of csrc/anm_mpc_only.hip on first use).

    ``size_class``: True forces a size class (tests), False forbids it."""
    tname = codegen.topology_name(topo)
    name = "mpc:" + tname
    exact = codegen.lib_path(tname, mpc_only=True)
    cls = None if size_class is False else codegen.mpc_class_for(_topology_counts(topo))
    if size_class is True and cls is None:
        raise E.UnsupportedNetworkError("no MPC size class takes this network")
    # the network's own library when it is there and FRESH; a stale one (other sources, another ABI revision) is rebuilt
    # where hipcc exists -- and where it does not, the size class serves the network instead of failing: a machine
    # without a compiler is exactly what the classes are for
    have_exact = os.path.exists(exact) and (codegen.mpc_library_is_fresh(topo) or codegen.hipcc_path() is not None)
    use_class = cls is not None and (size_class is True or not have_exact)
    if use_class:
        name = "mpcclass:" + cls
    if name in _CACHE:
        return _CACHE[name]
    path = codegen.build_mpc_class(cls) if use_class else codegen.build_library(topo, mpc_only=True)
    try:
        cdll = C.CDLL(path)
    except OSError as ex:
        raise E.HipExtensionError("cannot load the gfx950 library %s: %s" % (path, ex)) from ex
    be = MpcBackend(cdll, path)
    if not use_class and be.signature() != codegen.topology_signature(topo):
        raise E.HipExtensionError("library %s was built for another topology (%s)" % (path, be.signature()))
    be.size_class = cls if use_class else None
    _CACHE[name] = be
    return be


def as_c(arr, dtype):
    a = np.ascontiguousarray(arr, dtype=dtype)
    ptr_t = c_double_p if dtype == np.float64 else c_int32_p
    return a, a.ctypes.data_as(ptr_t)


def network_desc(model):
    """anm_network_desc for a :class:`gym_anm_amd.model.NetworkModel` (keeps the arrays alive)."""
    keep = []

    def d(a):
        arr, p = as_c(a, np.float64)
        keep.append(arr)
        return p

    def i(a):
        arr, p = as_c(a, np.int32)
        keep.append(arr)
        return p

    def cplx(a):
        a = np.asarray(a, dtype=np.complex128)
        return d(np.stack([a.real, a.imag], axis=-1))

    desc = NetworkDesc(
        n_bus=model.N_bus, n_dev=model.N_device, n_branch=model.N_branch,
        base_mva=float(model.baseMVA), delta_t=float(model.delta_t), lamb=float(model.lamb),
        bus_vmin=d(model.bus_vmin), bus_vmax=d(model.bus_vmax),
        br_from=i(model.br_f), br_to=i(model.br_t),
        br_series=cplx(model.br_series), br_shunt=cplx(model.br_shunt), br_tap=cplx(model.br_tap),
        br_rate=d(model.br_rate),
        dev_type=i(model.dev_type), dev_bus=i(model.dev_bus),
        dev_pmin=d(model.dev_p_min), dev_pmax=d(model.dev_p_max), dev_qmin=d(model.dev_q_min),
        dev_qmax=d(model.dev_q_max), dev_qp=d(model.dev_qp), dev_tau=d(model.dev_tau), dev_rho=d(model.dev_rho),
        dev_soc_min=d(model.dev_soc_min), dev_soc_max=d(model.dev_soc_max), dev_eff=d(model.dev_eff),
    )  # fmt: skip
    return desc, keep

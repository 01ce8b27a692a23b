"""Batched ``ANMEnv``: the Gymnasium surface of the reference over ``num_envs`` environments.

Mirrors ``gym_anm/envs/anm_env.py``: same constructor arguments (plus ``num_envs`` / ``device`` /
solver options), same hooks for task designers (``init_state``, ``next_vars``,
``observation_bounds``, ``observation``), same attributes (``K``, ``gamma``, ``lamb``,
``delta_t``, ``simulator``, ``state_values``, ``state_N``, ``action_space``, ``obs_values``,
``observation_space``, ``observation_N``, ``terminated``, ``timestep``, ``state``, ``e_loss``,
``penalty``, ``costs_clipping``, ``pfe_converged``) and the same ``reset`` / ``step`` contracts,
with a leading ``num_envs`` dimension on every per-environment quantity:

* ``reset(seed=, options=) -> (obs[num_envs, O], {})``
* ``step(action[num_envs, A]) -> (obs, reward[num_envs], terminated[num_envs], truncated[num_envs], {})``

``step`` is ONE launch of the fused HIP kernel (next_vars lookup in series mode, set-point
projection, Newton-Raphson power flow, branch flows, reward + clipping, state and observation);
nothing is copied to the host and nothing synchronises.  Non-convergence of the power flow is not
an error: it sets ``terminated`` (anm_env.py:421), the environment then stays in the absorbing
zero state with reward 0 (anm_env.py:365-367) until it is reset -- explicitly through
``reset(options={"mask": ...})``, or by itself at the next ``step`` when ``autoreset=True``
(series-mode tasks; Gymnasium's "next step" vector autoreset).
"""

from __future__ import annotations

import ctypes as C
from copy import deepcopy

import numpy as np
import torch

from .. import _lib
from .. import errors as E
from ..model import STATE_VARIABLES
from ..simulator import BatchedSimulator, StateView, _stream_ptr
from ..spaces import Box, GymEnv


def check_env_args(K, delta_t, lamb, gamma, observation, aux_bounds, state_bounds):
    """Argument validation of ``gym_anm/envs/utils.py:7-122`` (same exceptions)."""
    if K < 0:
        raise E.ArgsError("The argument K is %d but should be >= 0." % (K))
    if delta_t <= 0:
        raise E.ArgsError("The argument delta_t is %.2f but should be > 0." % (delta_t))
    if lamb < 0:
        raise E.ArgsError("The argument lamb is %d but should be >= 0." % (lamb))
    if gamma < 0 or gamma > 1:
        raise E.ArgsError("The argument gamma is %.4f but should be in [0, 1]." % (gamma))
    if isinstance(observation, str) and observation == "state":
        pass
    elif isinstance(observation, list):
        for obs in observation:
            if len(obs) not in (2, 3):
                raise E.ObsSpaceError("The observation tuple {} should be a list with 2 or 3 elements.".format(obs))
            key, nodes = obs[0], obs[1]
            if key not in STATE_VARIABLES:
                raise E.ObsNotSupportedError(key, STATE_VARIABLES)
            if isinstance(nodes, str) and nodes == "all":
                pass
            elif key == "aux":
                for n in nodes:
                    if n >= K:
                        raise E.ObsSpaceError("Aux variable index {} is out of bound for {} aux variables.".format(n, K))
            elif isinstance(nodes, list):
                for n in nodes:
                    if n not in state_bounds[key]:
                        raise E.ObsSpaceError(
                            "Observation {} is not supported for device/branch/bus with ID {}.".format(key, n)
                        )
            else:
                raise E.ObsSpaceError()
            if len(obs) == 3 and obs[2] not in STATE_VARIABLES[key]:
                raise E.UnitsNotSupportedError(obs[2], STATE_VARIABLES[key], key)
    elif callable(observation):
        pass
    else:
        raise E.ArgsError(
            'The argument observation is of type {} but should be either a list, a callable, or the string "state".'.format(
                type(observation)
            )
        )
    if aux_bounds is not None and len(aux_bounds) != K:
        raise E.ArgsError(
            "The argument aux_bounds has length {} but the environment has K={} auxiliary variables.".format(
                len(aux_bounds), K
            )
        )


class BatchedANMEnv(GymEnv):
    def __init__(self, network, observation, K, delta_t, gamma, lamb, aux_bounds=None, costs_clipping=None, seed=None,
                 num_envs=1, device="cuda", tol=1e-5, max_iter=100, precision="f64", autoreset=False, series=None,
                 env_offset=0, impl=None, straggler_after="auto", straggler_mid="auto", handoff_after="auto", track_full=False,
                 fuse_observation=True, variants=None, env_variant=None, _backend=None):  # fmt: skip
        GymEnv.reset(self, seed=seed)
        self.K, self.gamma, self.lamb, self.delta_t = K, gamma, lamb, delta_t
        self.aux_bounds = aux_bounds
        if costs_clipping is None:
            c1, c2 = np.inf, np.inf
        else:
            c1 = np.inf if costs_clipping[0] is None else costs_clipping[0]
            c2 = np.inf if costs_clipping[1] is None else costs_clipping[1]
        self.costs_clipping = (c1, c2)
        self.num_envs = int(num_envs)
        self.autoreset = bool(autoreset)
        self.env_offset = int(env_offset)  # global index of environment 0 (sharded batches)

        self.simulator = BatchedSimulator(network, delta_t, lamb, num_envs=num_envs, device=device, tol=tol,
                                          max_iter=max_iter, precision=precision, impl=impl,
                                          handoff_after=handoff_after, variants=variants, env_variant=env_variant,
                                          _backend=_backend)  # fmt: skip
        sim = self.simulator
        self.device = sim.device
        check_env_args(K, delta_t, lamb, gamma, observation, aux_bounds, sim.state_bounds)

        self.state_values = self._expand_all_ids(
            [("dev_p", "all", "MW"), ("dev_q", "all", "MVAr"), ("des_soc", "all", "MWh"), ("gen_p_max", "all", "MW"),
             ("aux", "all", None)]
        )  # fmt: skip
        self.state_N = sum(len(s[1]) for s in self.state_values)
        lo, hi = sim.model.action_bounds()
        self.action_space = Box(low=lo, high=hi, dtype=np.float64)
        self.single_action_space = self.action_space

        self.obs_values = self._build_observation_space(observation)
        self.observation_space = self.observation_bounds()
        if self.observation_space is not None:
            self.observation_N = self.observation_space.shape[0]
        self.single_observation_space = self.observation_space

        # ---- device-resident per-environment state ------------------------------------------------
        E_, S = self.num_envs, self.state_N
        f64 = dict(dtype=torch.float64, device=self.device)
        self._state_buf = torch.zeros((E_, S), **f64)
        self._state_same = None  # uint8 [E]: 1 = the state row equals the obs row and was left unwritten (see `state`)
        self._state_obs = torch.zeros((E_, S), **f64)  # clip(state, state-space bounds), written by the kernel
        self.reward = torch.zeros(E_, **f64)
        self.e_loss = torch.zeros(E_, **f64)
        self.penalty = torch.zeros(E_, **f64)
        self._term_u8 = torch.zeros(E_, dtype=torch.uint8, device=self.device)
        self._term_bool = self._term_u8.view(torch.bool)  # zero-copy: the kernel only writes 0 / 1
        self.timestep = torch.zeros(E_, dtype=torch.int32, device=self.device)
        self._conv_u8 = torch.zeros(E_, dtype=torch.uint8, device=self.device)
        # convergence of the last power flow (see `pfe_converged`): a tensor from the start; after a step it is
        # `~terminated`, formed when somebody asks (step() itself launches nothing but the step kernel)
        self._conv_bool = torch.ones(E_, dtype=torch.bool, device=self.device)
        self._conv_stale = False
        self._reset_count = torch.zeros(E_, dtype=torch.int32, device=self.device)
        self._truncated = torch.zeros(E_, dtype=torch.bool, device=self.device)
        # key of the device-side sampler (reset(options={"sampler": "device"}), autoreset).  Unseeded
        # environments draw it from np_random (entropy-seeded in that case), so that two unseeded
        # environments -- e.g. the ranks of a sharded batch -- are not correlated; explicit seeds stay
        # deterministic.
        self.rng_seed = int(self.np_random.integers(2**62)) if seed is None else int(seed)
        self._act_low = torch.as_tensor(lo, **f64)
        self._act_high = torch.as_tensor(hi, **f64)
        self.check_actions = True

        # ---- environment constants for the kernels ---------------------------------------------------
        self._series = None if series is None else np.ascontiguousarray(series, dtype=np.float64)
        slo, shi = self._state_bounds_vectors()
        self._obs_is_state = self.obs_values is not None and self.obs_values == self.state_values
        if self._obs_is_state and self.observation_space is not None:
            slo, shi = np.asarray(self.observation_space.low, float), np.asarray(self.observation_space.high, float)
        cfg = _lib.EnvConfig(
            K=K, gamma=float(gamma), clip_e_loss=float(c1), clip_penalty=float(c2),
            obs_low=_lib.as_c(slo, np.float64)[1], obs_high=_lib.as_c(shi, np.float64)[1],
            series=None if self._series is None else self._series.ctypes.data_as(_lib.c_double_p),
            period=0 if self._series is None else int(self._series.shape[1]),
        )  # fmt: skip
        self._cfg_keep = (slo, shi)
        with sim._device_ctx():
            sim.backend.check(sim.backend.lib.anm_model_set_env(sim._handle, C.byref(cfg)), "anm_model_set_env")
        # parameter classes: the reference builds one environment per network, each clipping its observation to the
        # Box of ITS network (anm_env.py:193-233, 313-331).  For the "state" observation every class gets its own
        # bounds; `class_observation_bounds` has them ([n_classes, state_N] each).  (observation_space itself, one
        # Box for the batch, stays that of class 0; list-form observations keep one Box.)
        self.class_observation_bounds = None
        vms = getattr(sim, "variant_models", [sim.model])
        if len(vms) > 1 and self._obs_is_state and type(self).observation_bounds is BatchedANMEnv.observation_bounds:
            los, his = [slo], [shi]
            with sim._device_ctx():
                for k, vm in enumerate(vms[1:], start=1):
                    lo_k, hi_k = self._state_bounds_vectors(vm.state_bounds())
                    a_lo, p_lo = _lib.as_c(lo_k, np.float64)
                    a_hi, p_hi = _lib.as_c(hi_k, np.float64)
                    sim.backend.check(sim.backend.lib.anm_model_set_class_obs_bounds(sim._handle, k, p_lo, p_hi),
                                      "anm_model_set_class_obs_bounds")
                    los.append(lo_k)
                    his.append(hi_k)
            self.class_observation_bounds = (np.stack(los), np.stack(his))
        # list-form observations (anm_env.py:497-521): gathered, scaled and clipped inside the step kernel when
        # the library can (anm_model_set_obs), else by anm_gather_obs_f64 from the electrical-state dump
        self._gather = None  # (index, scale, low, high) device tensors
        self._obs_fused = False
        self._obs_buf = None
        if self.obs_values is not None and not self._obs_is_state:
            self._gather = self._build_gather(self.obs_values)
            lib = sim.backend.lib
            if fuse_observation and lib.anm_model_obs_fusable(sim._handle):
                idx, sc, lo_, hi_ = (t.cpu().numpy() for t in self._gather)
                a_i, p_i = _lib.as_c(idx, np.int32)
                (a_s, p_s), (a_l, p_l), (a_h, p_h) = (_lib.as_c(x, np.float64) for x in (sc, lo_, hi_))
                with sim._device_ctx():
                    sim.backend.check(lib.anm_model_set_obs(sim._handle, len(a_i), p_i, p_s, p_l, p_h), "anm_model_set_obs")
                self._obs_fused = True
                self._obs_buf = torch.zeros((E_, len(a_i)), dtype=torch.float64, device=self.device)
        # track_full: also dump the electrical state of every step, so that simulator.state /
        # simulator.pfe_converged follow the environment like the reference's (simulator.py:529-537)
        self.track_full = bool(track_full)
        self._need_full_reset = self._gather is not None or self.track_full
        self._need_full = (self._gather is not None and not self._obs_fused) or self.track_full
        self._after_step = False
        self._step_args = None
        self._reset_count_ptr = self._reset_count.data_ptr()
        # compact time index next to `state` (series mode, thread-per-environment family): enables the
        # coalesced-row step kernel
        self._aux_index = None
        if self._series is not None and K == 1 and sim.impl == "thread":
            self._aux_index = torch.zeros(E_, dtype=torch.int32, device=self.device)
        self._aux_index_ptr = None if self._aux_index is None else self._aux_index.data_ptr()
        # "state" observation on the fast path: obs = clip(state) is the state itself unless a bound bites, so
        # the kernel writes the state row only then and flags the rest (anm_model_bind_state_same)
        if self._aux_index is not None and self._obs_is_state and sim.backend.device_type == "cuda":
            self._state_same = torch.zeros(E_, dtype=torch.uint8, device=self.device)
            with sim._device_ctx():
                sim.backend.check(sim.backend.lib.anm_model_bind_state_same(sim._handle, self._state_same.data_ptr()),
                                  "anm_model_bind_state_same")
        # two-phase step (see anm_step_ws in include/anm_mi355x.h): the first launch stops after
        # `straggler_after` Newton iterations, a second launch continues the solves still running
        self._ws = None
        self._ws_ref = None
        if straggler_after == "auto":
            # measured on MI355X (scripts/handoff_sweep.py, profiles/r02_d_handoff_sweep.txt): at 65 536
            # environments the in-wave lane-group hand-over (anm_solver_opts.handoff_after) is the faster cure for
            # stragglers (92 vs 114 us); the two-launch step -- stragglers of the whole batch packed into a second
            # launch, 8 per wavefront on lane groups -- wins once the batch is a multiple of the 65 536 lanes of the
            # chip (reference cap: 131 072: 121 vs 138 us, 262 144: 146 vs 183 us, 524 288: 193 vs 287 us, 1 M: 318
            # vs 484 us; with a cap of 20 the in-wave hand-over wins or ties up to 524 288, 1 M: 230 vs 239 us)
            big = 131072 if int(max_iter) >= 50 else 1048576
            straggler_after = 6 if self.num_envs >= big else None
        rec = sim.backend.lib.anm_step_ws_record_doubles()
        if self._aux_index is not None and straggler_after and rec > 0 and int(straggler_after) < int(max_iter):
            # records for 1/16 of the batch (~1.5 % of the solves are still running after 6 iterations; a solve
            # that finds no record continues on a lane group of its own wavefront)
            n_rec = min(self.num_envs, max(4096, self.num_envs // 16))
            self._ws_buf = torch.zeros(8 + n_rec * rec + (n_rec + 1) // 2, dtype=torch.float64, device=self.device)
            # straggler_mid: where the first straggler launch leaves the solves still running to a second one (None:
            # one level).  Measured (profiles/r03_f_throughput.txt): the straggler launch is bound by the 94-trip
            # chain of a diverging solve up to 524 288 environments -- a second level only adds a launch there (132 /
            # 158 / 206 us against 121 / 146 / 194) -- and by instruction issue at 1 M, where repacking the long
            # runners 8 per wavefront pays (298 against 318 us): "auto" = two levels from 1 M environments on
            if straggler_mid == "auto":
                straggler_mid = 12 if self.num_envs >= 1000000 else None
            mid = -1 if straggler_mid is None else int(straggler_mid)
            self._ws = _lib.StepWs(self._ws_buf.data_ptr(), self._ws_buf.numel(), int(straggler_after), mid)
            self._ws_ref = C.byref(self._ws)
        self._opts_ref = C.byref(sim.opts)

    # ---- hooks for task designers (anm_env.py:158-191) -----------------------------------------------
    def init_state(self):
        """Sample initial states: return a ``[num_envs, state_N]`` array/tensor (reference layout)."""
        raise NotImplementedError

    def next_vars(self, s_t):
        """Return ``[num_envs, N_load + N_gen + K]``: load injections (MW), generator potentials (MW)
        and the next auxiliary variables, given the state batch ``s_t``.  Not used in series mode."""
        raise NotImplementedError

    def observation_bounds(self):  # anm_env.py:193-233
        if self.obs_values is None:
            return None
        lower, upper = [], []
        bounds = self.simulator.state_bounds
        for key, nodes, unit in self.obs_values:
            for n in nodes:
                if key == "aux":
                    if self.aux_bounds is not None:
                        lower.append(self.aux_bounds[n][0])
                        upper.append(self.aux_bounds[n][1])
                    else:
                        lower.append(-np.inf)
                        upper.append(np.inf)
                else:
                    lower.append(bounds[key][n][unit][0])
                    upper.append(bounds[key][n][unit][1])
        return Box(low=np.array(lower), high=np.array(upper), dtype=np.float64)

    # ---- spaces helpers (anm_env.py:497-549) ------------------------------------------------------------
    def _build_observation_space(self, observation):
        if isinstance(observation, str) and observation == "state":
            obs_values = deepcopy(self.state_values)
        elif isinstance(observation, list):
            obs_values = deepcopy(observation)
            for idx, o in enumerate(obs_values):
                if len(o) == 2:
                    obs_values[idx] = tuple(list(o) + [STATE_VARIABLES[o[0]][0]])
        elif callable(observation):
            obs_values = None
            self.observation = observation  # shadows the method below, like the reference (anm_env.py:511-514)
            self.observation_fn = observation
        else:
            raise E.ObsSpaceError()
        return self._expand_all_ids(obs_values)

    def _expand_all_ids(self, values):
        m = self.simulator.model
        if values is None:
            return None
        for idx, o in enumerate(values):
            if isinstance(o[1], str) and o[1] == "all":
                if "bus" in o[0]:
                    ids = list(m.bus_ids)
                elif "dev" in o[0]:
                    ids = list(m.dev_ids)
                elif "des" in o[0]:
                    ids = [m.dev_ids[k] for k in m.des_idx]
                elif "gen" in o[0]:
                    ids = [m.dev_ids[k] for k in m.gen_idx]
                elif "branch" in o[0]:
                    ids = list(m.branch_ids)
                elif o[0] == "aux":
                    ids = list(range(0, self.K))
                else:
                    raise E.ObsNotSupportedError(o[0], STATE_VARIABLES.keys())
                values[idx] = (o[0], ids, o[2])
        return values

    def _state_bounds_vectors(self, bounds=None):
        """Box bounds of the *state* vector, used when the observation is the state (``bounds``: the state bounds of
        another network of the same topology: a parameter class)."""
        lo, hi = [], []
        bounds = self.simulator.state_bounds if bounds is None else bounds
        for key, nodes, unit in self.state_values:
            for n in nodes:
                if key == "aux":
                    if self.aux_bounds is not None:
                        lo.append(self.aux_bounds[n][0])
                        hi.append(self.aux_bounds[n][1])
                    else:
                        lo.append(-np.inf)
                        hi.append(np.inf)
                else:
                    lo.append(bounds[key][n][unit][0])
                    hi.append(bounds[key][n][unit][1])
        return np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64)

    def _build_gather(self, values):
        """Index/scale/clip vectors that turn the kernel's full-state dump (+ aux) into the
        list-form observation (anm_env.py:562-592 with the units of simulator.py:559-616)."""
        sim, m = self.simulator, self.simulator.model
        index, scale = [], []
        for key, nodes, unit in values:
            for n in nodes:
                if key == "aux":
                    index.append(sim.full_dim + n)  # aux columns are appended to the dump
                    scale.append(1.0)
                    continue
                if key.startswith("bus"):
                    j = m.bus_ids.index(n)
                elif key.startswith("dev"):
                    j = m.dev_ids.index(n)
                elif key == "des_soc":
                    j = [m.dev_ids[k] for k in m.des_idx].index(n)
                elif key == "gen_p_max":
                    j = [m.dev_ids[k] for k in m.gen_idx].index(n)
                else:
                    j = m.branch_ids.index(tuple(n))
                index.append(sim.full_offsets[key] + j)
                kv_j = j
                if key.startswith("branch"):  # kA for branches is not defined by the reference (pu only)
                    kv_j = 0
                scale.append(sim.unit_scale(key, unit, kv_j))
        space = self.observation_space
        dev = self.device
        return (
            torch.as_tensor(index, dtype=torch.int32, device=dev),
            torch.as_tensor(scale, dtype=torch.float64, device=dev),
            torch.as_tensor(np.asarray(space.low, float), dtype=torch.float64, device=dev),
            torch.as_tensor(np.asarray(space.high, float), dtype=torch.float64, device=dev),
        )

    # ---- observation ---------------------------------------------------------------------------------------
    def observation(self, s_t):
        """``o_t = observation(s_t)``: clip(state variables, Box) (anm_env.py:313-331), batched."""
        if self._obs_is_state:
            return self._state_obs
        sim = self.simulator
        index, scale, low, high = self._gather
        n_obs = index.numel()
        if self._obs_buf is None:
            self._obs_buf = torch.zeros((self.num_envs, n_obs), dtype=torch.float64, device=self.device)
        with sim._device_ctx():
            rc = sim.backend.lib.anm_gather_obs_f64(
                self.num_envs, sim.full.shape[1], sim.full.data_ptr(), self._state_buf.shape[1], self.K,
                self._state_buf.data_ptr(), self._term_u8.data_ptr(), n_obs, index.data_ptr(), scale.data_ptr(),
                low.data_ptr(), high.data_ptr(), self._obs_buf.data_ptr(), _stream_ptr(self.device),
            )  # fmt: skip
        sim.backend.check(rc, "anm_gather_obs_f64")
        return self._obs_buf

    @property
    def state(self):
        """State vectors ``[num_envs, state_N]`` (anm_env.py:139-147).  On the fast path this is assembled on
        access: rows the kernel did not write because they equal the observation are taken from it."""
        if self._state_same is None:
            return self._state_buf
        return torch.where(self._state_same.view(torch.bool).unsqueeze(1), self._state_obs, self._state_buf)

    @property
    def terminated(self):
        return self._term_bool

    @property
    def pfe_converged(self):
        """Convergence of the last power flow of every environment (simulator.pfe_converged in the
        reference): that of the reset right after a reset; after a step an environment has converged iff
        it is not terminated (anm_env.py:421).  Formed on access: `step()` only marks it stale."""
        if self._conv_stale:
            self._conv_bool = ~self._term_bool
            self._conv_stale = False
        return self._conv_bool

    # ---- reset (anm_env.py:235-311) --------------------------------------------------------------------------
    def _launch_reset(self, init_state, mask_u8):
        sim = self.simulator
        with sim._device_ctx():
            rc = sim.backend.lib.anm_reset_f64(
                sim._handle, self.num_envs, None if init_state is None else init_state.data_ptr(),
                None if mask_u8 is None else mask_u8.data_ptr(), self.rng_seed, self.env_offset, self._reset_count_ptr,
                sim.soc.data_ptr(), self._state_buf.data_ptr(), self._state_obs.data_ptr(), self._conv_u8.data_ptr(),
                self._term_u8.data_ptr(), self.timestep.data_ptr(), sim.nr_iters.data_ptr(),
                sim.full.data_ptr() if (self._need_full_reset or self._need_full) else None, self._aux_index_ptr,
                C.byref(sim.opts), _stream_ptr(self.device),
            )  # fmt: skip
        sim.backend.check(rc, "anm_reset_f64")
        if self._state_same is not None:  # the reset kernel writes both rows
            if mask_u8 is None:
                self._state_same.zero_()
            else:
                self._state_same.mul_(1 - mask_u8)
        self._after_step = False
        # convergence per environment: the environments this reset touched report the reset's power flow, the others
        # keep what their last step left (not terminated <=> converged, anm_env.py:421)
        conv = torch.ne(self._conv_u8, 0)
        if mask_u8 is None:
            self._conv_bool = conv
        else:
            self._conv_bool = torch.where(mask_u8.ne(0), conv, self.pfe_converged)
        self._conv_stale = False
        if self._need_full_reset or self._need_full:
            sim.state = StateView(sim, sim.full)
            sim.pfe_converged = self._conv_bool

    def reset(self, *, seed=None, options=None):
        """Reset every environment (or those selected by ``options["mask"]``).

        ``options["init_state"]`` (``[num_envs, state_N]``) bypasses ``init_state()``; otherwise
        initial states are drawn with ``init_state()`` and redrawn (up to 100 times, like the
        reference) for the environments whose first power flow does not converge.
        """
        GymEnv.reset(self, seed=seed)
        if seed is not None:
            self.rng_seed = int(seed)
        options = options or {}
        if options.get("sampler") == "device":
            return self._reset_on_device(options.get("mask"))
        mask = options.get("mask")
        mask_u8 = None
        if mask is not None:
            mask_u8 = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        given = options.get("init_state")
        todo = mask_u8.clone() if mask_u8 is not None else torch.ones(self.num_envs, dtype=torch.uint8,
                                                                     device=self.device)  # fmt: skip
        n_max = 100
        for attempt in range(1, n_max + 1):
            s0 = given if given is not None else self.init_state()
            s0 = torch.as_tensor(s0, dtype=torch.float64, device=self.device)
            if s0.dim() == 1:
                s0 = s0.unsqueeze(0).expand(self.num_envs, -1)
            if s0.shape[1] != self.state_N:
                raise E.EnvInitializationError(
                    "Expected size of initial state s0 is %d but actual is %d" % (self.state_N, s0.shape[1])
                )
            self._launch_reset(s0.contiguous(), todo)
            todo = todo * (1 - self._conv_u8)
            if given is not None or not bool(todo.any()):
                break
            if attempt == n_max:
                raise E.EnvInitializationError(
                    "No non-terminal state found out of %d initial states for environment %s"
                    % (n_max, type(self).__name__)
                )
        if given is not None:
            # an explicit initial state that does not converge leaves that environment terminated
            sel = todo.ne(0)
            self._term_u8[sel] = 1
        touched = mask_u8.ne(0) if mask_u8 is not None else slice(None)
        self.e_loss[touched] = 0.0
        self.penalty[touched] = 0.0
        obs = self.observation(self.state)
        if self.observation_space is None:
            n = obs.shape[1]
            self.observation_space = Box(low=-np.ones(n) * np.inf, high=np.ones(n) * np.inf)
            self.observation_N = n
        return obs, {}

    def _reset_on_device(self, mask=None):
        """``reset(options={"sampler": "device"})`` (series-mode tasks): initial states are drawn inside
        the reset kernel by the counter-based RNG that also serves autoreset, keyed by
        ``(seed, env_offset + env, reset_count[env])``, instead of by ``init_state()`` on the host.
        Environments whose first power flow does not converge are redrawn, up to the reference's
        limit of 100 attempts (anm_env.py:266-289)."""
        if self._series is None:
            raise E.EnvInitializationError("the device sampler needs a series-mode task")
        if mask is None:
            todo = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
        else:
            todo = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        touched = todo.ne(0)
        for attempt in range(100):
            self._launch_reset(None, todo)
            todo = todo * (1 - self._conv_u8)
            if not bool(todo.any()):
                break
        else:
            raise E.EnvInitializationError(
                "No non-terminal state found out of 100 initial states for environment %s" % type(self).__name__
            )
        self.e_loss[touched] = 0.0
        self.penalty[touched] = 0.0
        return self.observation(self.state), {}

    def sample_init_state(self, raw=False):
        """The initial states ``reset(options={"sampler": "device"})`` / the autoreset would draw NOW for every
        environment -- ``ANM6Easy.init_state`` (anm6_easy.py:25-52) generalised to the series-mode task, keyed by
        ``(seed, env_offset + env, reset_count[env])`` -- without resetting anything: ``[num_envs, state_N]`` in the
        layout ``reset(options={"init_state": ...})`` takes.  ``raw=True``: also the Philox words behind each row
        (``int64 [num_envs, blocks, 4]`` holding uint32 values; anm_sample_init_state_f64)."""
        if self._series is None:
            raise E.EnvInitializationError("the device sampler needs a series-mode task")
        sim = self.simulator
        out = torch.zeros((self.num_envs, self.state_N), dtype=torch.float64, device=self.device)
        words = None
        if raw:
            n_blocks = 1 + (sim.N_non_slack_gen + sim.N_des + 1) // 2
            words = torch.zeros((self.num_envs, n_blocks, 4), dtype=torch.int32, device=self.device)
        with sim._device_ctx():
            rc = sim.backend.lib.anm_sample_init_state_f64(
                sim._handle, self.num_envs, self.rng_seed, self.env_offset, self._reset_count_ptr, out.data_ptr(),
                None if words is None else words.data_ptr(), _stream_ptr(self.device))
        sim.backend.check(rc, "anm_sample_init_state_f64")
        if raw:
            return out, words.to(torch.int64) & 0xFFFFFFFF
        return out

    # ---- step (anm_env.py:333-453) -----------------------------------------------------------------------------
    def _step_call(self, action_ptr, exo_ptr, aux_ptr):
        """One anm_step_f64 launch on torch's current stream (all buffer pointers are persistent)."""
        sim = self.simulator
        dev = self.device
        if dev.type == "cuda":
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            switch = torch.cuda.current_device() != dev.index
        else:
            stream, switch = None, False
        args = self._step_args
        if args is None:
            args = self._step_args = (
                sim.soc.data_ptr(), self._state_buf.data_ptr(), self._term_u8.data_ptr(), self.timestep.data_ptr(),
                (self._obs_buf if self._obs_fused else self._state_obs).data_ptr(), self.reward.data_ptr(),
                self.e_loss.data_ptr(), self.penalty.data_ptr(), sim.nr_iters.data_ptr(),
                sim.full.data_ptr() if self._need_full else None,
            )  # fmt: skip
        fn = sim.backend.lib.anm_step_f64
        if switch:
            with torch.cuda.device(dev):
                rc = fn(sim._handle, self.num_envs, action_ptr, exo_ptr, aux_ptr, *args, 1 if self.autoreset else 0,
                        self.rng_seed, self.env_offset, self._reset_count_ptr, self._aux_index_ptr, self._ws_ref, self._opts_ref,
                        stream)  # fmt: skip
        else:
            rc = fn(sim._handle, self.num_envs, action_ptr, exo_ptr, aux_ptr, *args, 1 if self.autoreset else 0,
                    self.rng_seed, self.env_offset, self._reset_count_ptr, self._aux_index_ptr, self._ws_ref, self._opts_ref,
                        stream)  # fmt: skip
        if rc != 0:
            sim.backend.check(rc, "anm_step_f64")

    def step(self, action):
        sim = self.simulator
        if not (isinstance(action, torch.Tensor) and action.dtype == torch.float64 and action.device == self.device):
            action = torch.as_tensor(action, dtype=torch.float64, device=self.device)
        if action.dim() == 1:
            action = action.unsqueeze(0)
        if action.shape != (self.num_envs, sim.dims.action_dim):
            raise AssertionError("Action %r (%s) invalid." % (tuple(action.shape), type(action)))
        if self.check_actions:  # anm_env.py:356-357 (one device reduction + sync; disable for throughput runs)
            ok = bool(((action >= self._act_low) & (action <= self._act_high)).all())
            assert ok, "Action %r (%s) invalid." % (action, type(action))
        if not action.is_contiguous():
            action = action.contiguous()
        exo_ptr = aux_ptr = None
        if self._series is None:
            v = torch.as_tensor(self.next_vars(self.state), dtype=torch.float64, device=self.device)
            expected = sim.N_load + sim.N_non_slack_gen + self.K
            if v.dim() != 2 or v.shape[1] != expected:
                raise E.EnvNextVarsError(
                    "Next vars vector has size %d but expected is %d" % (v.shape[-1] if v.dim() else 0, expected)
                )
            n_exo = sim.N_load + sim.N_non_slack_gen
            exo = v[:, :n_exo].contiguous()
            aux = v[:, n_exo:].contiguous()
            exo_ptr, aux_ptr = exo.data_ptr(), (aux.data_ptr() if self.K > 0 else None)
        self._step_call(action.data_ptr(), exo_ptr, aux_ptr)
        self._after_step = True
        self._conv_stale = True  # pfe_converged = ~terminated, formed on access (no launch, no allocation here)
        if self._need_full:
            sim.state = StateView(sim, sim.full)
            sim.pfe_converged = self.pfe_converged
        if self._obs_is_state:
            obs = self._state_obs
        elif self._obs_fused:
            obs = self._obs_buf  # written by the step kernel itself
        elif self.obs_values is None:
            obs = self.observation_fn(self.state)
            obs = torch.where(self._term_bool.unsqueeze(1), torch.zeros_like(obs), obs)  # anm_env.py:365-367, 442-446
        else:
            obs = self.observation(self.state)  # the gather kernel zeroes the rows of terminated environments
        return obs, self.reward, self._term_bool, self._truncated, {}

    def render(self, mode="human"):
        raise NotImplementedError()

    def close(self):
        pass

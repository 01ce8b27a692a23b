"""Environments over DIFFERENT networks in one batch, behind one Gymnasium-style surface.

The reference builds one ``ANMEnv`` per network (``gym_anm/envs/anm_env.py:79-156``,
``examples/custom_anm6.py:20``); a learner that trains over a family of grids holds a list of them.
Here the environments of all those networks live in ONE batch of padded rows -- ``obs [E, W]``,
``action [E, A]`` with ``W`` / ``A`` the widest network's widths -- and a step is one launch per
TOPOLOGY, each told through a batch view (``anm_model_bind_view``) which rows are its own and each
in the kernel family that suits its network (ANM6 on its thread-per-environment kernels, a 30-bus
feeder on the tree kernel, a meshed network on the general lane-group kernel), the launches on
separate HIP streams (they touch disjoint rows).  Nothing is gathered, copied or scattered around
them.

Scope: what one ``ANMEnv`` per network gives the reference's user (``anm_env.py:158-191, 497-521``) --
* series-mode tasks (the exogenous variables are periodic tables indexed by one auxiliary time index, like ``ANM6Easy``:
  ``anm6_easy.py:54-132``): device-side initial states and next-step autoreset (``anm_env.py:266-311`` with the
  counter-based sampler of ``csrc/anm_device.hpp``), nothing but launches per step;
* HOOK tasks: ``init_state`` / ``next_vars`` callables and ``K`` auxiliary variables of the task designer
  (``anm_env.py:158-191``), evaluated by the host on the task's rows at every step, like ``BatchedANMEnv``'s;
* the ``"state"`` observation or a LIST-form one per task (``anm_env.py:497-521``: gathered, scaled and clipped inside the
  task's step kernel through its view, ``anm_model_set_obs``).
Inside its row every environment uses the layout of its OWN network / observation from column 0 (state: ``dev_p, dev_q,
des_soc, gen_p_max, aux``; the action ``[P_gen.., Q_gen.., P_des.., Q_des..]``); the padding of a row is never touched.
A task on the thread-per-environment family keeps it for the ``"state"`` observation with ``K <= 1``; a list-form
observation or ``K > 1`` moves it to its lane-group family (the thread family's step through a view moves rows per lane:
no LDS rows to gather from, one auxiliary slot).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .. import errors as E
from ..simulator import _stream_ptr
from ..spaces import Box
from .anm_env import BatchedANMEnv


_SIDE_STREAMS = {}


def _side_streams(device):
    """Three side streams per device, made ONCE per process and shared by every mixed batch.  A device serves its streams from
    four hardware queues (ROCm's default), handed out in creation order: with the current stream that is one queue per launch
    of a four-topology step.  Streams made per object land on queues already in use as soon as a process builds a few objects,
    and two launches on one queue run one after the other (measured: the same batch 164 us per step with the first three
    streams of the process, 220-250 us with later ones, 236-246 us on one stream; profiles/r05_n_mixed_streams.txt)."""
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(3)]
    return _SIDE_STREAMS[key]


def plan_slots(durations, max_slots=4, slack=1.05):
    """Launches -> slots (slot 0: the current stream, the others: side streams), from the time each launch takes alone: the
    longest first, each into the least loaded slot that stays within ``slack`` x the longest launch, else into a new slot while
    there is one, else into the least loaded.  A fork / join pair costs ~20 us per side stream, so launches that fit behind
    each other inside the longest one share a stream (87 / 72 / 48 / 37 us -> {87}, {72}, {48, 37}).  Returns a list of lists of
    indices into ``durations``; slot 0 holds the longest launch."""
    order = sorted(range(len(durations)), key=lambda i: -durations[i])
    if not order:
        return [[]]
    limit = slack * durations[order[0]]
    slots, load = [], []
    for i in order:
        fits = [j for j in range(len(slots)) if load[j] + durations[i] <= limit]
        if fits:
            j = min(fits, key=lambda j: load[j])
        elif len(slots) < max_slots:
            slots.append([]); load.append(0.0)
            j = len(slots) - 1
        else:
            j = min(range(len(slots)), key=lambda j: load[j])
        slots[j].append(i)
        load[j] += durations[i]
    return slots


class _Task(BatchedANMEnv):
    """One network of the mix as a one-environment ``BatchedANMEnv``: spec validation, the model with its task constants
    (``anm_model_set_env``: observation Box, clipping, series; ``anm_model_set_obs``: a list-form observation), the spaces.
    It is never stepped itself."""

    def __init__(self, spec, device, seed, tol, max_iter, precision, impl):
        from ..model import NetworkModel

        observation = spec.get("observation", "state")
        if callable(observation):
            raise E.ArgsError("MixedBatchedANMEnv: a task's observation is \"state\" or a list (a callable observation is a "
                              "function of the state the caller can apply to `state` itself)")
        self.hooked = spec.get("series") is None
        dt, lamb = spec.get("delta_t", 0.25), spec.get("lamb", 100)
        if self.hooked:
            if not callable(spec.get("next_vars")) or not callable(spec.get("init_state")):
                raise E.ArgsError("MixedBatchedANMEnv: a task needs `series` (periodic tables) or the hooks `init_state` and `next_vars`")
            K, series, aux_bounds = int(spec.get("K", 0)), None, spec.get("aux_bounds")
            self._init_fn, self._next_fn = spec["init_state"], spec["next_vars"]
        else:
            series = np.ascontiguousarray(spec["series"], dtype=np.float64)
            K, aux_bounds = 1, spec.get("aux_bounds", np.array([[0, series.shape[1] - 1]]))
        if impl is None and (isinstance(observation, list) or K > 1):
            topo = NetworkModel(spec["network"], dt, lamb).topology()
            if topo[0] <= 12:   # (the default there is the thread-per-environment family)
                impl = "radial" if _lib._is_tree(topo) else "mesh"
        super().__init__(spec["network"], observation, K, dt, spec.get("gamma", 0.995), lamb,
                         aux_bounds=aux_bounds, costs_clipping=spec.get("costs_clipping", (None, None)),
                         seed=seed, num_envs=1, device=device, tol=tol, max_iter=max_iter, precision=precision, series=series, impl=impl,
                         straggler_after=None)  # fmt: skip
        self.listed = not self._obs_is_state
        if self.listed and not self._obs_fused:
            raise E.UnsupportedNetworkError("MixedBatchedANMEnv: the list-form observation of this task cannot be gathered inside its step "
                                            "kernel (too wide for the LDS rows of its kernel family)")

    def init_state(self):  # (host sampler of the one-environment object: unused)
        raise NotImplementedError


class MixedBatchedANMEnv:
    """``tasks``: one dict per distinct network -- ``network`` (the reference's network dict) and either ``series``
    (``[n_load + n_gen, period]`` MW: loads by device id, then the generators' potentials: a series-mode task) or the hooks
    ``init_state(n) -> [n, state_N]`` and ``next_vars(s_t [n, state_N]) -> [n, n_load + n_gen + K]`` with ``K`` (arrays or
    tensors; the reference's hooks for ``n`` environments of the task at once); optionally ``observation`` (``"state"`` or the
    reference's list form), ``delta_t``, ``gamma``, ``lamb``, ``aux_bounds``, ``costs_clipping``, ``impl``.  ``env_task[e]``:
    the task of environment ``e``, in any order.  Hook tasks cost a host call and a few indexed copies per step, and their
    environments are not reset by ``autoreset`` (which lives inside the series-mode kernels): once terminated they stay in the
    absorbing terminal state (observation 0, reward 0: ``anm_env.py:365-367``) until ``reset(options={"mask": ...})``.

    ``reset(seed=...)`` / ``step(action)`` follow ``gymnasium.vector`` semantics with next-step autoreset, like
    ``BatchedANMEnv``.  ``step()`` never synchronises (with ``check_actions = False``; the Box check is a device reduction
    the host waits for, like the reference's assert) and never allocates: it can be captured into a HIP graph from the
    first call.  With ``streams=True`` the launches of a step are packed into streams by ``plan_slots`` from the time each
    takes alone; those times are measured by ``tune_streams()`` -- two timed rounds of step launches under a uniformly
    random action on a snapshot of the batch, which is restored afterwards -- called by the first ``reset()`` of the
    object's life (a call that synchronises anyway) or by hand; ``tuned`` tells, ``launch_us`` keeps the result (``None``
    before, and for ever with ``streams=False`` or a single live task).  Before that the launches are dealt round robin.
    ``observation_space`` / ``action_space`` are per-environment padded Boxes (``[E, W]`` bounds; the padding columns are
    ``[0, 0]``); ``single_observation_spaces[k]`` / ``single_action_spaces[k]`` are those of task k."""

    def __init__(self, tasks, env_task, device="cuda", seed=None, tol=1e-5, max_iter=100, precision="f64", autoreset=False,
                 env_offset=0, streams=True):
        env_task = np.asarray(env_task, dtype=np.int64)
        if env_task.ndim != 1 or env_task.size == 0 or env_task.min() < 0 or env_task.max() >= len(tasks):
            raise ValueError("env_task must be a 1-D array of indices into `tasks`")
        self.num_envs = int(env_task.size)
        self.env_task = env_task
        self.autoreset = bool(autoreset)
        self.env_offset = int(env_offset)
        self.np_random = np.random.default_rng(seed)
        self.rng_seed = int(self.np_random.integers(2**62)) if seed is None else int(seed)
        self.tasks = [_Task(t, device, seed, tol, max_iter, precision, t.get("impl")) for t in tasks]
        self.device = self.tasks[0].device
        if self.device.type != "cuda":
            raise E.HipExtensionError("MixedBatchedANMEnv steps through batch views of the gfx950 kernels: it needs a GPU device")
        for t in self.tasks:  # the flags of anm_model_bind_state_same are indexed by launch slot: not with a view
            with t.simulator._device_ctx():
                t.simulator.backend.check(t.simulator.backend.lib.anm_model_bind_state_same(t.simulator._handle, None),
                                          "anm_model_bind_state_same")
        self.impls = [t.simulator.impl for t in self.tasks]
        E_, dev = self.num_envs, self.device
        self.state_N = [t.state_N for t in self.tasks]
        self.obs_N = [t.observation_N for t in self.tasks]          # == state_N for the "state" observation
        self.action_N = [t.simulator.dims.action_dim for t in self.tasks]
        # one row width for state and observation rows (the "state" observation's rows have the stride of the state rows)
        self.W = max(self.state_N + self.obs_N)
        self.A = max(1, max(self.action_N))
        self.W_des = max(1, max(t.simulator.N_des for t in self.tasks))
        f64 = dict(dtype=torch.float64, device=dev)
        self.state = torch.zeros((E_, self.W), **f64)
        self._obs = torch.zeros((E_, self.W), **f64)
        self._hooked = [k for k, t in enumerate(self.tasks) if t.hooked]
        self._listed = [k for k, t in enumerate(self.tasks) if t.listed]
        self.W_exo = max([1] + [self.tasks[k].simulator.N_load + self.tasks[k].simulator.N_non_slack_gen for k in self._hooked])
        self.W_aux = max([1] + [self.tasks[k].K for k in self._hooked])
        self._exo = torch.zeros((E_, self.W_exo), **f64) if self._hooked else None
        self._aux = torch.zeros((E_, self.W_aux), **f64) if self._hooked else None
        self._init = torch.zeros((E_, self.W), **f64) if self._hooked else None
        # a list-form observation after a RESET is gathered from the electrical-state dump of the reset (anm_reset_f64 writes
        # the "state" observation: into a scratch buffer for those tasks)
        self.W_full = max([1] + [self.tasks[k].simulator.full_dim for k in self._listed])
        self._full = torch.zeros((E_, self.W_full), **f64) if self._listed else None
        self._obs_scratch = torch.zeros((E_, self.W), **f64) if self._listed else None
        self.soc = torch.zeros((E_, self.W_des), **f64)
        self.reward, self.e_loss, self.penalty = (torch.zeros(E_, **f64) for _ in range(3))
        self._term_u8 = torch.zeros(E_, dtype=torch.uint8, device=dev)
        self.terminated = self._term_u8.view(torch.bool)
        self._conv_u8 = torch.zeros(E_, dtype=torch.uint8, device=dev)
        self.timestep = torch.zeros(E_, dtype=torch.int32, device=dev)
        self.nr_iters = torch.zeros(E_, dtype=torch.int32, device=dev)
        self._reset_count = torch.zeros(E_, dtype=torch.int32, device=dev)
        self._truncated = torch.zeros(E_, dtype=torch.bool, device=dev)
        # which rows belong to which task, and the views that tell the models
        self.env_index, self._views = [], []
        for k, t in enumerate(self.tasks):
            idx = torch.as_tensor(np.nonzero(env_task == k)[0], dtype=torch.int32, device=dev)
            self.env_index.append(idx)
            v = _lib.BatchView(idx.data_ptr(), 0, 0, 0, self.W_des, self.A, self.W, self.W_exo if t.hooked else 0,
                               self.W_aux if t.hooked else 0, self.W_full if t.listed else 0, self.W if t.listed else 0)
            self._views.append(v)
            sim = t.simulator
            with sim._device_ctx():
                sim.backend.check(sim.backend.lib.anm_model_bind_view(sim._handle, C.byref(v)), "anm_model_bind_view")
        # spaces: per environment, padded
        lo, hi = np.zeros((E_, self.W)), np.zeros((E_, self.W))
        alo, ahi = np.zeros((E_, self.A)), np.zeros((E_, self.A))
        for k, t in enumerate(self.tasks):
            rows = env_task == k
            lo[rows, : self.obs_N[k]], hi[rows, : self.obs_N[k]] = t.observation_space.low, t.observation_space.high
            alo[rows, : self.action_N[k]], ahi[rows, : self.action_N[k]] = t.action_space.low, t.action_space.high
        self.observation_space = Box(low=lo, high=hi, dtype=np.float64)
        self.action_space = Box(low=alo, high=ahi, dtype=np.float64)
        self.single_observation_spaces = [t.observation_space for t in self.tasks]
        self.single_action_spaces = [t.action_space for t in self.tasks]
        self._act_low, self._act_high = torch.as_tensor(alo, **f64), torch.as_tensor(ahi, **f64)
        # columns beyond an environment's own action width are padding: ignored by the kernels, and by the Box check
        pad = np.ones((E_, self.A), dtype=bool)
        for k in range(len(self.tasks)):
            pad[env_task == k, : self.action_N[k]] = False
        self._act_pad = torch.as_tensor(pad, device=dev)
        self.check_actions = True
        self._live = [k for k in range(len(self.tasks)) if self.env_index[k].numel()]
        self._streams = _side_streams(dev)[: len(self.tasks) - 1] if (streams and len(self.tasks) > 1) else None
        if self._streams is not None:
            self._stream_ptrs = [C.c_void_p(s.cuda_stream) for s in self._streams]
            self._fork, self._done = torch.cuda.Event(), [torch.cuda.Event() for _ in self._streams]
        self._step_calls = None
        # launches -> slots: round robin until the first steps have been timed (plan_slots), see step()
        n_slots = (len(self._streams) + 1) if self._streams is not None else 1
        self._slots = [self._live[i::n_slots] for i in range(n_slots)]
        self.launch_us = None      # {task: microseconds of its step launch alone} once tune_streams() has run
        self.tuned = not (self._streams is not None and len(self._live) > 1)   # nothing to tune: one stream or one task

    # ------------------------------------------------------------------------------------------------------------
    def _fan_out(self, launch):
        """``launch(k, stream_ptr)`` for every task with environments: spread over torch's current stream and the side streams
        (forked from and joined back into the current stream), or one after the other on the current stream.
        (Host time matters here: four launches with their events took longer to ISSUE than the longest kernel runs -- the
        events and stream handles are made once, the device context is entered once per call.)"""
        cur = torch.cuda.current_stream(self.device)
        live = self._live
        with torch.cuda.device(self.device):
            if self._streams is None or len(live) < 2:
                ptr = _stream_ptr(self.device)
                for k in live:
                    launch(k, ptr)
                return
            # self._slots: slot 0 is the current stream itself, slot i > 0 side stream i - 1.  No more than three side streams: a
            # device serves its streams from four hardware queues (ROCm's default), and two streams that share a queue run their
            # kernels one after the other.
            slots = self._slots
            self._fork.record(cur)
            for i in range(1, len(slots)):
                if not slots[i]:
                    continue
                s = self._streams[i - 1]
                s.wait_event(self._fork)
                for k in slots[i]:
                    launch(k, self._stream_ptrs[i - 1])
                self._done[i - 1].record(s)
            ptr = _stream_ptr(self.device)
            for k in slots[0]:
                launch(k, ptr)
            for i in range(1, len(slots)):
                if slots[i]:
                    cur.wait_event(self._done[i - 1])

    def _timed_launches(self, launch):
        """every live task's launch alone on the current stream, HIP events around each: microseconds per task (synchronises:
        tune_streams() only)"""
        ptr = _stream_ptr(self.device)
        evs = []
        with torch.cuda.device(self.device):
            for k in self._live:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(k, ptr)
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize(self.device)
        return [e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]

    def _task_rows_of(self, t, k, width):
        """rows of task k's environments out of a padded batch tensor, cut to the task's own width (a copy)"""
        return t.index_select(0, self.env_index[k].long())[:, :width].contiguous()

    def _launch_reset(self, mask_u8):
        # hook tasks: the task designer's init_state() for the rows that are (re)drawn -- all of the task's rows are asked for,
        # the mask decides which are used (anm_env.py:266-289 draws per environment)
        for k in self._hooked:
            t = self.tasks[k]
            n = int(self.env_index[k].numel())
            if n == 0:
                continue
            s0 = torch.as_tensor(t._init_fn(n), dtype=torch.float64, device=self.device)
            if s0.dim() != 2 or tuple(s0.shape) != (n, t.state_N):
                raise E.EnvInitializationError("Expected size of initial state s0 is %d but actual is %s" % (t.state_N, tuple(s0.shape)))
            self._init[self.env_index[k].long(), : t.state_N] = s0

        def go(k, stream):
            t = self.tasks[k]
            sim = t.simulator
            rc = sim.backend.lib.anm_reset_f64(
                sim._handle, int(self.env_index[k].numel()), self._init.data_ptr() if t.hooked else None, mask_u8.data_ptr(),
                self.rng_seed, self.env_offset, self._reset_count.data_ptr(), self.soc.data_ptr(), self.state.data_ptr(),
                (self._obs_scratch if t.listed else self._obs).data_ptr(), self._conv_u8.data_ptr(), self._term_u8.data_ptr(),
                self.timestep.data_ptr(), self.nr_iters.data_ptr(), self._full.data_ptr() if t.listed else None, None,
                C.byref(sim.opts), stream)  # fmt: skip
            sim.backend.check(rc, "anm_reset_f64")

        self._fan_out(go)
        # list-form observations of the rows just reset: gathered from the dump (anm_gather_obs_f64 on the task's rows)
        for k in self._listed:
            t = self.tasks[k]
            idx = self.env_index[k].long()
            n = int(idx.numel())
            if n == 0:
                continue
            sim = t.simulator
            index, scale, low, high = t._gather
            full = self._task_rows_of(self._full, k, sim.full_dim)
            st = self._task_rows_of(self.state, k, t.state_N)
            out = torch.zeros((n, t.observation_N), dtype=torch.float64, device=self.device)
            with sim._device_ctx():
                rc = sim.backend.lib.anm_gather_obs_f64(n, sim.full_dim, full.data_ptr(), t.state_N, t.K, st.data_ptr(), None,
                                                        t.observation_N, index.data_ptr(), scale.data_ptr(), low.data_ptr(),
                                                        high.data_ptr(), out.data_ptr(), _stream_ptr(self.device))
            sim.backend.check(rc, "anm_gather_obs_f64")
            sel = mask_u8.index_select(0, idx).ne(0)
            rows = idx[sel]
            self._obs[rows, : t.observation_N] = out[sel]

    def reset(self, *, seed=None, options=None):
        """Initial states drawn on the device (``ANM6Easy.init_state`` generalised to any series-mode task, keyed by
        ``(seed, env_offset + env, reset count)``) or by a hook task's ``init_state``; environments whose first power flow
        does not converge are redrawn, up to the reference's 100 attempts (``anm_env.py:266-289``).
        ``options={"mask": bool[E]}`` resets a subset."""
        if seed is not None:
            self.np_random = np.random.default_rng(seed)
            self.rng_seed = int(seed)
            self._reset_count.zero_()
        mask = (options or {}).get("mask")
        todo = (torch.ones(self.num_envs, dtype=torch.uint8, device=self.device) if mask is None
                else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous())
        touched = todo.ne(0)
        for _ in range(100):
            self._launch_reset(todo)
            todo = todo * (1 - self._conv_u8)
            if not bool(todo.any()):
                break
        else:
            raise E.EnvInitializationError("No non-terminal state found out of 100 initial states for %d environments" % int(todo.sum()))
        self.e_loss[touched] = 0.0
        self.penalty[touched] = 0.0
        self.reward[touched] = 0.0
        if not self.tuned:   # (reset synchronises anyway: the stream plan is made here, never inside step())
            self.tune_streams()
        return self._obs, {}

    def _mutable(self):
        return [self.state, self._obs, self.soc, self.reward, self.e_loss, self.penalty, self._term_u8, self._conv_u8, self.timestep,
                self.nr_iters, self._reset_count] + [t for t in (self._exo, self._aux) if t is not None]

    def tune_streams(self, rounds=2, seed=0):
        """Measure what every task's step launch takes alone and pack the launches into streams from that (``plan_slots``).
        Runs ``rounds`` timed rounds of step launches (the last one counts: the first also pays for code loading) under a
        uniformly random action in the action Box, on a snapshot of the batch that is restored afterwards: the batch is left
        exactly as it was.  Synchronises; not under stream capture.  Returns ``launch_us`` (None: nothing to tune)."""
        if self._streams is None or len(self._live) < 2:
            self.tuned = True
            return None
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("MixedBatchedANMEnv.tune_streams() times launches with a synchronize: call it (or reset()) before "
                               "capturing a step into a HIP graph")
        keep = [t.clone() for t in self._mutable()]
        g = torch.Generator(device=self.device).manual_seed(int(seed))
        action = self._act_low + (self._act_high - self._act_low) * torch.rand((self.num_envs, self.A), generator=g,
                                                                                dtype=torch.float64, device=self.device)
        go = self._step_launcher(action)
        us = None
        for _ in range(max(1, int(rounds))):
            us = self._timed_launches(go)
        for t, k in zip(self._mutable(), keep):
            t.copy_(k)
        self.launch_us = dict(zip(self._live, us))
        self._slots = [[self._live[i] for i in slot] for slot in plan_slots(us, len(self._streams) + 1)]
        self.tuned = True
        torch.cuda.synchronize(self.device)
        return self.launch_us

    def _step_launcher(self, action):
        """``launch(k, stream)`` of task k's step over its rows of ``action`` (everything but the action pointer, the seed and
        the stream is the same at every step: the argument lists are made once)"""
        if self._step_calls is None:
            self._step_calls = []
            for k, t in enumerate(self.tasks):
                sim = t.simulator
                self._step_calls.append((sim, sim.backend.lib.anm_step_f64, [
                    sim._handle, int(self.env_index[k].numel()), 0, None, None, self.soc.data_ptr(), self.state.data_ptr(),
                    self._term_u8.data_ptr(), self.timestep.data_ptr(), self._obs.data_ptr(), self.reward.data_ptr(), self.e_loss.data_ptr(),
                    self.penalty.data_ptr(), self.nr_iters.data_ptr(), None, 1 if self.autoreset else 0, 0, self.env_offset,
                    self._reset_count.data_ptr(), None, None, C.byref(sim.opts), None]))  # fmt: skip
        a_ptr, seed, auto = action.data_ptr(), self.rng_seed, 1 if self.autoreset else 0
        # hook tasks: next_vars(s_t) of the task designer on the task's rows (anm_env.py:380-391), scattered into the padded
        # exo / aux arrays their launches read
        for k in self._hooked:
            t = self.tasks[k]
            idx = self.env_index[k].long()
            if idx.numel() == 0:
                continue
            v = torch.as_tensor(t._next_fn(self._task_rows_of(self.state, k, t.state_N)), dtype=torch.float64, device=self.device)
            n_exo = t.simulator.N_load + t.simulator.N_non_slack_gen
            if v.dim() != 2 or v.shape[1] != n_exo + t.K:
                raise E.EnvNextVarsError("Next vars vector has size %d but expected is %d" % (v.shape[-1] if v.dim() else 0, n_exo + t.K))
            self._exo[idx, :n_exo] = v[:, :n_exo]
            if t.K > 0:
                self._aux[idx, : t.K] = v[:, n_exo:]
        exo_ptr = None if self._exo is None else self._exo.data_ptr()
        aux_ptr = None if self._aux is None else self._aux.data_ptr()

        def go(k, stream):
            sim, fn, args = self._step_calls[k]
            hooked = self.tasks[k].hooked
            args[2], args[15], args[16], args[22] = a_ptr, (0 if hooked else auto), seed, stream
            if hooked:
                args[3], args[4] = exo_ptr, (aux_ptr if self.tasks[k].K > 0 else None)
            rc = fn(*args)
            if rc != 0:
                sim.backend.check(rc, "anm_step_f64")

        return go

    def step(self, action):
        """``action [E, A]``: row ``e`` holds environment ``e``'s own action vector from column 0, the columns beyond its
        width are ignored (by the Box check too).  Returns ``(obs [E, W], reward [E], terminated [E], truncated [E], info)``:
        tensors the next call overwrites.  Launches only: no synchronize, no allocation (``check_actions = False``)."""
        if not (isinstance(action, torch.Tensor) and action.dtype == torch.float64 and action.device == self.device):
            action = torch.as_tensor(action, dtype=torch.float64, device=self.device)
        if action.shape != (self.num_envs, self.A):
            raise AssertionError("Action %r (%s) invalid." % (tuple(action.shape), type(action)))
        if self.check_actions:  # anm_env.py:356-357 (one device reduction + sync; disable for throughput runs / graph capture)
            ok = bool((((action >= self._act_low) & (action <= self._act_high)) | self._act_pad).all())
            assert ok, "Action outside the action space of its environment."
        if not action.is_contiguous():
            action = action.contiguous()
        self._fan_out(self._step_launcher(action))
        return self._obs, self.reward, self.terminated, self._truncated, {}

    def task_rows(self, k):
        """indices (int64 tensor) of the environments of task ``k``"""
        return self.env_index[k].long()

    def close(self):
        pass

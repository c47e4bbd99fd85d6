"""ComposeAdversarialTransformSolver: N gradient-ascent steps over a chain of adversarial transforms
(reference: advchain/augmentor/adv_compose_solver.py:11-538; same public methods, arguments and control
flow, including the reference's quirks -- see DESIGN.md "Quirks kept").

What differs under the hood (results are unchanged, SURVEY §2.3 "Redundancy"):
  * the validity mask of the forward o backward warp is computed on ONE channel, without autograd
    (its gradient is exactly zero for 'mse'/'contour'), and re-uses the deformation fields of the data /
    prediction path instead of recomputing two more DemonsCompose per step;
  * no ``torch.cuda.empty_cache()`` inside the loop (5 allocator flushes per step in the reference);
  * optional batch sharding over a ``torch.distributed`` process group (RCCL on ROCm): every rank owns a
    contiguous slice of the batch; only scalars are exchanged (loss value / NaN guard, 3D step-count norm,
    intensity range, anatomy check) and every normaliser uses the GLOBAL batch size.
"""
import logging
import math

import torch

from .. import ops
from ..common.loss import calc_segmentation_consistency
from ..common.utils import _disable_tracking_bn_stats, _fix_dropout
from .adv_affine import AdvAffine
from .adv_bias import AdvBias
from .adv_morph import AdvMorph
from .adv_noise import AdvNoise

_NATIVE = (AdvNoise, AdvBias, AdvMorph, AdvAffine)


def _native_update(t):
    """True when `t` steps its parameters with one of the built-in update methods -- the ones that hand `t._gate` to the
    gated update kernels.  A subclass that overrides optimize_parameters() may never look at the gate: it gets the
    literal host-side NaN check of the reference instead (adv_compose_solver.py:343-347)."""
    return any(isinstance(t, cls) and type(t).optimize_parameters is cls.optimize_parameters for cls in _NATIVE)


class ComposeAdversarialTransformSolver(object):
    """apply a chain of transformation"""

    def __init__(self, chain_of_transforms=[], divergence_types=['mse', 'contour'],
                 divergence_weights=[1.0, 0.5], use_gpu=True, debug=False, if_norm_image=False,
                 min_intensity=None, max_intensity=None, is_gt=False, process_group=None, hip_graph=False,
                 deterministic=None):
        self.chain_of_transforms = chain_of_transforms
        self.use_gpu = use_gpu
        self.debug = debug
        self.divergence_weights = divergence_weights
        self.divergence_types = divergence_types
        self.require_bi_loss = self.if_contains_geo_transform()
        self.if_norm_image = if_norm_image
        self.min_intensity = min_intensity
        self.max_intensity = max_intensity
        self.is_gt = is_gt
        self.class_weights = None
        self.process_group = process_group     # extension: batch-sharded replicas
        # extension: the whole-batch size when sharded (the sum of the ranks' batches).  None: asked of the group at the start
        # of every call (one all-reduce and one host read-back per call); set it when every call has the same global batch
        self.global_batch = None
        self.device_nan_guard = True           # NaN guard of the ascent loop on the device (False: a host read-back per step)
        # extension: replay the ascent loop of adversarial_training (initial prediction + the n_iter ascent steps) as ONE
        # hipGraph once its launch sequence has been recorded (see _graphed_ascent); off by default -- the user's model
        # is captured with it, which needs a model without host-side control flow or side effects
        self.hip_graph = hip_graph
        # extension (round 6): bit-reproducible results run to run.  None: follow torch.are_deterministic_algorithms_enabled()
        # (what a user of the reference would have set); True / False: this solver's own choice.  See ops.set_deterministic
        self.deterministic = deterministic
        self.hip_graph_record_calls = 3        # ordinary calls recorded before the capture
        self.hip_graph_margin = 1.3            # recorded maxima x margin pick the frozen kernel selection (ops.LaunchPlan)
        self._graphs = {}
        self.graph_stats = {"recorded": 0, "captures": 0, "replays": 0, "violations": 0, "refused": 0}
        self._global_batch = None
        self._local_steps = None

    # ------------------------------------------------------------------------------- sharding helpers
    def _dist(self):
        if self.process_group is None:
            return None
        import torch.distributed as dist
        return dist

    def _all_reduce_(self, t, op="sum"):
        dist = self._dist()
        if dist is not None:
            ops_ = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}
            dist.all_reduce(t, op=ops_[op], group=self.process_group)
        return t

    def _resolve_global_batch(self, n_local, device):
        if self.process_group is None:
            self._global_batch = None
            return
        if self.global_batch is not None:
            # told by the caller (a fixed per-rank batch is the normal case): no collective and, more to the point, no host
            # read-back at the start of every call -- the `.item()` below waits for an all-reduce that itself waits for
            # everything the rank has queued, i.e. it drains the rank's queue once per call
            self._global_batch = int(self.global_batch)
        else:
            t = torch.tensor([float(n_local)], device=device)
            self._global_batch = int(self._all_reduce_(t).item())
        for tr in self.chain_of_transforms:
            if isinstance(tr, AdvMorph):
                tr.process_group = self.process_group

    def _global_value(self, local):
        """Whole-batch value of a per-shard partial loss; keeps the autograd path of the local part."""
        if self.process_group is None:
            return local
        if self._local_steps is not None:        # inside a capture (see _shardable_capture): no collective, the value is checked after the replay
            self._local_steps.append(local.detach().reshape(1))
            return local
        total = self._all_reduce_(local.detach().clone())
        return local + (total - local.detach())

    def _apply_deterministic(self, data):
        """The library's process-wide switch follows this solver for the duration of its call (ops.set_deterministic)."""
        if isinstance(data, torch.Tensor) and data.is_cuda:
            want = self.deterministic
            if want is None:
                want = torch.are_deterministic_algorithms_enabled()
            want = bool(want)
            if ops.is_deterministic() != want:
                ops.set_deterministic(want)
            self._deterministic_now = want      # (a plain attribute: part of the hipGraph key, a capture bakes the choice in)

    # ------------------------------------------------------------------------------- public API
    def adversarial_training(self, data, model, optimize_flags=None, init_output=None, lazy_load=False,
                             power_iteration=False, n_iter=1, step_sizes=None, anatomy_mask_images=None,
                             anatomy_reg_weight=50, volume_preserve_tolerance=5 * 1e-4):
        """Optimise the chain against ``model`` on ``data`` and return the adversarial consistency loss
        (adv_compose_solver.py:43-146)."""
        n_t = len(self.chain_of_transforms)
        if optimize_flags is not None:
            assert n_t == len(optimize_flags), \
                f'must specify each transform is learnable or not, expect {n_t} flags, but got {optimize_flags}'
        else:
            if n_iter == 0:
                optimize_flags = [False] * n_t
            elif n_iter > 0:
                optimize_flags = [True] * n_t
            else:
                raise NotImplementedError
        if isinstance(power_iteration, bool):
            power_iterations = [power_iteration] * n_t
        elif isinstance(power_iteration, list):
            assert n_t == len(power_iteration), 'must specify each transform optimization mode'
            power_iterations = power_iteration
        elif isinstance(power_iteration, str):
            if "smart" == power_iteration:
                power_iterations = [t.get_name() == 'noise' for t in self.chain_of_transforms]
            else:
                raise NotImplementedError(power_iteration)
        for i, pi in enumerate(power_iterations):
            self.chain_of_transforms[i].power_iteration = pi
        if step_sizes is None:
            step_sizes = [1] * n_t
        elif isinstance(step_sizes, (float, int)):
            step_sizes = [step_sizes] * n_t
        elif isinstance(step_sizes, list):
            assert len(step_sizes) == n_t, 'specify step size for each transformation'
        else:
            raise ValueError('please use scalar or a  list of scalar to set step size')
        self._resolve_global_batch(data.size(0), data.device)
        self._apply_deterministic(data)
        pending = None
        if self.hip_graph and n_iter >= 1 and self._graphable(data, model, init_output, anatomy_mask_images):
            # the ascent loop as one hipGraph replay (or one of the ordinary calls that record its launch plan)
            init_output, pending = self._graphed_ascent(data, model, init_output, lazy_load, n_iter, optimize_flags, step_sizes,
                                                        anatomy_mask_images, anatomy_reg_weight, volume_preserve_tolerance)
        else:
            if self.hip_graph and n_iter >= 1:
                self.graph_stats["refused"] += 1
            if init_output is None:
                init_output = self.get_init_output(data=data, model=model)
            self.init_random_transformation(lazy_load, anatomy_mask_images=anatomy_mask_images,
                                            volume_preserve_tolerance=volume_preserve_tolerance)
            if n_iter >= 1:
                self.chain_of_transforms = self.optimizing_transform(
                    data=data, model=model, init_output=init_output, n_iter=n_iter, optimize_flags=optimize_flags,
                    step_sizes=step_sizes, anatomy_mask_images=anatomy_mask_images,
                    anatomy_reg_weight=anatomy_reg_weight, volume_preserve_tolerance=volume_preserve_tolerance)
        if pending is not None and self._final_pass_has_side_effects(model):
            # a model in train() mode leaves something behind in its forward (BatchNorm running statistics, the dropout RNG):
            # the final pass must not run on transforms a violated replay would then discard -- the verdict of the replay
            # first (one event wait: the host's overlap with the graph is given up for such models), the pass once
            if not pending():
                init_output = self._redo_ascent()
            pending = None
        dist, adv_data, adv_output, warped_back_adv_output = self.calc_adv_consistency_loss(
            data.detach(), model, init_output=init_output, chain_of_transforms=self.chain_of_transforms)
        if pending is not None and not pending():
            # the replay measured a displacement outside the interval its frozen kernel selection is exact for: this call
            # again the ordinary way, from the same initial parameters (its measurements widen the plan)
            init_output = self._redo_ascent()
            dist, adv_data, adv_output, warped_back_adv_output = self.calc_adv_consistency_loss(
                data.detach(), model, init_output=init_output, chain_of_transforms=self.chain_of_transforms)
        self._redo_ascent = None        # (a closure over this call's tensors and over self: no cycle is left behind)
        self.init_output = init_output
        self.warped_back_adv_output = warped_back_adv_output
        self.origin_data = data
        self.adv_data = adv_data
        self.adv_predict = adv_output
        if self.debug:
            print('[outer loop] loss', dist.item())
        return dist

    # ------------------------------------------------------------------------------- hipGraph replay of the ascent loop
    @staticmethod
    def _final_pass_has_side_effects(model):
        """Does a forward of `model` change anything but its output?  Anything in train() mode may (BatchNorm statistics,
        num_batches_tracked, dropout RNG state); an eval()-mode module is taken to be a pure function of its input."""
        return isinstance(model, torch.nn.Module) and any(m.training for m in model.modules())

    def _graphable(self, data, model, init_output, anatomy_mask_images):
        """What a capture cannot hold: host-side decisions inside the loop (debug prints, the host NaN check, third-party
        transforms), collectives (not captured: sharded runs stay on the ordinary path) and CPU data.  The anatomy ladder
        (adv_compose_solver.py:369-403) is a host decision BEHIND the n_iter steps: the graph holds the steps and the score
        of the first volume check, the ladder is walked from that score the ordinary way (round 6); its score is a
        whole-batch mean, so a sharded run with an anatomy mask stays on the ordinary path."""
        chain = self.chain_of_transforms
        anat = anatomy_mask_images
        return (isinstance(data, torch.Tensor) and data.is_cuda and data.dtype == torch.float32
                and (anat is None or (isinstance(anat, torch.Tensor) and anat.is_cuda and anat.dtype == torch.float32
                                      and anat.size() == data.size() and self.process_group is None))
                and not self.debug and self._shardable_capture(data)
                and self.device_nan_guard and not getattr(self, 'full_backward', False)
                and ops.ADAPTIVE_HALO and len(chain) > 0 and all(type(t) in _NATIVE for t in chain)
                and not any(getattr(t, 'debug', False) for t in chain)
                and isinstance(model, torch.nn.Module)
                and (init_output is None or (isinstance(init_output, torch.Tensor) and init_output.is_cuda)))

    def _shardable_capture(self, data):
        """Sharded runs (round 5): collectives are never captured.  A 2D loop holds exactly one per step -- the all-reduce
        that turns the local loss into the whole-batch value gating the updates (optimizing_transform) -- and the gate only
        asks whether that value is finite: inside a capture the LOCAL value gates, the per-step values leave the graph in
        one vector, and ONE all-reduce per replayed call (with the plan's violation flag in the same vector, so that every
        rank takes the same decision) tells whether the premise "finite everywhere, at every step" held; if not, every rank
        runs the call again the ordinary way.  What else would need a collective inside the loop stays on the ordinary
        path: the 3D step rule (a norm per DemonsCompose), `if_norm_image` without a given range."""
        if self.process_group is None:
            return True
        return data.dim() == 4 and not (self.if_norm_image and (self.min_intensity is None or self.max_intensity is None))

    _PLAIN_TYPES = (int, float, str, bool)

    @classmethod
    def _plain(cls, v):
        """`v` as a hashable plain value (numbers, strings, flat lists / tuples of them), else None."""
        if type(v) in cls._PLAIN_TYPES or isinstance(v, cls._PLAIN_TYPES):
            return v
        if type(v) in (list, tuple) and all(type(x) in cls._PLAIN_TYPES or x is None or isinstance(x, cls._PLAIN_TYPES) for x in v):
            return tuple(v)
        return None

    @classmethod
    def _plain_attrs(cls, obj, skip=()):
        """The plain attributes of `obj` as a tuple of (name, value) in the object's own attribute order (one pass, no sort:
        the order of an object's __dict__ only changes when attributes are added, which changes the tuple anyway)."""
        out = []
        plain = cls._PLAIN_TYPES
        for k, v in obj.__dict__.items():
            if type(v) in plain:
                out.append((k, v))
            elif v is None or k in skip:
                continue
            else:
                p = cls._plain(v)
                if p is not None:
                    out.append((k, p))
        return tuple(out)

    def _graph_key(self, data, model, init_output, lazy_load, n_iter, optimize_flags, step_sizes, anatomy=None):
        """Everything the captured launch sequence depends on besides tensor CONTENTS: shapes, the model's storage, the
        arguments of the call, every plain attribute of the solver and of its transforms."""
        # (the OBJECTS as well: a replay hands its results to the transforms it captured -- a chain rebuilt from the same
        # configuration is a new signature)
        tr = tuple((type(t).__name__, id(t)) + self._plain_attrs(t) for t in self.chain_of_transforms)
        mod = (id(model), model.training) + tuple(p.data_ptr() for p in model.parameters()) \
            + tuple(b.data_ptr() for b in model.buffers())
        mine = self._plain_attrs(self, skip=('graph_stats',))
        return (tuple(data.shape), str(data.device), None if init_output is None else tuple(init_output.shape), bool(lazy_load),
                int(n_iter), tuple(bool(f) for f in optimize_flags), tuple(step_sizes), tr, mod, mine, anatomy)

    def _graphed_ascent(self, data, model, init_output, lazy_load, n_iter, optimize_flags, step_sizes,
                        anatomy_mask_images=None, anatomy_reg_weight=50, volume_preserve_tolerance=5 * 1e-4):
        """get_init_output + init_random_transformation + optimizing_transform with the launches of the prediction and of the
        ascent loop replayed from a hipGraph.

        The first `hip_graph_record_calls` calls with a new signature run the ordinary way while an ops.LaunchPlan records
        what their kernel selection read back from the device (displacement bounds, the 3D step count); the next call
        captures the loop once with the selection frozen from that record; every later call draws its initial parameters
        the ordinary way, copies them and the data into the captured buffers and replays.  The replay checks ON THE DEVICE
        that what it measures stays inside the intervals the frozen selection is exact for (ops.LaunchPlan.check); the
        returned closure waits for the graph (not for the final pass queued behind it) and tells whether the check held.
        Returns (init_output, closure or None).

        With an anatomy mask the graph holds the n_iter steps (each with its constant regulariser term) and the score of the
        first volume check; the host waits for the replay, reads flag and score together and walks the ladder
        (optimizing_transform's `_ladder`) from there -- nothing to do when the check passes, further steps the ordinary way
        when it does not.  The recorded calls are divided at the same place, so the plan holds the same sites."""
        anat = anatomy_mask_images
        anat_kw = dict(anatomy_mask_images=anat, anatomy_reg_weight=anatomy_reg_weight,
                       volume_preserve_tolerance=volume_preserve_tolerance)
        self.init_random_transformation(lazy_load, anatomy_mask_images=anat, volume_preserve_tolerance=volume_preserve_tolerance)
        key = self._graph_key(data, model, init_output, lazy_load, n_iter, optimize_flags, step_sizes,
                              None if anat is None else (tuple(anat.shape), float(anatomy_reg_weight),
                                                         float(volume_preserve_tolerance)))
        rec = self._graphs.get(key)
        if rec is None:
            if len(self._graphs) >= 8:       # (each holds a memory pool of one whole call)
                self._graphs.clear()
            rec = self._graphs[key] = {"plan": ops.LaunchPlan(float(self.hip_graph_margin)), "state": "record", "graph": None}
        chain = list(self.chain_of_transforms)
        init_params = [t.param.detach() for t in chain]
        given = init_output

        def ordinary(record):
            for t, p in zip(chain, init_params):
                t.param = p
            self.chain_of_transforms = chain
            if record:
                rec["plan"].begin_record()
            ops._PLAN = rec["plan"] if record else None
            lad = {"defer": True} if anat is not None else None
            try:
                io = given if given is not None else self.get_init_output(data=data, model=model)
                self.chain_of_transforms = self.optimizing_transform(
                    data=data, model=model, init_output=io, n_iter=n_iter, optimize_flags=optimize_flags,
                    step_sizes=step_sizes, _ladder=lad, **anat_kw)
            finally:
                ops._PLAN = None
            if lad is not None and "score" in lad:      # the ladder: never recorded, never captured
                self.chain_of_transforms = self.optimizing_transform(
                    data=data, model=model, init_output=io, n_iter=n_iter, optimize_flags=optimize_flags,
                    step_sizes=step_sizes, _ladder={"resume": lad["score"]}, **anat_kw)
            if record:
                rec["plan"].end_record()
                self.graph_stats["recorded"] += 1
                if rec["state"] == "record" and rec["plan"].calls >= max(1, int(self.hip_graph_record_calls)):
                    rec["state"] = "capture"
            return io

        self._redo_ascent = lambda: ordinary(True)
        if self.process_group is not None:
            # the record / capture / replay state is kept per rank (keyed by its local shapes and pointers): a rank that meets
            # a new key (an uneven last shard, a rebuilt chain) while the others replay would issue n_iter per-step all-reduces
            # against their one per call -- a collective mismatch.  One all-reduce per call tells every rank
            # whether all of them are in the same state; if not, all of them run this call the ordinary way (their per-step
            # collectives pair up), each recording where its own state still records
            order = {"record": 0.0, "off": 1.0, "capture": 2.0, "replay": 3.0}
            mine = order[rec["state"]]
            both = self._all_reduce_(torch.tensor([mine, -mine], device=data.device), "min").tolist()
            if both[0] != -both[1]:
                self.graph_stats["deferred"] = self.graph_stats.get("deferred", 0) + 1
                return ordinary(rec["state"] == "record"), None
        if rec["state"] == "record":
            return ordinary(True), None
        if rec["state"] == "off":
            return ordinary(False), None
        if rec["state"] == "capture":
            captured = True
            try:
                self._capture_ascent(rec, chain, init_params, data, model, given, n_iter, optimize_flags, step_sizes, anat_kw)
            except Exception as exc:         # not capturable after all (the model, most likely): the ordinary path from now on
                logging.warning('advchain_amd: hipGraph capture of the ascent loop failed (%s: %s); running it the ordinary way',
                                type(exc).__name__, exc)
                captured = False
            if self.process_group is not None:      # every rank replays or none does: their collectives have to pair up
                agreed = self._all_reduce_(torch.tensor([1.0 if captured else 0.0], device=data.device), "min")
                captured = bool(float(agreed) > 0)
            if not captured:
                rec["state"], rec["graph"] = "off", None
                return ordinary(False), None
            rec["state"] = "replay"
        # replay
        if data.data_ptr() != rec["data"].data_ptr():
            rec["data"].copy_(data)
        if given is not None and given.data_ptr() != rec["init_output"].data_ptr():
            rec["init_output"].copy_(given)
        if anat is not None and anat.data_ptr() != rec["anat"].data_ptr():
            rec["anat"].copy_(anat)
        torch._foreach_copy_(rec["params"], init_params)      # (one launch for the chain's parameters, not one each)
        model.zero_grad()
        rec["graph"].replay()
        sharded = self.process_group is not None
        if sharded:      # [violation flag, non-finite local losses, the per-step losses], summed over the ranks: one collective per call
            check = self._all_reduce_(rec["check"].clone())
            rec["check_host"].copy_(check, non_blocking=True)
        else:
            rec["flag_host"].copy_(rec["plan"].flag, non_blocking=True)
            if rec.get("score") is not None:
                rec["score_host"].copy_(rec["score"].reshape(1), non_blocking=True)
        rec["event"].record(ops._stream_obj())
        self.graph_stats["replays"] += 1
        outs = [torch.empty_like(op) for op in rec["out_params"]]
        torch._foreach_copy_(outs, rec["out_params"])
        for t, state, o in zip(rec["transforms"], rec["attrs"], outs):
            t.__dict__.update(state)
            t.param = o
        self.chain_of_transforms = list(rec["transforms"])
        self.last_inner_dist = check[-1].reshape(rec["out_last_inner"].shape) if sharded else rec["out_last_inner"].clone()
        io = given if given is not None else rec["out_init_output"].clone()

        def held():
            rec["event"].synchronize()
            recent = rec.setdefault("recent", [])
            if sharded:      # the same numbers on every rank: the same decision on every rank
                h = rec["check_host"].tolist()
                bad = h[0] != 0 or h[1] != 0 or not all(math.isfinite(v) for v in h)
            else:
                bad = int(rec["flag_host"][0]) != 0
            recent.append(bad)
            del recent[:-32]
            if not bad:
                return True
            # this call runs again the ordinary way (recorded: its measurements widen the plan's record).  The graph stays: it
            # is exact for every call inside its intervals.  Only when violations are frequent -- more than one replay in
            # eight of the last 32 -- is the loop captured again from the widened record (a capture costs ~6 calls)
            self.graph_stats["violations"] += 1
            rec["violations"] = rec.get("violations", 0) + 1
            if len(recent) >= 8 and sum(recent) * 8 > len(recent):
                rec["recaptures"] = rec.get("recaptures", 0) + 1
                rec["recent"] = []
                if rec["recaptures"] > 4:
                    rec["state"], rec["graph"] = "off", None
                else:
                    rec["plan"].margin = min(2.0, rec["plan"].margin * 1.15)
                    rec["plan"].thaw()
                    rec["state"], rec["graph"] = "capture", None
            return False
        if rec.get("score") is not None:
            # the ladder is a host decision: flag and score now (the host's run-ahead over the final pass is given up)
            if not held():
                return ordinary(True), None
            self.chain_of_transforms = self.optimizing_transform(
                data=data, model=model, init_output=io, n_iter=n_iter, optimize_flags=optimize_flags,
                step_sizes=step_sizes, _ladder={"resume": rec["score_host"][0].clone()}, **anat_kw)
            return io, None
        return io, held

    def _capture_ascent(self, rec, chain, init_params, data, model, given, n_iter, optimize_flags, step_sizes, anat_kw):
        plan = rec["plan"]
        anat = anat_kw["anatomy_mask_images"]
        rec["anat"] = None if anat is None else anat.detach().clone()
        lad = {"defer": True} if anat is not None else None
        anat_kw = dict(anat_kw, anatomy_mask_images=rec["anat"])
        rec["data"] = data.detach().clone()
        rec["init_output"] = None if given is None else given.detach().clone()
        rec["params"] = [p.clone() for p in init_params]
        rec["keep"] = [getattr(t, "_tables", None) for t in chain] + [model]      # what the graph's pointers refer to
        rec["flag_host"] = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        rec["event"] = torch.cuda.Event()
        for t, p in zip(chain, rec["params"]):
            t.param = p
        self.chain_of_transforms = chain
        plan.freeze(data.device)
        graph = torch.cuda.CUDAGraph()
        ops._PLAN = plan
        # the cyclic collector stays out of the capture: it may free another CUDAGraph (an earlier solver's, kept alive by a
        # reference cycle until now), and destroying a graph or returning its pool while a capture is open aborts the process
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        sharded = self.process_group is not None
        self._local_steps = [] if sharded else None
        try:
            with torch.cuda.graph(graph):
                plan.flag.zero_()
                plan.rewind()
                io = rec["init_output"] if given is not None else self.get_init_output(data=rec["data"], model=model)
                transforms = self.optimizing_transform(data=rec["data"], model=model, init_output=io, n_iter=n_iter,
                                                       optimize_flags=optimize_flags, step_sizes=step_sizes, _ladder=lad,
                                                       **anat_kw)
                plan.finish()
                if sharded:
                    vals = torch.cat(self._local_steps) if self._local_steps else torch.zeros(1, device=data.device)
                    rec["check"] = torch.cat([plan.flag.reshape(1).float(), (~torch.isfinite(vals)).float().sum().reshape(1), vals.float()])
        finally:
            ops._PLAN = None
            self._local_steps = None
            if gc_was_on:
                gc.enable()
        if plan.cursor != len(plan.frozen):
            raise ops.PlanMismatch("launch plan: the capture visited %d of %d recorded sites" % (plan.cursor, len(plan.frozen)))
        rec["graph"] = graph
        rec["transforms"] = list(transforms)
        rec["out_params"] = [t.param.detach() for t in transforms]
        # the plain attributes the loop leaves on the transforms (is_training, power_iteration, ...): restored after a replay
        rec["attrs"] = [{k: v for k, v in vars(t).items() if self._plain(v) is not None} for t in transforms]
        rec["out_init_output"] = io
        rec["out_last_inner"] = self.last_inner_dist
        rec["score"] = lad.get("score") if lad is not None else None
        if rec["score"] is not None:
            rec["score_host"] = torch.zeros(1, dtype=torch.float32, pin_memory=True)
        if sharded:
            rec["check_host"] = torch.zeros(rec["check"].numel(), dtype=torch.float32, pin_memory=True)
        self.graph_stats["captures"] += 1


    @property
    def diffs(self):
        return [t.diff for t in self._last_chain]

    def forward(self, data, chain_of_transforms=None, interp=None, padding_mode=None, _ride=False):
        """Apply the chain in order (adv_compose_solver.py:148-176)."""
        data.requires_grad = False
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        # the reference clones here (adv_compose_solver.py:160); the native transforms never write to their input, so the
        # copy is only made for a third-party transform that might
        native = len(chain_of_transforms) > 0 and all(isinstance(t, _NATIVE) for t in chain_of_transforms)
        t_data = data.detach() if native else data.detach().clone()
        self._last_chain = list(chain_of_transforms)
        self._ride = None
        if _ride:       # the validity mask rides along: every geometric transform warps (data, mask) in one call
            ride = None
            for transform in chain_of_transforms:
                if transform.is_geometric():
                    if ride is None:
                        ride = ops.cached_ones((t_data.shape[0], 1) + tuple(t_data.shape[2:]), t_data.device)
                    t_data, ride = transform.forward(t_data, _ride=ride)
                else:
                    t_data = transform.forward(t_data)
            self._ride = ride
        else:
            for transform in chain_of_transforms:
                t_data = transform.forward(t_data, interp=interp, padding_mode=padding_mode)
        if self.if_norm_image:
            lo = self.min_intensity
            hi = self.max_intensity
            if lo is None:
                lo = self._all_reduce_(torch.min(data).reshape(1), "min")[0]
            if hi is None:
                hi = self._all_reduce_(torch.max(data).reshape(1), "max")[0]
            t_data = torch.clamp(t_data, lo, hi)
        return t_data

    def predict_forward(self, data, chain_of_transforms=None, interp=None, padding_mode=None):
        # adv_compose_solver.py:184-197
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        self._last_chain = list(chain_of_transforms)
        for transform in chain_of_transforms:
            data = transform.predict_forward(data, interp=interp, padding_mode=padding_mode)
        return data

    def backward(self, data, chain_of_transforms=None, interp=None, padding_mode=None):
        # adv_compose_solver.py:199-208
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        for transform in reversed(chain_of_transforms):
            data = transform.backward(data, interp=interp, padding_mode=padding_mode)
        return data

    def predict_backward(self, data, chain_of_transforms=None, interp=None, padding_mode=None):
        # adv_compose_solver.py:210-219
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        for transform in reversed(chain_of_transforms):
            data = transform.predict_backward(data, interp=interp, padding_mode=padding_mode)
        return data

    def _ride_ok(self, chain):
        """The validity mask (ones -> forward chain -> backward chain -> != 0, adv_compose_solver.py:262-268) can ride
        through the warps of the data and of the prediction instead of being warped by four launches of its own: the
        built-in transforms only (exact types: a subclass may override forward), default interpolation / padding, and loss
        terms whose gradient w.r.t. the mask is not needed."""
        return (ops.RIDE_MASK and len(chain) > 0 and all(type(t) in _NATIVE for t in chain)
                and all(t in ('mse', 'contour') for t in self.divergence_types)
                and all(t._ride_ok() for t in chain if t.is_geometric()))

    def _predict_backward_with_mask(self, data, chain, init_output):
        """predict_backward(data) and the validity mask in the same launches (self._ride: the mask after forward())."""
        ride, self._ride = self._ride, None
        geo = [t for t in chain if t.is_geometric()]
        for transform in reversed(chain):
            if transform.is_geometric():        # (the mask is a second, non-differentiable output of the warp)
                data, ride = transform.predict_backward(data, _ride=ride, _ride_nonzero=transform is geo[0])
            else:
                data = transform.predict_backward(data)
        return data, ride.expand(init_output.shape)

    def loss_fn(self, pred, reference, mask=None):
        """Inconsistency of two predictions in the same coordinates (adv_compose_solver.py:221-234).  Under
        batch sharding the returned tensor is this rank's partial (global normalisers): partials sum to the
        whole-batch loss."""
        if self.process_group is not None and self._global_batch is None:
            self._resolve_global_batch(pred.size(0), pred.device)
        # (the loss is differentiated w.r.t. `pred` only: a reference that carries a graph -- a caller's init_output -- is a
        # constant here, as it is for every result of the reference's solver: adv_compose_solver.py:268-270,310,366)
        reference = reference.detach() if isinstance(reference, torch.Tensor) else reference
        return calc_segmentation_consistency(output=pred, reference=reference, divergence_types=self.divergence_types,
                                             divergence_weights=self.divergence_weights, scales=[0], mask=mask,
                                             class_weights=self.class_weights, is_gt=self.is_gt,
                                             global_batch=self._global_batch)

    # ------------------------------------------------------------------------------- masks
    def _shared_fields(self, chain, on):
        for t in chain:
            hook = getattr(t, '_begin_shared_fields' if on else '_end_shared_fields', None)
            if hook is not None:
                hook()

    def _validity_mask(self, init_output, chain):
        """ones -> forward chain -> backward chain -> (x != 0)  (adv_compose_solver.py:262-268,321-325, Q12)."""
        native = all(isinstance(t, _NATIVE) for t in chain)
        zero_grad_ok = all(t in ('mse', 'contour') for t in self.divergence_types)
        if native and zero_grad_ok:
            with torch.no_grad():
                ones = ops.cached_ones((init_output.shape[0], 1) + tuple(init_output.shape[2:]), init_output.device)
                fb = self.predict_backward(self.predict_forward(ones, chain), chain)
                m = ops.nonzero_mask(fb) if fb is not ones else ones       # one launch for `!= 0` and the cast
            return m.expand(init_output.shape)
        masks = torch.ones_like(init_output, dtype=init_output.dtype, device=init_output.device, requires_grad=False)
        fb = self.predict_backward(self.predict_forward(masks, chain), chain)
        fb[fb != 0] = 1
        return fb

    def calc_adv_consistency_loss(self, data, model, init_output, chain_of_transforms=None):
        """Consistency loss under the current (fixed) transforms (adv_compose_solver.py:236-279)."""
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        for tr in chain_of_transforms:
            tr.eval()
        ops.HINT_SLOT = 0           # (kernel-selection hints are kept per position in the call: 0 = the final pass)
        self._shared_fields(chain_of_transforms, True)
        try:
            geo = self.if_contains_geo_transform(chain_of_transforms)
            ride = geo and self._ride_ok(chain_of_transforms)
            adv_data = self.forward(data, chain_of_transforms, _ride=ride)
            old_state = model.training
            model.train()
            with _fix_dropout(model):
                adv_output = self.get_net_output(model, adv_data.detach().clone())
            if geo:
                if ride:
                    warped_back_adv_output, mask = self._predict_backward_with_mask(adv_output, chain_of_transforms, init_output)
                else:
                    mask = self._validity_mask(init_output, chain_of_transforms)
                    warped_back_adv_output = self.predict_backward(adv_output, chain_of_transforms)
                dist = self.loss_fn(pred=warped_back_adv_output, reference=init_output.detach(), mask=mask)
            else:
                warped_back_adv_output = adv_output
                dist = self.loss_fn(pred=adv_output, reference=init_output.detach())
            model.train(old_state)
        finally:
            self._shared_fields(chain_of_transforms, False)
        return self._global_value(dist), adv_data, adv_output, warped_back_adv_output

    def compute_anatomy_misoverlapping_loss(self, anatomy_mask_images):
        """Round-trip warp of the anatomy mask, thresholded, MSE to the original (adv_compose_solver.py:281-287).
        Value only: its gradient w.r.t. the transform parameters is exactly zero (Q18)."""
        with torch.no_grad():
            rec = self.predict_backward(self.predict_forward(anatomy_mask_images))
            rec = (rec >= 0.5).to(anatomy_mask_images.dtype)
            if self.process_group is None:
                score = torch.nn.functional.mse_loss(rec, anatomy_mask_images)
            else:
                se = self._all_reduce_(((rec - anatomy_mask_images) ** 2).sum().reshape(1))[0]
                per_sample = anatomy_mask_images.numel() // anatomy_mask_images.shape[0]
                score = se / (float(self._global_batch) * per_sample)
        if self.debug:
            print('anatomy preserving error:', score)
        return score

    def _ascent_step(self, model, data, init_output, optimize_flags, step_sizes, i_iter, use_anatomy,
                     anatomy_mask_images, anatomy_reg_weight):
        """One step of the ascent loop (adv_compose_solver.py:300-366): forward through the chain and the model, the
        consistency loss, backward to the parameters, the normalised-gradient updates."""
        model.zero_grad()
        ops.HINT_SLOT = i_iter      # (ascent step i of this call resembles ascent step i of the previous call)
        self.make_learnable_transformation(optimize_flags=optimize_flags,
                                           chain_of_transforms=self.chain_of_transforms)
        self._shared_fields(self.chain_of_transforms, True)
        try:
            geo = self.if_contains_geo_transform(self.chain_of_transforms)
            ride = geo and self._ride_ok(self.chain_of_transforms)
            augmented_data = self.forward(data.detach(), _ride=ride)     # (a new tensor object: forward() clears ITS requires_grad flag)
            with _disable_tracking_bn_stats(model):
                perturbed_output = self.get_net_output(model, augmented_data)
            if geo:
                if ride:
                    warped_back_prediction, mask = self._predict_backward_with_mask(perturbed_output, self.chain_of_transforms,
                                                                                    init_output)
                else:
                    warped_back_prediction = self.predict_backward(perturbed_output)
                    mask = self._validity_mask(init_output, self.chain_of_transforms)
                dist = self.loss_fn(pred=warped_back_prediction, reference=init_output, mask=mask)
                if use_anatomy:
                    assert anatomy_mask_images.size() == data.size(), \
                        "gt mask should be of the same size as input image "
                    reg_loss = anatomy_reg_weight * self.compute_anatomy_misoverlapping_loss(
                        anatomy_mask_images=anatomy_mask_images)
                    if self.debug:
                        print("consistency loss", dist.item())
                        print("reg_loss:", reg_loss.item())
                    # constant w.r.t. every parameter (Q18); a rank-local share keeps the partials summable
                    dist = dist + reg_loss / (1 if self.process_group is None else self._dist().get_world_size(self.process_group))
            else:
                dist = self.loss_fn(pred=perturbed_output, reference=init_output.detach())
            value = self._global_value(dist)
            if self.debug:
                print('[inner loop], step {}: dist {}'.format(str(i_iter), value.item()))
            self.last_inner_dist = value.detach()
            # NaN / inf guard (adv_compose_solver.py:343-347: a non-finite loss skips backward and updates).  For an
            # all-native chain it is evaluated ON THE DEVICE: backward and updates are enqueued unconditionally and
            # every update kernel keeps the old parameters when the loss is not finite -- the host never reads the
            # loss back in the middle of a step, so it runs ahead of the GPU for the whole call (the read-back
            # stalled the GPU once per step: 9.8 ms per cfg-2 call for 8.8 ms of kernels).  Third-party transforms
            # get the literal check.
            flagged = [t for flag, t in zip(optimize_flags, self.chain_of_transforms) if flag]
            device_guard = (self.device_nan_guard and value.is_cuda and all(_native_update(t) for t in flagged)
                            and not getattr(self, 'full_backward', False))
            if not device_guard and not math.isfinite(float(value.detach())):     # one read-back, no launches
                dist = 0
            else:
                for t in flagged:
                    t._gate = value.detach() if device_guard else None
                self._backward_to_transforms(dist, optimize_flags)
                i_tr = 0  # never advanced in the reference (adv_compose_solver.py:349-364): every transform
                #           is stepped with step_sizes[0]; kept for result parity
                if device_guard and self._fused_update(flagged, step_sizes):
                    flagged = []          # (one launch stepped every transform: nothing left for the loop below)
                for flag, transform in zip(optimize_flags, self.chain_of_transforms):
                    if flag and flagged:
                        if self.debug:
                            print('update {} parameters'.format(transform.get_name()))
                        try:
                            step_size = step_sizes[i_tr]
                        except Exception:
                            step_size = transform.get_step_size()
                            logging.warning(f'use default step size:{step_size}')
                        transform.optimize_parameters(step_size=step_size)
        finally:
            # (also when the backward or an update raised: a stale loss must not gate a later manual update)
            for transform in self.chain_of_transforms:
                if getattr(transform, '_gate', None) is not None:
                    transform._gate = None
            self._shared_fields(self.chain_of_transforms, False)
        model.zero_grad()

    def optimizing_transform(self, model, data, init_output, optimize_flags, n_iter=1, step_sizes=None,
                             anatomy_mask_images=None, anatomy_reg_weight=50, volume_preserve_tolerance=5 * 1e-4,
                             _ladder=None):
        """The N-step ascent loop with the volume-preservation ladder behind it (adv_compose_solver.py:289-405).

        `_ladder` (private; the hipGraph path) splits the loop where the host first has to look at a device value:
        {"defer": True} stops BEFORE the first volume check, leaving its score (a device scalar, no read-back) in
        `_ladder["score"]`; {"resume": score} enters the loop AT that check with the score handed in and walks the ladder
        from there the ordinary way.  defer + resume == the undivided loop, launch for launch."""
        stop_flag = False if n_iter > 0 else True
        i_iter = 0
        one_time_iter = n_iter
        transforms = []
        use_anatomy = anatomy_mask_images is not None and abs(anatomy_reg_weight) > 1e-32
        ladder = _ladder if _ladder is not None else {}
        resumed = "resume" in ladder
        if resumed:
            # the first n_iter steps and the rescaling behind them were run (replayed) by the caller, who hands in the score of
            # the first volume check: the loop is entered at that check
            i_iter = n_iter
        while stop_flag is False:
            if not resumed:
                i_iter += 1
                self._ascent_step(model, data, init_output, optimize_flags, step_sizes, i_iter, use_anatomy,
                                  anatomy_mask_images, anatomy_reg_weight)

            if i_iter == n_iter:
                if resumed:
                    transforms = list(self.chain_of_transforms)      # (rescaled and in eval() since the deferred part)
                else:
                    transforms = []
                    for flag, transform in zip(optimize_flags, self.chain_of_transforms):
                        if flag:
                            transform.rescale_parameters()
                            transform.eval()
                        transforms.append(transform)
                if self.if_contains_geo_transform(transforms) and use_anatomy:
                    if resumed:
                        score, resumed = ladder["resume"], False
                    else:
                        score = self.compute_anatomy_misoverlapping_loss(anatomy_mask_images)
                        if ladder.get("defer"):
                            ladder["score"] = score
                            break
                    print('activating volume preserving check')
                    if abs(score) <= volume_preserve_tolerance:
                        print('Success! pass the volume preserving check')
                        stop_flag = True
                    else:
                        if i_iter >= 3 * one_time_iter:
                            stop_flag = True
                            self.init_random_transformation(anatomy_mask_images=anatomy_mask_images,
                                                            volume_preserve_tolerance=volume_preserve_tolerance)
                        else:
                            if i_iter == 2 * one_time_iter:
                                self.init_random_transformation(anatomy_mask_images=anatomy_mask_images,
                                                                volume_preserve_tolerance=volume_preserve_tolerance)
                                n_iter += one_time_iter
                                print('warning: the volume is not preserved, will continue search with a new initialization')
                            else:
                                n_iter += 1
                                print('warning: the volume is not preserved, will continue search with one more step')
                        for flag, transform in zip(optimize_flags, self.chain_of_transforms):
                            if flag:
                                transform.train()
                        transforms.append(transform)  # reference quirk (line 399): last transform listed twice
                else:
                    stop_flag = True
        ops.HINT_SLOT = 0       # (a later user-level forward() keys its kernel-selection hints to "no ascent step")
        return transforms

    def _fused_update(self, flagged, step_sizes):
        """The updates of one ascent step as ONE launch (ops.update_multi) when every transform is a built-in one stepping with
        its own built-in method: the per-transform optimize_parameters() calls of adv_compose_solver.py:349-364 with the same
        formulas (adv_noise.py:51-64, adv_bias.py:139-148, adv_morph.py:501-516, adv_affine.py:182-198), the device-side NaN
        gate included.  Anything else -- a subclass, a method replaced on the instance, a missing gradient (the reference
        logs a warning there), more than 8 transforms -- returns False and the loop steps them one by one."""
        if not ops.FUSED_UPDATE or not flagged or len(flagged) > 8 or self.debug:
            return False
        try:
            step = float(step_sizes[0])
        except Exception:
            return False
        items, fused, alone = [], [], []
        for t in flagged:
            if type(t) not in _NATIVE or 'optimize_parameters' in vars(t):
                return False
            p = t.param
            g = p.grad if isinstance(p, torch.Tensor) else None
            if not (isinstance(g, torch.Tensor) and g.is_cuda and g.dtype == torch.float32 and g.shape == p.shape):
                return False
            # one workgroup per (transform, sample) row: a 3D noise row (a million values) belongs to the two-launch form,
            # whose workgroups share a row
            if not isinstance(t, AdvAffine) and g.numel() // max(1, g.shape[0]) > self._FUSED_UPDATE_ROW_MAX:
                alone.append(t)
                continue
            fused.append(t)
            items.append((None if t.power_iteration else p, g, 1.0 if t.power_iteration else step,
                          1 if isinstance(t, AdvAffine) else 0, p))
        if not items:
            return False
        outs = ops.update_multi(items, gate=flagged[0]._gate)
        for t, o in zip(fused, outs):
            t.param = o
            if isinstance(t, AdvMorph):
                t._field_cache = {}
        for t in alone:
            t.optimize_parameters(step_size=step)
        return True

    _FUSED_UPDATE_ROW_MAX = 1 << 18

    def _backward_to_transforms(self, dist, optimize_flags):
        """``dist.backward()`` of adv_compose_solver.py:348, restricted to the transform parameters: the reference
        also accumulates (and immediately zeroes, lines 310/366) the gradients of the model's weights -- for a conv
        net that is a weight-gradient convolution per ascent step that nobody reads.  Set ``full_backward = True``
        to get the literal behaviour."""
        flagged = [t for flag, t in zip(optimize_flags, self.chain_of_transforms) if flag]
        restricted = (not getattr(self, 'full_backward', False)
                      and all(isinstance(t, _NATIVE) and isinstance(t.param, torch.Tensor) and t.param.requires_grad
                              for t in flagged))
        if not flagged or not restricted:
            # a third-party transform may keep several leaves (or a list / dict of them): only the literal
            # backward() reaches every one of them
            dist.backward()
            return
        leaves = [t.param for t in flagged]
        # (the seed gradient is a cached scalar one: autograd's default builds a ones_like per call)
        seed = ops.cached_ones((), dist.device) if dist.is_cuda and dist.dim() == 0 and dist.dtype == torch.float32 else None
        grads = torch.autograd.grad(dist, leaves, grad_outputs=seed, allow_unused=True)
        for p, g in zip(leaves, grads):
            if g is None:
                continue
            p.grad = g if p.grad is None else p.grad + g      # accumulate, as backward() does

    def rescale_intensity(self, data, new_min=0, new_max=1, eps=1e-20):
        # adv_compose_solver.py:407-421
        old_size = data.size()
        flat = data.view(data.size(0), -1)
        old_max = torch.max(flat, dim=1, keepdim=True).values
        old_min = torch.min(flat, dim=1, keepdim=True).values
        out = (flat - old_min + eps) / (old_max - old_min + eps) * (new_max - new_min) + new_min
        return out.view(old_size)

    def get_net_output(self, model, data):
        """Override point (README: custom network outputs), adv_compose_solver.py:423-427."""
        return model.forward(data)

    def get_init_output(self, model, data):
        # adv_compose_solver.py:429-433
        with torch.no_grad():
            with _disable_tracking_bn_stats(model):
                reference_output = self.get_net_output(model, data)
        return reference_output

    def get_adv_data(self, data, model, init_output=None, n_iter=0, optimize_flags=None, step_sizes=None,
                     anatomy_mask_images=None, anatomy_reg_weight=50, volume_preserve_tolerance=5 * 1e-4):
        """Augmented input and the correspondingly warped reference (adv_compose_solver.py:435-463)."""
        self._resolve_global_batch(data.size(0), data.device)
        if init_output is None:
            init_output = self.get_init_output(model, data)
        if optimize_flags is None:
            optimize_flags = [True] * len(self.chain_of_transforms)
        if step_sizes is None:
            step_sizes = [1] * len(self.chain_of_transforms)
        self.init_random_transformation(lazy_load=False, anatomy_mask_images=anatomy_mask_images,
                                        volume_preserve_tolerance=volume_preserve_tolerance)
        origin_data = data.detach().clone()
        if n_iter > 0:
            optimized_transforms = self.optimizing_transform(
                data=data, model=model, init_output=init_output, n_iter=n_iter, optimize_flags=optimize_flags,
                step_sizes=step_sizes, anatomy_mask_images=anatomy_mask_images,
                anatomy_reg_weight=anatomy_reg_weight, volume_preserve_tolerance=volume_preserve_tolerance)
        else:
            optimized_transforms = self.chain_of_transforms
        augmented_data = self.forward(origin_data, optimized_transforms)
        augmented_label = self.predict_forward(init_output, optimized_transforms)
        return augmented_data, augmented_label

    def if_contains_geo_transform(self, chain_of_transforms=None):
        # adv_compose_solver.py:465-477
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        return sum(t.is_geometric() for t in chain_of_transforms) > 0

    def init_random_transformation(self, lazy_load=False, anatomy_mask_images=None,
                                   volume_preserve_tolerance=5 * 1e-4):
        """(Re-)draw transform parameters; geometric ones are re-drawn up to 11 times until the anatomy mask
        survives the round trip (adv_compose_solver.py:479-500)."""
        for transform in self.chain_of_transforms:
            if lazy_load:
                if transform.param is None:
                    transform.init_parameters()
            else:
                transform.init_parameters()
            if transform.is_geometric() == 1 and anatomy_mask_images is not None:
                i_iter = 0
                while self.compute_anatomy_misoverlapping_loss(anatomy_mask_images) > volume_preserve_tolerance:
                    transform.init_parameters()
                    i_iter += 1
                    if i_iter > 10:
                        break

    def reset_transformation(self, anatomy_mask_images=None, volume_preserve_tolerance=5 * 1e-4):
        self.init_random_transformation(lazy_load=False, anatomy_mask_images=anatomy_mask_images,
                                        volume_preserve_tolerance=volume_preserve_tolerance)

    def set_transformation(self, parameter_list):
        # adv_compose_solver.py:505-514
        for i, param in enumerate(parameter_list):
            self.chain_of_transforms[i].set_parameters(param)

    def train(self):
        if self.chain_of_transforms is not None:
            for transform in self.chain_of_transforms:
                transform.train()

    def eval(self):
        if self.chain_of_transforms is not None:
            for transform in self.chain_of_transforms:
                transform.eval()

    def make_learnable_transformation(self, optimize_flags, chain_of_transforms=None):
        # adv_compose_solver.py:525-538
        if chain_of_transforms is None:
            chain_of_transforms = self.chain_of_transforms
        for flag, transform in zip(optimize_flags, chain_of_transforms):
            if flag:
                transform.train()

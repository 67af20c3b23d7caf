"""Host side of the SE-SSD training step on the HIP library (BASELINE configs[2] / [3]).

What is here
  * FlatParams        all parameters (and all gradients) of a model in ONE 16-byte-aligned float32 buffer each; the
                      model's tensors become views, so autograd accumulates straight into the flat gradient buffer
  * one_cycle         the OneCycle schedule of det3d/solver/learning_schedules_fastai.py:70-95 (config.py:260)
  * FusedAdamEMA      clip_grad_norm_ + true-weight-decay Adam + EMA teacher as two launches over the flat buffers
                      (sessd_grad_clip_coef, sessd_adam_ema_step) instead of ~100 x 6 per-tensor host-issued kernels
                      (hooks/optimizer.py:50-53, fastai_optim.py:155-176, trainer_sessd.py:315-318)
  * allreduce_flat    ONE all-reduce of the flat gradient buffer (15.2 MB) over RCCL, replacing DDP's buckets plus the
                      duplicate DistOptimizerHook all-reduce (dist_utils.py:45-57); pre-divides like the reference
  * TrainStep         teacher forward (no grad, `*_raw` inputs) -> student forward -> loss -> backward -> all-reduce ->
                      fused update, in the order of trainer_sessd.py:250-275,340-357; `capture()` / `replay()`: the whole
                      iteration as ONE hipGraph (capacity-sized inputs with device-side counts, the OneCycle schedule and the
                      Adam constants computed on the device: sessd_one_cycle_args) -- nothing in it reads the host
  * capacity_example  pads a collated example to fixed row capacities and adds the device-side voxel counts that switch the
                      sparse module path to its capacity mode (spconv.SparseConvTensor(n_dev=...))
The sparse backbone (spconv.IndiceConvFunction) and the twelve dense convs of the neck (ops.Conv2dFunction) run forward
AND backward on the HIP kernels; BatchNorm, activations, the four 1x1 heads and the loss are torch ops."""
import copy
import math

import torch
import torch.distributed as dist

from . import ops
from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class FlatParams:
    """Flatten `model.parameters()` (in order, each tensor padded to a multiple of 4 floats) into `self.data` and their
    gradients into `self.grad`; every parameter's `.data` / `.grad` become views of those buffers."""

    def __init__(self, model, with_grad=True):
        self.params = [p for p in model.parameters()]
        if not self.params:
            raise ValueError("model has no parameters")
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatParams needs float32 parameters on one device")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev) if with_grad else None
        for p, o in zip(self.params, self.offsets):
            view = self.data[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            if with_grad:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        if self.grad is None:
            return
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach in case something set .grad to None / a new tensor
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)


    def release_grads(self):
        """Before backward: detach the parameters from the flat gradient buffer (`.grad = None`), so that autograd hands each
        parameter's gradient over as it is instead of ADDING it into a zeroed view -- ~100 small add launches and one fill per
        iteration (3 % of the kernel time of the round-2 trace). gather_grads() then packs them."""
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """After backward: every parameter's gradient into its slot of the flat buffer with one multi-tensor copy (a parameter
        that received none gets zeros), and `.grad` re-attached to the views (all-reduce and the fused update read the buffer)."""
        srcs, dsts = [], []
        for p, o in zip(self.params, self.offsets):
            view = self.grad[o:o + p.numel()].view_as(p)
            g = p.grad
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                srcs.append(g.detach().to(torch.float32).contiguous())
                dsts.append(view)
            p.grad = view
        if srcs:
            torch._foreach_copy_(dsts, srcs)


def one_cycle(step, total_steps, lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
    """(lr, momentum) of OneCycle at `step` (learning_schedules_fastai.py:70-95: cosine low->max over the first
    pct_start, then max->low/1e4; momentum mirrors it)."""
    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    a1 = int(total_steps * pct_start)
    low = lr_max / div_factor
    if step < a1:
        pct = step / float(a1)
        return cos(low, lr_max, pct), cos(moms[0], moms[1], pct)
    pct = (step - a1) / float(total_steps - a1)
    return cos(lr_max, low / 1e4, pct), cos(moms[1], moms[0], pct)


def ema_alpha(global_step):
    """trainer_sessd.py:316"""
    return min(1.0 - 1.0 / (global_step + 1), 0.999)


class FusedAdamEMA:
    """Adam(betas=(mom, 0.99), eps=1e-8) with decoupled decay wd on ALL parameters (bn_wd=True) and the EMA teacher,
    on the flat buffers of a student FlatParams and (optionally) a teacher FlatParams of identical layout."""

    def __init__(self, student, teacher=None, weight_decay=0.01, beta2=0.99, eps=1e-8, max_grad_norm=35.0):
        if teacher is not None and teacher.numel != student.numel:
            raise ValueError("teacher / student layouts differ")
        self.s, self.t = student, teacher
        self.wd, self.beta2, self.eps, self.max_norm = weight_decay, beta2, eps, max_grad_norm
        dev = student.data.device
        self.exp_avg = torch.zeros_like(student.data)
        self.exp_avg_sq = torch.zeros_like(student.data)
        self.norm_coef = torch.zeros(2, dtype=torch.float32, device=dev)  # [grad norm, clip coefficient], device side
        self._ws = torch.empty(int(lib.sessd_grad_clip_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.steps = 0
        # device-resident schedule state (step_dev()): iteration counter, the nine Adam / EMA constants, (lr, momentum)
        self.global_step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.args_dev = torch.zeros(9, dtype=torch.float32, device=dev)
        self.lr_mom_dev = torch.zeros(2, dtype=torch.float32, device=dev)

    def step(self, lr, beta1, global_step):
        self.steps += 1
        s = _stream()
        check(lib.sessd_grad_clip_coef(self.s.grad.data_ptr(), self.s.numel, float(self.max_norm or 0.0), self._ws.data_ptr(),
                                       self._ws.numel(), self.norm_coef.data_ptr(), s), "grad_clip_coef")
        check(lib.sessd_adam_ema_step(self.s.data.data_ptr(), self.s.grad.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), 0 if self.t is None else self.t.data.data_ptr(), self.s.numel,
                                      float(lr), float(self.wd), float(beta1), float(self.beta2), float(self.eps), self.steps,
                                      self.norm_coef.data_ptr(), float(ema_alpha(global_step)), s), "adam_ema_step")
        ops.bump_param_generation()  # student and teacher were written through raw pointers: packed-weight caches are stale

    def step_dev(self, total_steps, lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
        """The same update with the schedule evaluated ON THE DEVICE from the device iteration counter (three launches, no host
        scalar in any kernel argument): what a captured iteration replays. Advances `global_step_dev`."""
        self.steps += 1
        s = _stream()
        check(lib.sessd_one_cycle_args(self.global_step_dev.data_ptr(), int(total_steps), float(lr_max), float(moms[0]), float(moms[1]),
                                       float(div_factor), float(pct_start), float(self.wd), float(self.beta2), float(self.eps),
                                       self.args_dev.data_ptr(), self.lr_mom_dev.data_ptr(), s), "one_cycle_args")
        check(lib.sessd_grad_clip_coef(self.s.grad.data_ptr(), self.s.numel, float(self.max_norm or 0.0), self._ws.data_ptr(),
                                       self._ws.numel(), self.norm_coef.data_ptr(), s), "grad_clip_coef")
        check(lib.sessd_adam_ema_step_dev(self.s.data.data_ptr(), self.s.grad.data_ptr(), self.exp_avg.data_ptr(),
                                          self.exp_avg_sq.data_ptr(), 0 if self.t is None else self.t.data.data_ptr(), self.s.numel,
                                          self.args_dev.data_ptr(), self.norm_coef.data_ptr(), s), "adam_ema_step_dev")
        ops.bump_param_generation()


def capacity_example(example, voxel_capacity, device=None):
    """A collated example (collate_kitti + example_to_device) with its voxel tables padded to `voxel_capacity` rows and the
    device-side counts `num_voxels_dev` (and `_raw`) added: the form a captured iteration takes its inputs in (static shapes;
    the counts, not the shapes, say how many rows are voxels). Padded rows: coordinates -1, one point per voxel (so that the
    mean reader does not divide by zero), zero features. Tensors are new; copy a later batch INTO them to replay on it."""
    out = dict(example)
    for suffix in ("", "_raw"):
        if "voxels" + suffix not in example:
            continue
        vox, coo, npt = example["voxels" + suffix], example["coordinates" + suffix], example["num_points" + suffix]
        dev = vox.device if device is None else device
        n = int(vox.shape[0])
        if n > voxel_capacity:
            raise ValueError("batch has %d voxels, capacity is %d" % (n, voxel_capacity))
        v = torch.zeros((voxel_capacity,) + tuple(vox.shape[1:]), dtype=torch.float32, device=dev)
        c = torch.full((voxel_capacity, coo.shape[1]), -1, dtype=torch.int32, device=dev)
        p = torch.ones((voxel_capacity,), dtype=torch.int32, device=dev)
        v[:n], c[:n], p[:n] = vox.to(dev), coo.to(dev).int(), npt.to(dev).int()
        out["voxels" + suffix], out["coordinates" + suffix], out["num_points" + suffix] = v, c, p
        out["num_voxels_dev" + suffix] = torch.tensor([n], dtype=torch.int32, device=dev)
    if "transformation" in example and "transformation_dev" not in example:
        # the recorded global augmentation (flip / rotation / scale) the consistency loss maps the teacher's boxes with, as a
        # (B, 5) device tensor: [flipped, cos, sin, noise_rotation, noise_scale]
        from det3d.models.bbox_heads.mg_head_sessd import MultiGroupHead
        dev = device if device is not None else example["voxels"].device
        out["transformation_dev"] = MultiGroupHead.transformation_tensor(example, dev)
    return out


def allreduce_flat(flat_grad, group=None):
    """Average the flat gradient buffer over the ranks with ONE collective (dist_utils.py:38-42 divides before the
    all-reduce; kept so the summation order matches)."""
    from .dist import collectives_enabled
    if not collectives_enabled(group):
        return flat_grad
    world = dist.get_world_size(group)
    flat_grad.div_(world)
    dist.all_reduce(flat_grad, group=group)
    return flat_grad


def consistency_rampup(epoch, max_epochs=60):
    """trainer_sessd.py:306-312 sigmoid_rampup: exp(-5 (1 - min(epoch, 15)/15)^2); 1 when there are no epochs."""
    if max_epochs == 0:
        return 1.0
    phase = 1.0 - min(max(float(epoch), 0.0), 15.0) / 15.0
    return float(math.exp(-5.0 * phase * phase))


class TrainStep:
    """One SE-SSD iteration in the order of trainer_sessd.py:250-275,340-357. With loss_fn=None the loss is the reference's:
    teacher_preds = teacher(example, is_ema=[True, None]); losses = student(example, is_ema=[False, teacher_preds],
    return_loss=True) (= MultiGroupHead.loss); loss = losses['loss'][0] + consistency_weight * losses['consistency_loss'][0][0].
    A custom `loss_fn(example, student_preds, teacher_preds, consistency_weight) -> scalar` replaces it (tests / timing
    without targets). The teacher is a deep copy of the student (both start from the same checkpoint,
    trainer_sessd.py:212-217) and is never back-propagated."""

    def __init__(self, student, loss_fn=None, total_steps=1000, teacher=None, weight_decay=0.01, max_grad_norm=35.0,
                 lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
        self.student = student
        self.teacher = copy.deepcopy(student) if teacher is None else teacher
        for p in self.teacher.parameters():
            p.requires_grad_(False)
        self.loss_fn = loss_fn
        self.total_steps = int(total_steps)
        self.sched = dict(lr_max=lr_max, moms=moms, div_factor=div_factor, pct_start=pct_start)
        self.flat_s, self.flat_t = FlatParams(self.student), FlatParams(self.teacher, with_grad=False)
        self.opt = FusedAdamEMA(self.flat_s, self.flat_t, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.global_step = 0
        self.graph = None
        self.repack = ops.RepackRegistry()   # packed weights of both networks, re-packed together after every update
        self.overlap_teacher = True          # teacher forward on a second stream beside the student's
        self._side = None
        self.direct_grads = True  # gradients handed over by autograd and packed with one multi-tensor copy (FlatParams.gather_grads)
        # loss_fn=None: the reference loss as the capacity-form device op (MultiGroupHead.loss_device -> sessd_head_loss); False
        # keeps the eager torch restatement (MultiGroupHead.loss: boolean masks, host reads -- not capturable)
        self.device_loss = True
        self.pos_capacity, self.cons_capacity = None, 2048
        # SyncBN (apis/train_sessd.py:286-294): None = like the reference, on exactly when the process group has more than one
        # rank; True / False force it (True with one rank runs the split passes without a collective: same bits as the fused ones)
        self.sync_bn = None
        self.sync_bn_group = None   # process group of the SyncBN all-reduces (None: whatever ops.set_sync_bn was given, else WORLD)
        self.loss_overflow = None  # sticky device int32 bit mask: bit 0 positives / bit 1 consistency candidates beyond their capacity
        # sticky device int32 bit mask OR-ed from BOTH networks' SparseConvTensor.err after every pass (bit 0 student, bit 1 teacher):
        # SpMiddleFHD clears its own flag at the start of each pass, so without this only the last pass of the student was ever
        # visible and a run could train on truncated sparse tensors unnoticed (round-5 review item)
        self.sparse_overflow = None
        self.cw_dev = None        # consistency weight as a device scalar (a captured iteration reads it from here)
        self._cw_host = None
        self.last_record = None

    def _world(self):
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _collective(self):
        """the iteration talks to other ranks (or to itself: sessd_hip.dist.collectives_enabled, SESSD_FORCE_COLLECTIVES=1)"""
        from .dist import collectives_enabled
        return collectives_enabled()

    def _fwd_bwd(self, example, consistency_weight):
        """Teacher forward, student forward, loss, backward, gradients packed into the flat buffer -- everything of an iteration
        BEFORE the ranks exchange gradients. Returns the loss tensor."""
        self.student.train()
        self.teacher.train()  # trainer_sessd.py:321-322: both nets in train mode
        # SyncBN like the reference's distributed path (apis/train_sessd.py:286-294): statistics over the batches of all ranks
        prev_sync = ops.sync_bn_state()   # the whole state: a configured process group / reduce hook survives the iteration
        on = self.sync_bn if self.sync_bn is not None else self._collective()
        if self.sync_bn_group is not None:
            ops.set_sync_bn(on, group=self.sync_bn_group)
        else:
            ops.set_sync_bn(on)
        try:
            return self._fwd_bwd_body(example, consistency_weight)
        finally:
            ops.restore_sync_bn(prev_sync)

    def _fwd_bwd_body(self, example, consistency_weight):
        # packed weights of both networks are kept and re-packed together at the first use after an update (two launches instead
        # of ~95); the 56 BatchNorm batch counters of the two networks: one launch at the end
        with ops.batched_repack(self.repack), ops.deferred_batch_counts():
            # with SyncBN both networks issue collectives: they stay on one stream, in one order on every rank
            main = torch.cuda.current_stream() if (self.overlap_teacher and self.flat_s.data.is_cuda
                                                   and not (ops.sync_bn_active() and self._collective())) else None
            if main is not None:
                # The two forward passes are independent until the loss: the teacher's runs on a second stream (a parallel branch
                # of the captured graph), so that the launch-bound sparse half of one network fills the gaps of the other's dense
                # half. The packed weights are refreshed BEFORE the fork (both branches read them).
                if self.repack is not None:
                    self.repack.ensure_fresh()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side), torch.no_grad():
                    teacher_preds = self.teacher.forward_preds(example, raw="voxels_raw" in example)
            else:
                with torch.no_grad():
                    teacher_preds = self.teacher.forward_preds(example, raw="voxels_raw" in example)
            if self.direct_grads:
                self.flat_s.release_grads()
            else:
                self.flat_s.zero_grad()
            student_preds = self.student.forward_preds(example)
            if main is not None:
                main.wait_stream(self._side)
            self._note_sparse_overflow()
            head = self.student.bbox_head
            if self.loss_fn is None and self.device_loss and head.device_loss_covers(example, student_preds):
                # the reference loss (MultiGroupHead.loss + trainer_sessd.py:267) as one capacity-form device op: six launches,
                # no host read -- the form that can sit inside the captured iteration. The log terms stay on the device
                # (self.last_record; head.record_to_dict(rec) reads them, every N iterations).
                self._cw_dev(consistency_weight, example)
                loss, self.last_record = head.loss_device(example, student_preds, teacher_preds, self.cw_dev, unit_grad=True,
                                                          pos_capacity=self.pos_capacity, cons_capacity=self.cons_capacity)
                self.last_losses = None
                # capacity overflow of the loss (positives beyond pos_capacity / candidates beyond cons_capacity are dropped):
                # STICKY on the device, like the engine's flag -- one elementwise launch, also inside a captured iteration; read
                # and raised by check_overflow() / record() whenever the log is read (round-4 advisor finding: nothing looked)
                if self.loss_overflow is None:
                    self.loss_overflow = torch.zeros(1, dtype=torch.int32, device=self.last_record.device)
                o = ops.HEAD_LOSS_RECORD["overflow"]
                # a BIT mask: OR, not max (an iteration with flag 1 followed by one with flag 2 must leave 3: round-5 advisor finding)
                self.loss_overflow.bitwise_or_(self.last_record.detach()[o:o + 1].to(torch.int32))
            elif self.loss_fn is None:   # VoxelNet.forward(example, is_ema=[False, teacher_preds], return_loss=True) after its forward
                losses = head.loss(example, student_preds, teacher_preds)
                loss = losses["loss"][0] + losses["consistency_loss"][0][0] * consistency_weight
                self.last_losses = losses
            else:
                loss = self.loss_fn(example, student_preds, teacher_preds, consistency_weight)
        with ops.batched_repack(self.repack):
            loss.backward()
        if self.direct_grads:
            self.flat_s.gather_grads()
        return loss.detach()

    def _note_sparse_overflow(self):
        """OR both backbones' capacity-overflow flags of the passes just enqueued into the sticky mask (two small launches on the
        current stream, after the teacher's branch has joined; also inside a captured iteration)."""
        flags = [getattr(getattr(net, "backbone", None), "last_err", None) for net in (self.student, self.teacher)]
        if all(f is None for f in flags):
            return
        if self.sparse_overflow is None:
            dev = next(f for f in flags if f is not None).device
            self.sparse_overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        for bit, f in enumerate(flags):
            if f is not None:
                self.sparse_overflow.bitwise_or_((f.detach().reshape(1) != 0).to(torch.int32) << bit)

    def _update(self, device_schedule):
        """Clip + Adam + EMA on the (all-reduced) flat gradient. Returns (lr, momentum) of a host-scheduled step."""
        if device_schedule:
            self.opt.step_dev(self.total_steps, **self.sched)
            return None, None
        lr, mom = one_cycle(self.global_step, self.total_steps, **self.sched)  # lr_scheduler.step(global_step) first
        self.opt.step(lr, mom, self.global_step)
        return lr, mom

    def _iteration(self, example, consistency_weight, device_schedule):
        loss = self._fwd_bwd(example, consistency_weight)
        allreduce_flat(self.flat_s.grad)
        lr, mom = self._update(device_schedule)
        return loss, lr, mom

    def _cw_dev(self, consistency_weight, example):
        """The consistency weight of trainer_sessd.py:306-312 (a host float that changes once per epoch) as a device scalar;
        refilled only when it changes and never while capturing (capture() fills it before the capture begins)."""
        if self.cw_dev is None:
            self.cw_dev = torch.zeros((), dtype=torch.float32, device=self.flat_s.data.device)
        if self._cw_host != float(consistency_weight) and not torch.cuda.is_current_stream_capturing():
            self.cw_dev.fill_(float(consistency_weight))
            self._cw_host = float(consistency_weight)

    def check_overflow(self):
        """Synchronising: raise if ANY iteration since the last check overflowed a sparse level capacity in either network's
        SpMiddleFHD pass, or dropped positives (bit 0) or consistency candidates (bit 1) beyond the loss capacities -- such an
        iteration trained on truncated tensors / a truncated loss. Re-arms the flags."""
        sp = int(self.sparse_overflow.item()) if self.sparse_overflow is not None else 0
        if sp:
            self.sparse_overflow.zero_()
            raise RuntimeError("sparse level capacity overflow in SpMiddleFHD (mask %d: bit 0 student pass, bit 1 teacher pass) in at "
                               "least one iteration since the last check: those iterations trained on truncated sparse tensors; "
                               "raise spconv.CAPACITY_GROWTH (level 1 is the one the augmented clouds fill) or the voxel capacity" % sp)
        if self.loss_overflow is None:
            return
        v = int(self.loss_overflow.item())
        if v:
            self.loss_overflow.zero_()
            raise RuntimeError("sessd_head_loss capacity overflow (bit mask %d: bit 0 = positives beyond pos_capacity, bit 1 = "
                               "consistency candidates beyond cons_capacity): raise TrainStep.pos_capacity / cons_capacity" % v)

    def record(self):
        """The last iteration's log terms as the dict MultiGroupHead.loss returns (one host read); raises on a capacity overflow
        of any iteration since the last call."""
        self.check_overflow()
        return None if self.last_record is None else self.student.bbox_head.record_to_dict(self.last_record)

    def __call__(self, example, consistency_weight=1.0, device_schedule=False):
        """One eager iteration. device_schedule=True evaluates the OneCycle schedule and the Adam constants on the device from
        the device iteration counter (the arithmetic a captured iteration replays) instead of passing host scalars."""
        if device_schedule and int(self.opt.global_step_dev.item()) != self.global_step:
            self.opt.global_step_dev.fill_(self.global_step)
        out = self._iteration(example, consistency_weight, device_schedule)
        self.global_step += 1
        if not device_schedule:
            self.opt.global_step_dev.fill_(self.global_step)
        return out

    # ------------------------------------------------------------------ the iteration as ONE hipGraph
    def capture(self, example, consistency_weight=1.0, warmup=2):
        """Capture teacher forward + student forward / backward + all-reduce + fused update on `example` as one graph. `example`
        must be in capacity form (capacity_example: fixed shapes, device-side voxel counts, `transformation_dev`) and the loss
        free of host reads: with loss_fn=None that is the reference loss as the capacity-form device op (sessd_head_loss; round 3
        had only the eager torch restatement, whose boolean masks read shapes back), or a custom `loss_fn` on the head outputs.
        Runs `warmup` real iterations first (they count: parameters and the step counter advance). Later batches are copied INTO
        the example's tensors before replay(); replay(consistency_weight=...) refills the device scalar the graph reads."""
        if "num_voxels_dev" not in example:
            raise ValueError("capture() needs a capacity-form example (sessd_hip.train.capacity_example)")
        if self.loss_fn is None and self.device_loss and "transformation_dev" not in example:
            # without it loss_device() builds the (B, 5) augmentation tensor from host floats INSIDE the capture: a pageable
            # host-to-device copy that either aborts the capture or bakes this batch's flip / rotation / scale into every replay
            raise ValueError("capture() with the device loss needs example['transformation_dev'] (capacity_example adds it from "
                             "example['transformation']); refill it with each new batch before replay()")
        if self._collective() and (self.sync_bn if self.sync_bn is not None else True):
            raise RuntimeError("TrainStep.capture() at world size %d with SyncBN: the BatchNorm all-reduces sit inside the forward and "
                               "backward passes and cannot be captured; run the iteration eagerly (step(example)) or set "
                               "step.sync_bn = False (rank-local BatchNorm statistics) to capture it as two graphs around the "
                               "gradient all-reduce" % self._world())
        self.static_example = example
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self(example, consistency_weight, device_schedule=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.opt.global_step_dev.fill_(self.global_step)
        # the iteration's output lives OUTSIDE the graph's private memory pool (allocated before the capture, written by a captured
        # copy): a tensor inside the pool read back only after other work had run on the device was once seen overwritten
        # (tests/test_train_gpu.py, three trainers in one process; not reproduced with the output outside the pool)
        self.static_loss = torch.zeros((), dtype=torch.float32, device=self.flat_s.data.device)
        if self.loss_overflow is None:   # (warmup=0: allocated outside the graph's pool as well)
            self.loss_overflow = torch.zeros(1, dtype=torch.int32, device=self.flat_s.data.device)
        if self.sparse_overflow is None:
            self.sparse_overflow = torch.zeros(1, dtype=torch.int32, device=self.flat_s.data.device)
        ops.new_capture_epoch()   # scratch caches: nothing allocated by an earlier capture is reused in this one
        if self.repack is not None:
            self.repack.prepare()  # job tables of everything the warm-up iterations packed: uploaded before the capture
            self.repack.gen = -1   # the batched re-pack of all weights is the captured iteration's first two launches
        if self._collective():
            # THE COLLECTIVE OF A CAPTURED ITERATION (review item): an RCCL all-reduce recorded inside a hipGraph is not something
            # this stack lets us validate (one GPU per test box), so it is NOT captured. The iteration is captured as TWO graphs
            # split at the one point where the ranks talk -- [teacher fwd + student fwd + loss + backward + gradient packing] and
            # [clip + Adam + EMA] -- and replay() issues allreduce_flat (ONE collective on the flat gradient) eagerly between
            # them. SyncBN puts ~110 more collectives inside the first part: that configuration runs eagerly.
            # (checked at the top of capture(): SyncBN off here)
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                loss = self._fwd_bwd(example, consistency_weight)
                self.static_loss.copy_(loss)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool()):
                self._update(True)
            self.opt.steps -= 1
            self.graph = (ga, gb)
            return self.graph
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, _, _ = self._iteration(example, consistency_weight, True)
            self.static_loss.copy_(loss)
        # the captured launches did not run: undo the host-side bookkeeping of the capture pass
        self.opt.steps -= 1
        self.graph = g
        return g

    def replay(self, consistency_weight=None):
        """One captured iteration on whatever the static example's tensors hold now. Returns the (device) loss tensor; with the
        device loss, `self.last_record` is the 64-float device log of that iteration."""
        if consistency_weight is not None and self.cw_dev is not None and float(consistency_weight) != self._cw_host:
            self.cw_dev.fill_(float(consistency_weight))
            self._cw_host = float(consistency_weight)
        if isinstance(self.graph, tuple):   # world size > 1: two graphs around the eager gradient all-reduce (capture())
            self.graph[0].replay()
            allreduce_flat(self.flat_s.grad)
            self.graph[1].replay()
        else:
            self.graph.replay()
        self.global_step += 1
        self.opt.steps += 1
        ops.bump_param_generation()  # parameters changed under the packed-weight caches of any eager code that runs next
        return self.static_loss

"""The training step's exchange + update for data-parallel replicas.

Reference: `DistOptimizerHook.after_train_iter` (mmdet/core/utils/dist_utils.py:44-58) = zero_grad, backward,
`allreduce_grads` (flatten per dtype, all_reduce, / world_size, copy back; :9-41), clip_grad_norm_(max_norm 35),
SGD step.  Here one process per GPU, RCCL through torch.distributed (backend "nccl"; "gloo" in the CPU tests):

  * `FlatParams` owns ONE flat f32 buffer for the parameters, one for the gradients and one for the momentum; every
    nn.Parameter (and its .grad) is a view into them, so there is nothing to flatten, copy back or bucket: the exchange
    is a single `all_reduce` of the gradient buffer (the reference makes the same single call per dtype);
  * the division by world_size and the clip coefficient are folded into the update kernel (`hvr_sgd_step`), which reads
    the clip norm from a device-side reduction -- no host synchronisation anywhere in the step.
"""
import torch
import torch.distributed as dist

from . import native


class FlatParams(object):
    """Re-homes the trainable parameters of `module` into one flat f32 buffer (parameters keep their shapes as views)."""

    def __init__(self, module):
        from .backbone import PackedMixin
        # inference-layout weight caches (BN folded, packed per dtype) of the modules whose parameters this object updates:
        # they are functions of the parameters and must be rebuilt after every update (see sgd_step)
        def owned(m):  # the parameters a packed module folds into its cache: its subtree minus nested packed modules
            for p in m.parameters(recurse=False):
                yield p
            for c in m.children():
                if not isinstance(c, PackedMixin):
                    for p in owned(c):
                        yield p
        self._packed_modules = [m for m in module.modules() if isinstance(m, PackedMixin) and any(p.requires_grad for p in owned(m))]
        self.params = [p for p in module.parameters() if p.requires_grad]
        assert self.params and all(p.dtype == torch.float32 for p in self.params), 'f32 parameters only'
        dev = self.params[0].device
        al = 64                                            # every view starts on a 256-byte boundary (GEMM operands need 16)
        n = sum((p.numel() + al - 1) // al * al for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)   # padding stays zero: zero gradient, zero update
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(n, dtype=torch.float32, device=dev)
        self.steps = 0
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            off += (k + al - 1) // al * al

    def zero_grad(self):
        self.grad.zero_()

    def allreduce_grads(self):
        """Sum over the replicas (the reference's allreduce_grads without its division, which the update applies)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grad)
            return dist.get_world_size()
        return 1

    def sgd_step(self, lr, momentum=0.9, weight_decay=1e-4, max_norm=35.0, world_size=1):
        """clip_grad_norm_(max_norm) on the averaged gradient + torch.optim.SGD step, on the device (hvr_sgd_step)."""
        native.sgd_step(self.flat, self.grad, self.momentum, lr, momentum, weight_decay, grad_scale=1.0 / world_size,
                        max_norm=max_norm, first_step=self.steps == 0)
        self.steps += 1
        # The update wrote the parameters in place through the flat buffer: every packed (graph-free) forward -- res5 in
        # HNMBRCNN.get_triplet_patches, the RPN's proposal pass, an evaluation after training -- must see the new weights,
        # as the reference's modules do (they hold no second copy).
        for m in self._packed_modules:
            m._drop_packed()


def train_iteration(flat, loss_fn, lr, momentum=0.9, weight_decay=1e-4, max_norm=35.0, overlap_wgrad=False, fewrow=False):
    """One iteration in the reference's order (dist_utils.py:52-58): zero_grad, backward, all-reduce, clip, step.
    loss_fn() builds the graph and returns the scalar to differentiate.  overlap_wgrad: conv weight gradients run on a
    second HIP stream and land directly in the flat gradient buffer (train_ops.wgrad_overlap); joined before the exchange.
    Off by default: at three frames per iteration the step is bound by the host's enqueue rate, and the extra events / stream
    switches cost more than the overlap returns (22.1 vs 20.5 ms measured); it pays once the per-rank batch grows."""
    from . import train_ops
    flat.zero_grad()
    train_ops.prep_begin(flat)          # every trainable layer's operands for this iteration in two launches (recorded on the first)
    try:
        # fewrow: the few-row forms (native.fewrow_split: K sliced across the waves of a workgroup, kpar.hip, or across workgroups) for the
        # step's products -- a rank's batch is three frames / a few hundred RoIs.  Measured without gain at that size (HVR 17.3 / 17.9 ms,
        # SELSA 15.5 / 15.4 ms off / on, profiles/r06_train_steps.txt: the step was bound by launch cadence, not by these kernels): off
        with native.fewrow_split(bool(fewrow)):
            loss = loss_fn()
            prev = train_ops.wgrad_overlap(overlap_wgrad)
            prev_direct = train_ops.wgrad_direct(True, pristine={id(p) for p in flat.params})   # weight gradients land straight in flat's (zeroed) gradient buffer: this scope only
            prev_defer = train_ops.wgrad_defer(not overlap_wgrad)   # conv weight gradients of equal shape parked, then one batched product per shape
            try:
                loss.backward()
                train_ops.wgrad_flush()
            finally:
                train_ops.wgrad_defer(prev_defer)
                train_ops.wgrad_direct(prev_direct)
                train_ops.wgrad_overlap(prev)
                train_ops.join_wgrad()
    finally:
        train_ops.prep_end()            # the update below makes the table's operands stale
    world = flat.allreduce_grads()
    flat.sgd_step(lr, momentum, weight_decay, max_norm, world)
    return loss


class C4Prefetcher(object):
    """The frozen backbone of the NEXT batch on a second HIP stream while the current batch trains.

    HNMBRCNN's training step computes C4 under no_grad and reuses it as a constant (hnmb_rcnn.py:269-283): backbone parameters get no
    gradient and no update there (enable_training keeps them out of the flat parameter set), so the backbone pass over batch i + 1
    depends on nothing batch i's step writes.  It is 15 frames of chip-filling convs (a third of the step's kernel time), the rest of the
    step is hundreds of few-row kernels with four host reads in between: run side by side the big kernels fill what the small ones and
    the read-backs leave idle.  Same work per iteration, same numbers (the C4 map is bit-identical to the in-line pass's).

        pre = C4Prefetcher(model); pre.start(batch[0]['img'])
        for i, data in enumerate(batches):
            c4 = pre.take()
            if i + 1 < len(batches): pre.start(batches[i + 1]['img'])
            train_detector_iteration(model, flat, dict(data, c4=c4), lr)
    """

    def __init__(self, model):
        assert not any(p.requires_grad for p in model.backbone.parameters()), 'the prefetched pass is only valid for a FROZEN backbone'
        self.model, self.stream, self.pending = model, None, None

    def start(self, img):
        dev = img.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(main)                       # img (and the packed weights' first use) are ordered behind the caller's stream
        with torch.cuda.stream(self.stream), torch.no_grad():
            self.stream.wait_event(ready)
            c4 = self.model.extract_feat(img)[0]
            done = torch.cuda.Event()
            done.record(self.stream)
        img.record_stream(self.stream)
        self.pending = (c4, done)

    def take(self):
        """-> the C4 maps of the batch given to the last start(); the current stream waits for them."""
        c4, done = self.pending
        self.pending = None
        main = torch.cuda.current_stream(c4.device)
        main.wait_event(done)
        c4.record_stream(main)
        return c4


def parse_losses(losses):
    """mmdet/apis/train.py:17-34 without the per-entry `.item()` host copies: -> (loss to differentiate, {name: 0-dim tensor});
    the log values stay on the device (read them once per logging interval, not once per entry per iteration)."""
    log_vars = {}
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(name))
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, {k: v.detach() for k, v in log_vars.items()}


def train_detector_iteration(model, flat, data, lr, momentum=0.9, weight_decay=1e-4, max_norm=35.0, overlap_wgrad=False):
    """batch_processor + the optimizer hook (mmdet/apis/train.py:37-54, mmdet/core/utils/dist_utils.py:52-58) for one rank's
    batch: model(**data) -> parse_losses -> backward -> one flat all-reduce over RCCL -> fused clip + SGD.  -> log_vars."""
    box = {}

    def loss_fn():
        loss, box['log'] = parse_losses(model(**data))
        return loss

    train_iteration(flat, loss_fn, lr, momentum, weight_decay, max_norm, overlap_wgrad)
    return box['log']

"""The train step as ONE hipGraph: forward + rotated-GIoU loss + backward + optimizer captured once per (batch shape, target
count) and replayed -- the host's share of a step drops from ~660 ctypes launches (13.7 ms of Python against a 19.5 ms step on
one MI355X; with eight ranks on one host the margin is thinner) to a buffer copy and one hipGraphLaunch.

The reference has no counterpart (its step is eager PyTorch, src/train.py:205-235); this is the MI355X-native way to issue a
step whose shapes are static.  What makes the step capturable: every launch of the engine is stream-ordered with no host
synchronisation after the first (tuning) steps; the per-step scalars the reference keeps on the host live on the device
(optim.FusedAdam(capturable=True): step count and learning rates; Darknet's metrics); the engine's alternating statistics
tables restart from a defined state every pass; side streams fork from and join the capturing stream through events.

    step = GraphedTrainStep(model, optimizer)          # optimizer = FusedAdam(..., capturable=True)
    loss = step(images, targets)                       # device scalar; model.yolo_layers[i].metrics as usual

STATUS (round 3, ROCm 7.2 / MI355X): correct -- parameters, running statistics and losses bit-identical to eager steps in the
deterministic mode (tests/test_gpu_r3.py::test_graphed_train_step_equals_eager_steps) -- but NOT faster: one replay of the 660-node
graph with its 110 forks to the weight-gradient stream takes 33.1 ms where the eager step takes 19.2 ms, the runtime spends more
per node than a launch costs.  bench.py therefore measures the eager step (--graph 1 selects this class).  Requires the engine's
two-stream backward (the default): with CY_WGRAD_SIDE_STREAM=0 a replay raised a GPU memory access fault that was not tracked down,
and the constructor refuses that configuration.

A new (image shape, number of target rows) captures a new graph (KITTI batches differ in their number of boxes; pad the
target rows to a few bucket sizes with rows of sample index -1 to bound the number of graphs -- such rows are rejected by
cy_yolo_loss and only count as `errors` in the metrics).  Not for data-parallel runs yet: the gradient all-reduce hooks are
not part of the capture.  Not with optim.DynamicLossScale either (refused): the overflow scan sits between backward() and
step() outside the capture; fp16 runs through this class use a static ``loss_scale``.
"""
import torch

from . import ops


class GraphedTrainStep:
    """``warmup``: how many batches of a new (image shape, target rows) run EAGERLY before that shape is captured (default 1).
    They are ordinary training steps on the batches the caller passes -- nothing is stepped twice -- and double as what a
    capture needs to have happened once: kernel / tile choices, workspaces, optimizer state.  ``warmup=0`` captures at first
    sight and is for callers that have already run an eager step of that shape themselves."""

    def __init__(self, model, optimizer, warmup=1, max_graphs=32):
        self.model = getattr(model, 'module', model)
        if self.model is not model:
            raise ops.CyoloError('GraphedTrainStep captures a single-process step (no gradient all-reduce inside the graph)')
        if not getattr(optimizer, 'capturable', False):
            raise ops.CyoloError('GraphedTrainStep needs a capturable optimizer: FusedAdam(..., capturable=True)')
        import os
        if os.environ.get('CY_WGRAD_SIDE_STREAM', '1') == '0' and os.environ.get('CY_GRAPH_SINGLE_STREAM_OK') != '1':      # (the override: fault hunting)
            raise ops.CyoloError('GraphedTrainStep needs the two-stream backward (CY_WGRAD_SIDE_STREAM=0 is set): see the module docstring')
        if getattr(optimizer, 'skip_flag', None) is not None:
            # a DynamicLossScale is attached: its cy_grad_nonfinite scan runs between backward() and step(), outside what this
            # class captures, so replays would consult a flag nobody refreshes -- fp16 overflow protection would be silently off
            raise ops.CyoloError('GraphedTrainStep does not capture optim.DynamicLossScale (the non-finite scan is not part of the '
                                 'graph): use bf16, a static loss_scale, or the eager step')
        self.opt, self.warmup, self.max_graphs = optimizer, int(warmup), int(max_graphs)
        self._graphs = {}
        self._seen = {}
        self.replays = 0

    def _eager(self, x, tg):
        self.opt.zero_grad(set_to_none=True)
        loss, _ = self.model(x, tg)
        loss.backward()
        self.opt.step()
        return loss

    def _capture(self, x, tg):
        ops.check_device_tensor(x, 'GraphedTrainStep')
        sx, st = x.clone(), tg.clone()
        for e in self.model._engines.values():
            e.pin_retired = True          # a captured graph has buffer addresses baked in: nothing an engine retires may be freed
        self.model.engine_budget_bytes = None      # ... and no engine may be evicted under it
        torch.cuda.synchronize(x.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = self._eager(sx, st)
        return dict(graph=g, x=sx, tg=st, loss=loss)

    def __call__(self, x, targets):
        if getattr(self.opt, 'skip_flag', None) is not None:
            raise ops.CyoloError('a DynamicLossScale was attached to the optimizer after GraphedTrainStep was built: its overflow '
                                 'check is not part of the captured step')
        targets = targets.to(x.device).float().contiguous()
        key = (tuple(x.shape), int(targets.shape[0]), self.model.training)
        rec = self._graphs.get(key)
        if rec is None:
            seen = self._seen.get(key, 0)
            if seen < self.warmup or len(self._graphs) >= self.max_graphs:
                self._seen[key] = seen + 1
                return self._eager(x, targets)        # a real step of this batch (new shape, or too many shapes already)
            rec = self._graphs[key] = self._capture(x.float().contiguous(), targets)
            # (the capture itself does not execute the step: fall through to the replay for this batch)
        rec['x'].copy_(x, non_blocking=True)
        rec['tg'].copy_(targets, non_blocking=True)
        self.opt.refresh_groups()
        rec['graph'].replay()
        self.opt.note_replayed()
        self.replays += 1
        return rec['loss']

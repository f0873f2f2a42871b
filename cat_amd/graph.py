"""hipGraph capture of one whole distillation step.

`optimize_parameters` issues ~1 000 (inception) to ~2 700 (SPADE) kernel launches from Python; at 256 x 512 x batch 4 the SPADE
step is HOST-bound (measured: 85-105 ms to issue vs ~65 ms of GPU work).  The step is static -- same shapes, same launch
sequence, optimiser scalars resident in HBM (cat_adam_step_dev) -- so it is captured ONCE into a hipGraph (through
torch.cuda.CUDAGraph, which also pins the caching-allocator pool the captured kernels address) and replayed per batch:

    step = GraphedStep(model, example_batch)        # 3 eager warm-up steps + capture
    step(batch)                                     # copies the batch into the static input buffers, one graph launch

What is captured is exactly model.set_input + model.optimize_parameters, including the side-stream branches (fork / join
events become graph dependencies) and both Adam updates.  Loss tensors are graph outputs: `model.get_current_losses()` reads
them after a replay.  Data-parallel steps (RCCL collectives between the backward passes) stay eager."""
import os

import torch


class GraphedStep:
    def __init__(self, model, example_batch, warmup=3):
        if getattr(model, 'dp', None) is not None:
            raise RuntimeError('GraphedStep: data-parallel steps are not captured (collectives between the backward passes)')
        self.model = model
        # static input buffers live on the model's device (a DataLoader batch is a host tensor: its H2D copy is done per call,
        # outside the graph); non-tensor entries (image paths) are refreshed per call
        dev = getattr(model, 'device', None) or torch.device('cuda', torch.cuda.current_device())
        self.static = {k: (v.to(dev).clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(warmup):          # settles lazy state: flattened optimiser buffers, weight re-housing, workspaces
                model.set_input(self.static)
                model.optimize_parameters(i)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            model.set_input(self.static)
            model.optimize_parameters(warmup)
        self.replays = 0
        # the graph's outputs are the tensors the model's attributes were bound to DURING the capture (loss terms, fake images, input
        # fields, tapped activations).  An eager fallback step rebinds those attributes to fresh tensors; the bindings are restored before
        # the next replay's results are read, otherwise get_current_losses() / get_current_visuals() would keep returning the values of
        # that one eager step
        self._bound = self._graph_bindings()
        self._stale = False

    def _graph_bindings(self):
        m = self.model
        names = set(getattr(m, 'visual_names', [])) | {'real_A', 'real_B', 'input_semantics', 'Sfake_B', 'Tfake_B', 'loss_G', 'loss_D'}
        bound = {k: v for k, v in vars(m).items() if k.startswith('loss_') or k in names}
        taps = {k: dict(getattr(m, k)) for k in ('Sacts', 'Tacts') if isinstance(getattr(m, k, None), dict)}
        return bound, taps

    def _rebind(self):
        bound, taps = self._bound
        for k, v in bound.items():
            setattr(self.model, k, v)
        for k, d in taps.items():
            getattr(self.model, k).update(d)
        self._stale = False

    def __call__(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v) and (k not in self.static or tuple(v.shape) != tuple(self.static[k].shape)):
                # e.g. the smaller last batch of an epoch: shapes are baked into the capture -> run this step eagerly
                self.model.set_input(batch)
                self.model.optimize_parameters(self.replays)
                self._stale = True
                return
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
            else:
                self.static[k] = v
        for opt in self.model.optimizers:       # LambdaLR / manual lr changes since the capture
            if hasattr(opt, 'sync_hyper_for_replay'):
                opt.sync_hyper_for_replay()
        self.graph.replay()
        if self._stale:
            self._rebind()
        # host-side field set_input would have refreshed (base_inception_distiller.py set_input: A_paths / B_paths / path)
        opt_ = getattr(self.model, 'opt', None)
        key = 'A_paths' if getattr(opt_, 'direction', 'AtoB') == 'AtoB' else 'B_paths'
        paths = batch.get('path', batch.get(key))
        if paths is not None:
            self.model.image_paths = paths
        for opt in self.model.optimizers:
            opt.note_graph_replay()
        self.replays += 1


class GraphedDPStep:
    """The InceptionDistiller step as hipGraph SEGMENTS: the frozen teacher's forward is its own graph on a side stream, and -- with data
    parallelism -- the two collectives sit between the segments, so that the N-GPU job launches its kernels the way the one-GPU
    headline run does (InceptionDistiller._optimize_parameters_dp is the eager form of the same schedule):

        main : g_in  (set_input: layout kernels into static buffers)
        side : g_T   (frozen-teacher forward of this batch -- overlaps the wait for the previous step's student all-reduce)
        main : [wait G bucket of the previous step -> Adam G (one eager launch)] -> g_A (student forward, backward_D)
               -> RCCL all-reduce of the D bucket -> join side -> g_B (Adam D, backward_G) -> async all-reduce of the G bucket
        g_A is THREE graphs when the discriminator can be cut (InceptionDistiller._dp_first_stages): after each one the gradient slice that
        became final (first the 512 -> 1024 layer's 8.4 M of 11 M parameters) is all-reduced while the next graph runs.

    Without a reducer (one GPU, `model.dp is None`) the same four graphs run with no collective in between and Adam G at the end of g_B:
    the teacher then simply overlaps the student forward + discriminator step.  g_T has its own memory pool (it runs concurrently with
    g_A); g_A / g_B share one (the autograd tape of the student forward is consumed by backward_G) and are always replayed in capture
    order.  Collectives are never captured: RCCL enqueues them between the graph launches with ordinary stream ordering.  The SPADE step
    exchanges SynchronizedBatchNorm statistics inside its passes (dozens of small collectives) and stays eager."""

    def __init__(self, model, example_batch, warmup=3):
        if not hasattr(model, '_dp_first'):
            raise RuntimeError('GraphedDPStep: needs an InceptionDistiller')
        if getattr(model, 'dp', None) is not None and not getattr(model, 'dp_overlap', True):
            # the segments ARE the overlapped schedule (deferred Adam G, teacher on the side stream); a model configured with
            # enable_data_parallel(overlap=False) keeps its eager serial schedule -- callers fall back to plain model.optimize_parameters
            raise RuntimeError('GraphedDPStep: the model was configured with dp_overlap=False; its serial schedule is launched eagerly')
        self.model = model
        self.dp = getattr(model, 'dp', None) is not None
        dev = model.device
        if getattr(model, '_side_stream', None) is None:
            model._side_stream = torch.cuda.Stream(device=dev)
        self.static = {k: (v.to(dev).clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream()
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):
            for i in range(warmup):          # eager steps (with a reducer: the DP schedule, collectives included -- every rank does the same)
                model.set_input(self.static)
                model.optimize_parameters(i)
            model.finish_pending()
        cur.wait_stream(warm)
        torch.cuda.synchronize()
        if self.dp:
            from . import parallel
            parallel.settle_collectives()      # the RCCL watchdog must not poll a warm-up collective's event inside a capture
        side = model._side_stream
        kw = dict(capture_error_mode='thread_local')      # RCCL's watchdog thread may query events while this thread captures
        self.g_in, self.g_T, self.g_B = (torch.cuda.CUDAGraph() for _ in range(3))
        with torch.cuda.graph(self.g_in, **kw):
            model.set_input(self.static)
        with torch.cuda.graph(self.g_T, stream=side, **kw):
            model._dp_teacher()
        # student forward + backward_D: one graph, or -- with a reducer and a discriminator that can be cut -- one graph per gradient-ready
        # stage, so that each slice of the D bucket is all-reduced while the next stage's graph runs.  The stages share one memory pool
        # (the autograd tape crosses them) and are always replayed in capture order.
        stages = model._dp_first_stages() if self.dp else [(model._dp_first, (None, None))]
        self.g_A, self.slices = [], []
        for run, sl in stages:
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, **(kw if not self.g_A else dict(kw, pool=self.g_A[0].pool()))):
                run()
            self.g_A.append(gk)
            self.slices.append(sl)
        with torch.cuda.graph(self.g_B, pool=self.g_A[0].pool(), **kw):
            model._dp_second(warmup)
            if not self.dp:
                model.optimizer_G.step()
        torch.cuda.synchronize()
        self._bound = GraphedStep._graph_bindings(self)
        self._stale = False
        self.replays = 0

    _rebind = GraphedStep._rebind

    def __call__(self, batch):
        m = self.model
        for k, v in batch.items():
            if torch.is_tensor(v) and (k not in self.static or tuple(v.shape) != tuple(self.static[k].shape)):
                m.set_input(batch)             # another shape (last batch of an epoch): this step runs eagerly on every rank
                m.optimize_parameters(self.replays)
                self._stale = True
                return
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
            else:
                self.static[k] = v
        for opt in m.optimizers:
            if hasattr(opt, 'sync_hyper_for_replay'):
                opt.sync_hyper_for_replay()
        main = torch.cuda.current_stream(m.device)
        side = m._side_stream
        self.g_in.replay()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.g_T.replay()
        t_done = side.record_event()
        if self.dp:
            m.finish_pending()                  # wait for the previous step's G bucket, Adam G (eager: one launch)
        pend = []
        for gk, (lo, hi) in zip(self.g_A, self.slices):
            gk.replay()
            if self.dp:
                pend.append(m.dp.reduce_slice_async(m.optimizer_D, lo, hi) if lo is not None else m.dp.reduce_async(m.optimizer_D))
        for w in pend:
            w.wait()
        main.wait_event(t_done)
        self.g_B.replay()
        m.optimizer_D.note_graph_replay()       # Adam D ran inside g_B
        if self.dp:
            m._pending_G = m.dp.reduce_async(m.optimizer_G)
        else:
            m.optimizer_G.note_graph_replay()   # ... and so did Adam G
        if self._stale:
            self._rebind()
        opt_ = getattr(m, 'opt', None)
        key = 'A_paths' if getattr(opt_, 'direction', 'AtoB') == 'AtoB' else 'B_paths'
        paths = batch.get('path', batch.get(key))
        if paths is not None:
            m.image_paths = paths
        self.replays += 1

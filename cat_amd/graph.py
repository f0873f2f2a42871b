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
import torch


class GraphedStep:
    def __init__(self, model, example_batch, warmup=3):
        if getattr(model, 'dp', None) is not None:
            raise RuntimeError('GraphedStep: data-parallel steps are not captured (collectives between the backward passes)')
        self.model = model
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
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

    def __call__(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        for opt in self.model.optimizers:
            opt.note_graph_replay()
        self.replays += 1

"""HIP-graph replay of a forward pass, and of a whole training step (forward + backward).

The kernels of this library never synchronise, allocate or touch the host
(include/sputnik_hip.h), so a whole ``SparseAttention`` / ``SparseLinear``
forward can be captured into one hipGraph and replayed with a single launch:
at the attention shapes the PyTorch glue between the kernels (transposes,
reshapes, allocator calls) costs as much as the kernels themselves.

    fast = capture_forward(layer, x, x, x)     # warm-up + capture, inputs of fixed shape
    y = fast(x2, x2, x2)                       # copies into the static inputs, replays
"""
import torch


class GraphedForward:
    """Callable replaying ``module(*inputs)`` from a captured graph.  Inputs must
    keep the example's shapes and dtypes; the returned tensor is the graph's
    static output (clone it to keep it across calls)."""

    def __init__(self, module, *example_inputs, warmup=3):
        if not all(torch.is_tensor(t) and t.is_cuda for t in example_inputs):
            raise ValueError("capture_forward needs GPU tensors as example inputs")
        self.module = module
        # Inputs that are ONE tensor in the example (self-attention: module(x, x, x)) stay
        # one static buffer: the module takes its shared-input paths (the three projections
        # as one group launch) in the captured graph as it does eagerly, and a replay copies
        # the input once, not three times.
        clones = {}
        self.static_inputs = [clones.setdefault(id(t), t.detach().clone()) for t in example_inputs]
        self._alias = [id(t) for t in example_inputs]
        side = torch.cuda.Stream(device=self.static_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # builds plans, fills the allocator's pools
                module(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_output = module(*self.static_inputs)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        first, copied = {}, set()
        for dst, src, alias in zip(self.static_inputs, inputs, self._alias):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("input shape / dtype differs from the captured example")
            if first.setdefault(alias, src) is not src:
                raise ValueError("inputs that were one tensor in the captured example must be "
                                 "one tensor in every replay")
            # (callers that fill `static_inputs` in place skip the copy)
            if alias not in copied and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
            copied.add(alias)
        self.graph.replay()
        return self.static_output


def capture_forward(module, *example_inputs, warmup=3):
    return GraphedForward(module, *example_inputs, warmup=warmup)


class GraphedTrainingStep:
    """Forward AND backward of ``module(*inputs).backward(grad_output)`` as one hipGraph
    (static shapes, warm caches): the ~27 launches of a `SparseAttention` training step
    and all the Python between them -- autograd's graph walk, the Functions' bookkeeping,
    allocator calls -- become ONE launch, so the step no longer depends on how fast the
    host feeds the queue (the same eager step read 0.58 ms on one box and 0.67
    on another).

        step = capture_training_step(attn, x, x, x, grad_output=g)
        out = step(x2, x2, x2, grad_output=g2)    # copies into the static buffers, replays
        step.input_grads[0], step.param_grads      # the graph's static tensors (also p.grad)

    The gradients are OVERWRITTEN by every replay (they were None when the graph was
    captured), not accumulated; clone what must outlive the next call.  The module's
    static topologies must be registered (they are for this package's modules), so that
    transposed topologies and kernel plans come from the caches: the warm-up builds them,
    the captured step then contains kernels only."""

    def __init__(self, module, *example_inputs, grad_output, warmup=3):
        if not all(torch.is_tensor(t) and t.is_cuda for t in example_inputs):
            raise ValueError("capture_training_step needs GPU tensors as example inputs")
        self.module = module
        clones = {}
        self.static_inputs = [
            clones.setdefault(id(t), t.detach().clone().requires_grad_(t.is_floating_point()))
            for t in example_inputs]
        self._alias = [id(t) for t in example_inputs]
        self.static_grad_output = grad_output.detach().clone()
        self.params = [p for p in module.parameters() if p.requires_grad]
        seen = set()
        unique_inputs = [t for t in self.static_inputs
                         if t.requires_grad and not (id(t) in seen or seen.add(id(t)))]
        wanted = unique_inputs + self.params

        # torch.autograd.grad, not .backward(): no AccumulateGrad nodes take part (theirs is
        # the stream they were first used on -- the default stream, if the module has trained
        # eagerly before -- which would put the accumulation outside the captured stream)
        def run():
            out = module(*self.static_inputs)
            return out, torch.autograd.grad(out, wanted, self.static_grad_output, allow_unused=True)

        side = torch.cuda.Stream(device=self.static_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # builds plans and transposes, fills the allocator's pools
                run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_output, grads = run()
        self.static_output = self.static_output.detach()
        self.input_grads = list(grads[:len(unique_inputs)])
        self.param_grads = list(grads[len(unique_inputs):])
        # the parameters' .grad point at the graph's static gradients: an optimizer step
        # after a replay reads them as after an eager backward
        for p, g in zip(self.params, self.param_grads):
            p.grad = g

    def __call__(self, *inputs, grad_output=None):
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        first, copied = {}, set()
        with torch.no_grad():
            for dst, src, alias in zip(self.static_inputs, inputs, self._alias):
                if dst.shape != src.shape or dst.dtype != src.dtype:
                    raise ValueError("input shape / dtype differs from the captured example")
                if first.setdefault(alias, src) is not src:
                    raise ValueError("inputs that were one tensor in the captured example must be "
                                     "one tensor in every replay")
                if alias not in copied and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
                copied.add(alias)
            if grad_output is not None and grad_output.data_ptr() != self.static_grad_output.data_ptr():
                self.static_grad_output.copy_(grad_output)
        self.graph.replay()
        return self.static_output


def capture_training_step(module, *example_inputs, grad_output, warmup=3):
    return GraphedTrainingStep(module, *example_inputs, grad_output=grad_output, warmup=warmup)

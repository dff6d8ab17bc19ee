"""HIP-graph replay of a forward pass.

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

"""Whole-step HIP graphs for the TIM hot path.

The reference leaves launch overhead to eager PyTorch (and, on NVIDIA, to whatever the user wraps around it); on MI355X the
encoder step of a small model (C1: d_model 256, 2 layers) is launch-bound - ~330 kernels at ~5 us of host time each - so this
module captures one full step (time MLP, encoder forward, loss, backward and optionally the optimizer) into a single
`hipGraph` through `torch.cuda.graph` and replays it: one host call per step.

What makes the HIP path capturable:
  * every `timhip_*` launch goes to torch's current stream (the capture stream); the weight-gradient side stream forks from
    and joins back into it with events, which capture as graph edges; so does the data-parallel wrapper's comm stream, whose
    reduce-scatter / all-gather collectives RCCL records as graph nodes (tim_amd/dp.py, collective "rs_ag": tested on a
    one-rank RCCL group, tools/dp_graph_check.py);
  * workspaces / saved activations come from torch's allocator, so they land in the graph's private pool;
  * dropout seeds are launch arguments, which a graph would freeze - `functional.graph_safe_dropout` moves the per-step part
    of the seed into a device word that a node of the graph advances (include/timhip.h: timhip_dropout_salt);
  * the operand-dtype weight copies are rebuilt by a cast kernel at the head of the captured step, so an optimizer update
    between (or inside) replays is always picked up.

Usage (the shape of the reference's train loop, scripts/train.py:250-330):

    static = {k: torch.empty_like(v) for k, v in first_batch.items()}
    def step():
        optimizer.zero_grad(set_to_none=False)
        loss = criterion(model(...static...))
        loss.backward()
        return loss
    graphed = GraphedStep(model, step)
    for batch in loader:
        for k in static: static[k].copy_(batch[k])
        loss = graphed()          # replay; `loss` is the captured output tensor, refreshed in place
        optimizer.step()
"""
import torch

from . import functional as F


def _count_graph_nodes(raw_graph):
    """(all nodes, kernel nodes) of a captured hipGraph_t (an integer handle as torch's `raw_cuda_graph()` returns it)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n = C.c_size_t(0)
    g = C.c_void_p(int(raw_graph))
    if hip.hipGraphGetNodes(g, None, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    arr = (C.c_void_p * n.value)()
    if hip.hipGraphGetNodes(g, arr, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    kernels = 0
    for i in range(n.value):
        t = C.c_int(-1)
        if hip.hipGraphNodeGetType(C.c_void_p(arr[i]), C.byref(t)) == 0 and t.value == 0:   # hipGraphNodeTypeKernel = 0
            kernels += 1
    return int(n.value), kernels


class GraphedStep:
    """Capture `fn()` - a closure over static input tensors - once and replay it.

    fn      callable without arguments running forward + backward (+ anything else that is capturable) of `model`
    warmup  eager runs on a side stream before capture (allocator warm-up, weight copies, gradient buckets)
    count_nodes  keep the captured hipGraph until it has been walked: `kernel_nodes` / `nodes` = the launches of one step,
            counted (hipGraphGetNodes + hipGraphNodeGetType), not assumed (bench.py's `launches_per_step`)
    """

    def __init__(self, model, fn, warmup=3, count_nodes=False):
        inner = model.module if hasattr(model, "module") else model
        if getattr(model, "active", False) and getattr(model, "collective", None) == "a2a":
            # all_to_all_single is send / receive pairs underneath; captured, they hang or crash hipStreamEndCapture on this
            # stack (ROCm 7.0 / RCCL 2.26, profiles/r05_rccl_capture_probe.txt) - a segmentation fault, not an exception
            raise RuntimeError("GraphedStep: the data-parallel wrapper exchanges its buckets with all_to_all_single, which cannot be "
                               "captured in a HIP graph on this stack; construct tim_amd.dp.DataParallel with collective='rs_ag' "
                               "(the default for an fp32 wire over RCCL) or run the step eagerly")
        self.rt = inner.rt
        inner._ws_pinned = True   # the graph keeps raw pointers into the model's workspaces: they are retired, never freed
        dev = next(inner.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep needs the model on a GPU; the HIP path has no CPU fallback")
        self.salt = F.graph_safe_dropout(dev)  # kept alive: the captured kernels hold its address
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.rt.invalidate_weights()
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        n_before = len(self.rt._gs_captured)   # (blocks of an earlier bare capture stay with the runtime)
        self.kernel_nodes = self.nodes = None
        try:
            self.graph = torch.cuda.CUDAGraph(keep_graph=True) if count_nodes else torch.cuda.CUDAGraph()
        except TypeError:   # (a torch without keep_graph: no count)
            self.graph, count_nodes = torch.cuda.CUDAGraph(), False
        self.rt.invalidate_weights()  # the cast of every weight is part of the captured step
        # With a process group alive, RCCL's watchdog thread polls the events of the warm-up steps' collectives while this thread
        # captures; under the default (global) capture mode that hipEventQuery is an error on the OTHER thread ("operation not
        # permitted when stream is capturing": the watchdog dies and takes the process with it - seen on a one-rank group,
        # tools/dp_graph_check.py).  Thread-local mode checks the capturing thread only.
        mode = "global"
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                mode = "thread_local"
        except Exception:  # noqa: BLE001
            pass
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.out = fn()
        if count_nodes:
            try:
                self.nodes, self.kernel_nodes = _count_graph_nodes(self.graph.raw_cuda_graph())
            except Exception:  # noqa: BLE001  (the count is a report, never a reason to lose the step)
                self.nodes = self.kernel_nodes = None
            self.graph.instantiate()
        # the gradient-scale blocks (non-finite flags) of the captured backward passes live and die with this object
        self._gs_blocks = self.rt.adopt_captured(since=n_before)
        # the gradient tensors the captured backward writes: re-attached on every replay, so eager steps in between (which
        # may re-allocate .grad) do not detach the parameters from the graph's results
        self._grads = [(p, p.grad) for p in inner.parameters() if p.grad is not None]
        self.replays = 0

    def reset(self):
        """Drop the captured graph (before a re-capture with new shapes, or when the loop goes back to eager steps): its
        gradient-scale blocks are no longer watched by `Runtime.grads_finite()`."""
        self._gs_blocks = []
        self.graph = None
        self._grads = []

    def __call__(self):
        self.graph.replay()
        for p, g in self._grads:
            p.grad = g
        self.replays += 1
        return self.out

"""Data-parallel wrapper: one process per GPU, gradient mean over RCCL (torch.distributed
backend "nccl" on ROCm) — the replacement for the DistributedDataParallel wrap of
recognition/time_interval_machine/models/build.py:58-63.

Windows are independent, so the only exchange per step is the gradient all-reduce.  The
encoder backward produces one flat fp32 bucket per layer (plus a heads and a front-end
bucket); each bucket is all-reduced on a side stream the moment its layer's backward has
been enqueued, so the collective of layer l overlaps the backward of layers l-1..0.
The remaining small parameters (time MLP, DRLoc MLP: ~1.6 M) are reduced as one flat
buffer when the backward finishes.
"""
import torch
import torch.distributed as dist
from torch import nn


class DataParallel(nn.Module):
    def __init__(self, module, process_group=None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._comm = None
        self._pending = []
        self._callback_queued = False
        rt = module.rt
        rt.bucket_hook = self._on_bucket
        rt.finish_hook = self._on_encoder_done
        self._small = [p for n, p in module.named_parameters()
                       if n.startswith("time_mlp.") or n.startswith("drloc_mlp.")]
        self._hook_handles = []
        if self.world > 1:
            for p in self._small:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._on_small_grad))

    # ---- encoder buckets: asynchronous, overlapped with the rest of the backward --------------------
    def _comm_stream(self, dev):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        return self._comm

    def _on_bucket(self, name, flat, ready=None):
        if self.world == 1:
            return
        if flat.is_cuda:
            comm = self._comm_stream(flat.device)
            comm.wait_stream(torch.cuda.current_stream())
            if ready is not None:
                comm.wait_event(ready)  # weight gradients written on the side stream
            with torch.cuda.stream(comm):
                flat.div_(self.world)
                dist.all_reduce(flat, group=self.pg)
            flat.record_stream(comm)
        else:  # gloo / CPU tests of the bucket logic
            flat.div_(self.world)
            dist.all_reduce(flat, group=self.pg)

    def _on_encoder_done(self):
        if self.world > 1 and self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)

    # ---- the few parameters outside the encoder Function -------------------------------------------
    def _on_small_grad(self, p):
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce_small)

    def _reduce_small(self):
        self._callback_queued = False
        ps = [p for p in self._small if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        flat.div_(self.world)
        dist.all_reduce(flat, group=self.pg)
        off = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def allreduce_buckets_reference(bucket_tensors, world, group=None):
    """The bucket arithmetic on its own (used by the gloo tests): mean over ranks, in place."""
    for t in bucket_tensors:
        t.div_(world)
        dist.all_reduce(t, group=group)

"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU node, "gloo" in the CPU tests).

The ray-march path shards by rays: rays (and whole frames) are independent, so rendering needs no
data-path collective at all.  The only exchange in the path is the gradient of the replicated MLP
parameters in a training step (one flat fp32 all-reduce), provided here as
`allreduce_mean_`; everything else is index arithmetic.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_rank():
    return (dist.get_world_size(), dist.get_rank()) if is_dist() else (1, 0)


def shard_frames(world: int, rank: int, frame0: int = 0) -> int:
    """Frame (camera pose) index rendered by `rank` in one step: frames are dealt round-robin."""
    return frame0 + rank


def shard_frame_indices(n_frames: int, world: int, rank: int):
    """Frames of a data set held by `rank` in data-parallel training: rank, rank + world, ... (shard by image).
    The union over ranks is range(n_frames), every frame exactly once; counts differ by at most one."""
    return list(range(rank, n_frames, world))


def shard_range(n: int, world: int, rank: int):
    """Contiguous [begin, end) slice of n rays for `rank`; sizes differ by at most one and the
    concatenation over ranks is exactly range(n) (ragged n is fine)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_rays(tensors, world: int = None, rank: int = None):
    """Slice every per-ray tensor of a pipeline input list along dim 0 for this rank."""
    if world is None:
        world, rank = world_rank()
    n = tensors[0].shape[0]
    b, e = shard_range(n, world, rank)
    return [t[b:e] for t in tensors]


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves host memory: a CUDA tensor is staged through the host for the collective (the CPU test backend, and
    the dry runs of the multi-rank path with several ranks sharing one GPU, where RCCL cannot form a group)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of `t` from rank `src`."""
    if is_dist():
        if _host_staged(t):
            h = t.detach().cpu()
            dist.broadcast(h, src)
            t.detach().copy_(h)
        else:
            dist.broadcast(t.detach(), src)
    return t


def barrier(device=None):
    if is_dist():
        if device is not None and device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not is_dist():
        return float(value)
    on_host = device is None or dist.get_backend() == "gloo"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value: float, device=None) -> float:
    if not is_dist():
        return float(value)
    on_host = device is None or device.type != "cuda" or dist.get_backend() == "gloo"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather ragged per-rank row blocks (the inverse of shard_rays) - used to assemble a frame
    rendered cooperatively by several ranks (196 KiB of RGB per 128x128 frame)."""
    if not is_dist():
        return local
    world, rank = world_rank()
    sizes = [shard_range(n_total, world, r) for r in range(world)]
    maxn = max(e - b for b, e in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if _host_staged(pad):
        pad = pad.cpu()
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: e - b] for o, (b, e) in zip(outs, sizes)], 0).to(local.device)


def allreduce_mean_(flat_grad: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks of the flat gradient buffer (1 220 872 fp32 = 4.88 MB for nerf): the
    one collective of a data-parallel training step.  A single bucket: at this size a ring over xGMI
    is latency-bound, so splitting it only adds launches."""
    if is_dist():
        if _host_staged(flat_grad):
            h = flat_grad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat_grad.copy_(h)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad.div_(dist.get_world_size())
    return flat_grad


class RcclComm:
    """An RCCL communicator of this process group, created through the library (include/smplnerf.h "8(e)": snerf_comm_unique_id on
    rank 0, the 128 id bytes handed to the other ranks through torch.distributed - any backend - and snerf_comm_init_rank on every
    rank with its device current).  `handle` is the ncclComm_t the one-call data-parallel steps take
    (snerf_nerf_train_step_dp_f32 / snerf_smpl_nerf_train_step_dp_f32): the gradient average runs inside the call, on the
    compute stream, between the backward and the optimiser - no torch.distributed call, no separate division, in the step."""

    def __init__(self, device: torch.device):
        import ctypes
        from . import _lib
        if device.type != "cuda":
            raise RuntimeError("RcclComm: needs this rank's GPU")
        self._lib = _lib
        self.device = device
        lib = _lib.load()
        world, rank = world_rank()
        ident = (ctypes.c_ubyte * 128)()
        if rank == 0:
            _lib.check(lib.snerf_comm_unique_id(ctypes.cast(ident, ctypes.c_void_p)), "snerf_comm_unique_id")
        if is_dist() and world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.snerf_comm_init_rank(ctypes.cast(ident, ctypes.c_void_p), world, rank, ctypes.byref(handle)), "snerf_comm_init_rank")
        self.handle = handle
        w, r = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(lib.snerf_comm_info(handle, ctypes.byref(w), ctypes.byref(r)), "snerf_comm_info")
        self.world, self.rank = int(w.value), int(r.value)      # as RCCL reports them

    def allreduce_avg_(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place average over the ranks on the current stream (snerf_comm_allreduce_avg_f32)."""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        with torch.cuda.device(flat.device):
            self._lib.check(self._lib.load().snerf_comm_allreduce_avg_f32(self.handle, flat.data_ptr(), flat.numel(), self._lib.current_stream()),
                            "snerf_comm_allreduce_avg_f32")
        return flat

    def close(self):
        if self.handle is not None and self.handle.value:
            torch.cuda.synchronize(self.device)
            self._lib.load().snerf_comm_destroy(self.handle)
        self.handle = None


def rccl_usable(device) -> bool:
    """The in-library RCCL path applies when every rank has its own GPU (the group runs on the nccl backend, or there is no
    group / a world of one on a GPU); ranks that share a GPU (the gloo dry runs) keep the host-staged torch path."""
    if device is None or device.type != "cuda":
        return False
    return (not is_dist()) or dist.get_backend() == "nccl"

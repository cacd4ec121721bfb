"""Device-buffer plumbing: PyTorch-ROCm tensors are used only as allocations."""
import numpy as np
import torch

from . import _lib


def device():
    if not torch.cuda.is_available():
        raise _lib.AgxError('no ROCm GPU visible: the agx hot path runs only on an MI355X (gfx950); '
                            'there is no CPU fallback')
    idx = torch.cuda.current_device()
    _lib.init_device(idx)
    return torch.device('cuda', idx)


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def r4(n):
    return (int(n) + 3) & ~3


def empty(shape, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=device())


def zeros(shape, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=device())


def to_dev(a, dtype=torch.float32):
    """Host NumPy / torch tensor -> contiguous device tensor of `dtype`."""
    dev = device()
    if isinstance(a, DeviceArray):
        a = a.tensor
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype={
        torch.float32: np.float32, torch.int32: np.int32, torch.float64: np.float64}[dtype])).to(dev)


def padded_matrix(a):
    """[N, G] host/device fp32 matrix -> device tensor [N, r4(G)] (16-B aligned rows)."""
    t = to_dev(a)
    if t.dim() != 2:
        raise ValueError('expected a 2-D matrix, got shape %s' % (tuple(t.shape),))
    n, g = t.shape
    ld = r4(g)
    if ld == g:
        return t, ld
    out = torch.zeros((n, ld), dtype=torch.float32, device=t.device)
    out[:, :g] = t
    return out, ld


def world():
    """(world_size, rank) of the torch.distributed job, (1, 0) outside one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


_native = {}
_comm_streams = {}


def native_comm():
    """libagx's own RCCL communicator (agx_comm_*: collectives queued by the C library on the streams it is
    given, no c10d call per collective) or None.  Under torch.distributed it is used when the job's backend is
    'nccl' (= RCCL) unless AGX_NATIVE_COMM=0, or whatever the backend with AGX_NATIVE_COMM=1; the 128-byte RCCL id
    travels over the process group that launched the job.  Outside a distributed job AGX_DP_FORCE=1 yields a
    ONE-rank communicator (every collective is the identity): the data-parallel code paths on one GPU, for tests."""
    import os
    import ctypes as C
    import torch.distributed as dist
    env = os.environ.get('AGX_NATIVE_COMM')
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if distributed:
        if env == '0' or (env != '1' and dist.get_backend() != 'nccl'):
            return None
        key = (dist.get_world_size(), dist.get_rank())
    elif os.environ.get('AGX_DP_FORCE') == '1':
        key = (1, 0)
    else:
        return None
    if key not in _native:
        lib = _lib.load()
        device()
        buf = (C.c_char * 128)()
        if key[1] == 0:
            _lib.check(lib.agx_comm_unique_id(buf), 'agx_comm_unique_id')
        box = [bytes(buf)]
        if distributed:
            dist.broadcast_object_list(box, src=0)
        comm = C.c_void_p()
        _lib.check(lib.agx_comm_init(C.create_string_buffer(box[0], 128), key[0], key[1], C.byref(comm)), 'agx_comm_init')
        _native[key] = comm
    return _native[key]


def comm_stream():
    """The stream the overlapped data-parallel exchange runs on (one per device, kept)."""
    dev = device()
    st = _comm_streams.get(dev.index)
    if st is None:
        st = _comm_streams[dev.index] = torch.cuda.Stream(device=dev)
    return st


def allreduce_sum(t):
    """Sum all-reduce of a device tensor (RCCL; via the host under a gloo rehearsal)."""
    import torch.distributed as dist
    comm = native_comm() if t.is_cuda and t.dtype in (torch.float32, torch.float64) and t.is_contiguous() else None
    if comm is not None:
        _lib.check(_lib.load().agx_comm_allreduce_sum(comm, stream(), t.data_ptr(), t.numel(), int(t.dtype == torch.float64)),
                   'agx_comm_allreduce_sum')
        return t
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def reduce_scatter_sum(out, inp):
    """out = this rank's 1/R slice of the sum of inp over ranks (RCCL reduce-scatter; via the host under gloo)."""
    import torch.distributed as dist
    comm = native_comm() if inp.is_cuda and inp.dtype == torch.float32 and inp.is_contiguous() and out.is_contiguous() else None
    if comm is not None:
        _lib.check(_lib.load().agx_comm_reduce_scatter_sum(comm, stream(), inp.data_ptr(), out.data_ptr(), out.numel()),
                   'agx_comm_reduce_scatter_sum')
        return out
    if inp.is_cuda and dist.get_backend() == 'gloo':
        ho, hi = out.cpu(), inp.cpu()
        dist.reduce_scatter_tensor(ho, hi)
        out.copy_(ho)
    else:
        dist.reduce_scatter_tensor(out, inp)
    return out


def all_gather(out, inp):
    """out = concatenation of every rank's inp (RCCL all-gather; via the host under gloo)."""
    import torch.distributed as dist
    comm = native_comm() if inp.is_cuda and inp.dtype == torch.float32 and inp.is_contiguous() and out.is_contiguous() else None
    if comm is not None:
        _lib.check(_lib.load().agx_comm_all_gather(comm, stream(), inp.data_ptr(), out.data_ptr(), inp.numel()),
                   'agx_comm_all_gather')
        return out
    if inp.is_cuda and dist.get_backend() == 'gloo':
        ho, hi = out.cpu(), inp.cpu()
        dist.all_gather_into_tensor(ho, hi)
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, inp)
    return out


class DeviceArray:
    """Result living in HBM; converts lazily (the only host sync is the conversion)."""

    def __init__(self, tensor):
        self.tensor = tensor

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def dtype(self):
        return np.dtype(str(self.tensor.dtype).replace('torch.', ''))

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __float__(self):
        return float(self.numpy().reshape(-1)[0])

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return 'DeviceArray(shape=%s)' % (self.shape,)

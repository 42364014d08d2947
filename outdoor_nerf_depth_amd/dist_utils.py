"""Backend-agnostic distributed helpers (RCCL on GPUs, gloo in the CPU tests)."""
import os

import torch

# RCCL's footprint for this path's collective: ONE flat 4.81 MB float32 all-reduce per cascade level and step, overlapped with
# MLP / weight-gradient launches that are sized to whole rounds of one-workgroup-per-CU tiles.  Every CU an RCCL channel holds
# while such a launch runs pushes a tile into an extra round (profiles/r04_rccl_standin.json, r05_rccl_standin_small.json:
# +2 ... 7 % per step for 1-4 held CUs, +6 ... 11 % for 8-32), and 4.81 MB over 7 xGMI links needs no more than a few
# channels: at most 4 (the full weight-gradient launch leaves exactly 4 of the 256 CUs idle: 12 jobs x 21 slices).
# Defaults only (os.environ.setdefault before the communicator is created); a caller's own NCCL_* settings win.
RCCL_ENV_DEFAULTS = {'NCCL_MAX_NCHANNELS': '4', 'NCCL_MIN_NCHANNELS': '2'}


def single_node(world_size=None):
    """True when every rank of the job runs on this host (torchrun exports LOCAL_WORLD_SIZE; the spawn launchers of this
    package are single-node by construction)."""
    world = int(world_size if world_size is not None else os.environ.get('WORLD_SIZE', '1'))
    local = os.environ.get('LOCAL_WORLD_SIZE')
    return local is None or int(local) >= world


def apply_rccl_env_defaults(world_size=None, channels=None):
    """Call before init_process_group('nccl') / RcclComm(): returns the values in effect ({} = RCCL's own defaults).
    The channel cap is tuned for ONE xGMI node (see above) and is process-global -- it also binds every other RCCL communicator
    of the process -- so it is applied only when the job fits one node; `channels` (the CLIs' --rccl_channels): 0 = leave RCCL
    alone, N > 0 = NCCL_MAX_NCHANNELS = N (min(2, N) as the minimum) also across nodes.  A caller's own NCCL_* settings win."""
    if channels is not None and int(channels) == 0:
        return {k: os.environ[k] for k in RCCL_ENV_DEFAULTS if k in os.environ}
    if channels is not None and int(channels) > 0:
        want = {'NCCL_MAX_NCHANNELS': str(int(channels)), 'NCCL_MIN_NCHANNELS': str(min(2, int(channels)))}
    elif single_node(world_size):
        want = RCCL_ENV_DEFAULTS
    else:
        want = {}
    for k, v in want.items():
        os.environ.setdefault(k, v)
    return {k: os.environ[k] for k in RCCL_ENV_DEFAULTS if k in os.environ}


def rccl_debug_file_env(tag='nerfpp'):
    """Environment that makes RCCL write its INIT lines to a per-process file (unless the caller configured NCCL_DEBUG
    already): the channel count a communicator actually got is only visible there (no public query)."""
    if 'NCCL_DEBUG' in os.environ:
        return {}
    import tempfile
    return {'NCCL_DEBUG': 'INFO', 'NCCL_DEBUG_SUBSYS': 'INIT',
            'NCCL_DEBUG_FILE': os.path.join(tempfile.gettempdir(), '%s_rccl_%%h_%%p.log' % tag)}


def rccl_channels_in_effect():
    """Parse this process's RCCL debug file (rccl_debug_file_env) for the channel counts of its communicators: a list of
    {'coll': n, 'p2p': m} in creation order, or None when there is no file / no such line (gloo, NCCL_DEBUG set by the caller)."""
    import re
    import socket
    path = os.environ.get('NCCL_DEBUG_FILE')
    if not path:
        return None
    path = path.replace('%h', socket.gethostname()).replace('%p', str(os.getpid()))
    try:
        with open(path) as f:
            text = f.read()
    except OSError:
        return None
    found = []
    for m in re.finditer(r'(\d+) coll channels(?:, (\d+) collnet channels)?(?:, (\d+) nvls channels)?, (\d+) p2p channels', text):
        found.append({'coll': int(m.group(1)), 'p2p': int(m.group(4))})
    return found or None


def allreduce_mean_(flat, world_size, prescaled=False):
    """DDP gradient semantics (ddp_train_nerf.py:323): every rank ends with the average.
    prescaled=True means the caller already multiplied by 1/world_size (the HIP backward does)."""
    if world_size <= 1:
        return flat
    import torch.distributed as dist
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if not prescaled:
        flat.div_(world_size)
    return flat


def rank_seeds(rank):
    """ddp_train_nerf.py:406-408: numpy and torch are both seeded with (rank+1)*777 so every rank
    draws a different frame, different rays and different depth perturbations."""
    return (rank + 1) * 777


def shard_sizes(n_items, world_size, allow_ragged=True):
    """Contiguous split of a frame's rays over ranks (ddp_train_nerf.py:142-143).  The reference
    raises unless n_items % world_size == 0 (:137-139); 375*1242 is not divisible by 4 or 8, so the
    default here gives the remainder to the last rank (documented divergence)."""
    base = n_items // world_size
    if base * world_size != n_items and not allow_ragged:
        raise Exception('Number of pixels in the image is not divisible by the number of GPUs!\n\t'
                        '# pixels: {}\n\t# GPUs: {}'.format(n_items, world_size))
    sizes = [base] * world_size
    sizes[-1] = n_items - base * (world_size - 1)
    return sizes


def gather_ragged(t, sizes, rank, world_size):
    """Gather per-rank [sizes[r], ...] tensors to rank 0 (ddp_train_nerf.py:229-243 with ragged
    shards): pads to the largest shard, all_gather, trims."""
    if world_size <= 1:
        return t
    import torch.distributed as dist
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    if pad.is_cuda and dist.get_backend() == 'gloo':
        pad = pad.cpu()          # gloo (the shared-GPU / CPU test backend) gathers host tensors only; RCCL gathers in place
    outs = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(outs, pad)
    if rank != 0:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)



class RcclComm(object):
    """An RCCL communicator driven through the C ABI (include/nerfpp_hip.h: nerfpp_rccl_* / nerfpp_allreduce_mean) instead of
    torch.distributed: what a host without torch would bind.  One process per GPU; `exchange(id_bytes_or_None) -> id_bytes`
    moves rank 0's 128-byte unique id to every rank (default: torch.distributed.broadcast_object_list over whatever process
    group is initialised, e.g. gloo)."""

    def __init__(self, rank, world_size, exchange=None):
        import ctypes as C
        from . import _lib as L
        self.rank, self.world_size = int(rank), int(world_size)
        lib = L.lib()
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            self._check(lib.nerfpp_rccl_unique_id(buf), 'nerfpp_rccl_unique_id')
        uid = buf.raw if self.rank == 0 else None
        if self.world_size > 1:
            if exchange is None:
                import torch.distributed as dist

                def exchange(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            uid = exchange(uid)
        self._comm = C.c_void_p()
        self._check(lib.nerfpp_rccl_comm_init(C.byref(self._comm), self.world_size, uid, self.rank), 'nerfpp_rccl_comm_init')

    @staticmethod
    def _check(rc, what):
        from . import _lib as L
        if rc != L.OK:
            raise L.NerfppError('%s failed (code %d): %s' % (what, rc, L.lib().nerfpp_comm_last_error().decode('utf-8', 'replace')))

    def allreduce_mean(self, grads, prescaled=True):
        """grads (device float32, contiguous) <- mean over the ranks, in place, on torch's current stream."""
        import ctypes as C
        import torch
        from . import _lib as L
        assert grads.is_cuda and grads.dtype == torch.float32 and grads.is_contiguous()
        self._check(L.lib().nerfpp_allreduce_mean(C.c_void_p(torch.cuda.current_stream().cuda_stream), self._comm,
                                                   C.c_void_p(grads.data_ptr()), grads.numel(), self.world_size, int(bool(prescaled))),
                    'nerfpp_allreduce_mean')
        return grads

    def destroy(self):
        from . import _lib as L
        if getattr(self, '_comm', None) is not None and self._comm.value:
            self._check(L.lib().nerfpp_rccl_comm_destroy(self._comm), 'nerfpp_rccl_comm_destroy')
            self._comm = None

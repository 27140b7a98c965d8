"""One process per GPU: the rank context the two CLIs and bench.py share.

The reference is single-process (build-index.py:17, query-index.py:20); this is the multi-GPU shape the
north star adds: launch N processes with `python -m torch.distributed.run --nproc-per-node N <script>`,
each owning one GPU. The data path uses ONE collective — the all-gather of per-shard top-K lists (RCCL,
`index.ShardedFlatIP`). Everything here is control plane: small host objects (file lists, a query line,
per-batch results for rank 0's store) travel over a gloo group, never over the data-path group.
"""
import os


class Ranks:
    """rank / world / local rank from the launcher's environment; process groups created on demand.

    world == 1 needs no process group at all: every helper degenerates to the identity, so the single-GPU
    CLI runs exactly as before."""

    def __init__(self, device_type="cuda", rank=None, world=None, local=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.local = int(os.environ.get("LOCAL_RANK", str(self.rank))) if local is None else local
        self.device_type = device_type
        self.dist = None
        self.ctl = None          # gloo group: host objects
        self.data = None         # nccl (RCCL) group on GPUs, the same gloo group on CPU
        self._own_pg = False

    @property
    def device(self):
        import torch
        return torch.device(f"cuda:{self.local}") if self.device_type == "cuda" else torch.device("cpu")

    @property
    def leader(self):
        return self.rank == 0

    def init(self):
        """Join the process group(s). RCCL for device tensors (backend "nccl" is RCCL on ROCm), gloo for host
        objects. Rendezvous over 127.0.0.1 unless the launcher said otherwise."""
        if self.world == 1 or self.dist is not None:
            return self
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        self.dist = dist
        if not dist.is_initialized():
            if self.device_type == "cuda":
                torch.cuda.set_device(self.local)
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self._own_pg = True
        self.data = dist.group.WORLD
        self.ctl = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
        return self

    def close(self):
        if self.dist is not None and self._own_pg and self.dist.is_initialized():
            self.dist.barrier(group=self.ctl)
            self.dist.destroy_process_group()
        self.dist = None

    # ---- control plane (host objects) -------------------------------------------------------------
    def bcast(self, obj, src=0):
        if self.world == 1:
            return obj
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src, group=self.ctl)
        return box[0]

    def gather(self, obj, dst=0):
        """List of every rank's object on `dst` (rank order), None elsewhere."""
        if self.world == 1:
            return [obj]
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(obj, out, dst=dst, group=self.ctl)
        return out

    def all_gather_ints(self, vals):
        """Every rank's short list of ints on every rank (rank order): ONE small gloo all-gather. The build loop's per-round
        agreement (progress counts + "a rank was asked to stop") rides on it."""
        vals = [int(v) for v in vals]
        if self.world == 1:
            return [vals]
        import torch
        mine = torch.tensor(vals, dtype=torch.int64)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.ctl)
        return [o.tolist() for o in out]

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.ctl)

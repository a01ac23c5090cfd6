"""The slice of the torch.distributed API haphic_amd/sharded.py uses, on a process group whose backend cannot move device
memory (gloo): every collective stages its tensors through host memory.  NOT the product transport (that is RCCL over xGMI,
backend "nccl"): it exists so that the exact N-rank command — kernels, layouts, order and content of every exchange — can be run
and verified on a box without xGMI, with several ranks sharing ONE GPU, where RCCL refuses duplicate devices
(`bench.py --gpus N --transport host`, tests/test_gpu_multirank.py)."""


class HostStagedCollectives:
    """The slice of the torch.distributed API this module uses, on a process group whose backend cannot move device
    memory (gloo): every collective stages its tensors through host memory.  The data path — kernels, layouts, the
    order and content of every exchange — is the RCCL one; only the wire differs.  Used to run the HIP engine under
    several ranks that share ONE GPU (tests; a box without xGMI), where RCCL refuses duplicate devices."""

    def __init__(self, dist, group=None):
        self._d, self._g = dist, group
        self.ReduceOp = dist.ReduceOp

    def get_world_size(self):
        return self._d.get_world_size(self._g)

    def get_rank(self):
        return self._d.get_rank(self._g)

    def barrier(self):
        self._d.barrier(self._g)

    def all_gather_object(self, out, obj):
        self._d.all_gather_object(out, obj, group=self._g)

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op if op is not None else self.ReduceOp.SUM, group=self._g)
        t.copy_(h)

    def all_gather_into_tensor(self, out, inp):
        h = out.new_empty(out.shape, device='cpu')
        self._d.all_gather_into_tensor(h, inp.cpu().contiguous(), group=self._g)
        out.copy_(h)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        h = out.new_empty(out.shape, device='cpu')
        self._d.all_to_all_single(h, inp.cpu().contiguous(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes,
                                  group=self._g)
        out.copy_(h)



"""Large-message probe of the collectives haphic_amd/sharded.py uses, on a one-rank RCCL group (all a one-GPU box offers): does a
single call with more than 2^31 / 2^32 BYTES come back intact?  Round 2 saw a 2.6 GB all-to-all "come back truncated" and capped
every call at MAX_MESSAGE_BYTES = 1 GiB without finding the cause; this script finds the size at which each collective breaks
(if any), so that the cap can be documented as a limit of the stack below torch.distributed or removed.  One JSON line per case."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29544')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    print(json.dumps({'torch': torch.__version__, 'nccl': list(torch.cuda.nccl.version())}))
    sizes_gb = [float(x) for x in sys.argv[1:]] or [0.5, 0.8, 1.0, 1.0737, 1.08, 1.2, 1.5, 1.9, 2.7, 4.6, 8.7]
    for gb in sizes_gb:
        n = int(gb * 1e9 / 8)
        src = torch.arange(n, dtype=torch.int64, device='cuda:0') * 3 + 1
        for name in ('all_to_all_single', 'all_gather_into_tensor', 'all_reduce_min'):
            out = torch.zeros_like(src)
            try:
                if name == 'all_to_all_single':
                    dist.all_to_all_single(out, src, output_split_sizes=[n], input_split_sizes=[n])
                elif name == 'all_gather_into_tensor':
                    dist.all_gather_into_tensor(out, src)
                else:
                    out.copy_(src)
                    dist.all_reduce(out, op=dist.ReduceOp.MIN)
                torch.cuda.synchronize()
                bad = (out != src)
                n_bad = int(bad.sum().item())
                first = int(torch.nonzero(bad)[0].item()) if n_bad else -1
                rec = {'op': name, 'bytes': n * 8, 'elements': n, 'intact': n_bad == 0, 'bad_elements': n_bad, 'first_bad_element': first,
                       'first_bad_byte': first * 8 if first >= 0 else -1}
            except Exception as e:           # noqa: BLE001 — the probe reports whatever the stack raises
                rec = {'op': name, 'bytes': n * 8, 'elements': n, 'error': str(e)[:300]}
            print(json.dumps(rec), flush=True)
            del out
        del src
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

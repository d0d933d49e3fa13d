"""Helper for test_parallel_gloo.py: one rank of a gloo group started by bench.spawn_ranks."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    fail_rank = int(sys.argv[1]) if len(sys.argv) > 1 else -1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if rank == fail_rank:
        sys.exit(7)   # a rank that never joins: the launcher must report it
    if fail_rank >= 0:
        return        # (no collective in the failing scenario: it would hang the others)
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_ranks": dist.get_world_size(), "sum": t.item(), "world_env": world}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

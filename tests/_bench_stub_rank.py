"""Stand-in rank for tests/test_distributed_gloo.py: drives bench.py's N > 1 control flow (launch_ranks -> env://
rendezvous -> barrier-bracketed timed loop -> one all_gather per step -> MAX all_reduce -> rank 0 prints ONE JSON line)
on the CPU with gloo; the GPU work of a step is replaced by a rank-tagged uint8 tensor.  `--fail-rank R` makes rank R
exit non-zero before the rendezvous (the launcher must propagate that and stop the other ranks)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fail-rank", type=int, default=-1)
    a = ap.parse_args()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    assert world == a.gpus and int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if rank == a.fail_rank:
        sys.exit(7)
    dist.init_process_group("gloo")
    coll = torch.device("cpu")
    seen = []

    def step(i):
        time.sleep(0.01 * (rank + 1))  # ranks finish at different times: the reported time is the slowest rank's
        out = torch.full((2, 3, 4, 4), 10 * rank + 1, dtype=torch.uint8)
        bufs = bench.exchange(out, world, coll)
        seen.append([int(b[0, 0, 0, 0]) for b in bufs])
        return out

    elapsed = bench.run_steps(step, a.steps, a.warmup, world, coll, lambda: None)
    assert len(seen) == a.steps + a.warmup and all(s == [10 * r + 1 for r in range(world)] for s in seen)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "steps": a.steps, "warmup": a.warmup, "elapsed": elapsed,
                          "min_expected": 0.01 * world * a.steps}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

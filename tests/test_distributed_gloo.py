"""N > 1 path on CPU: two gloo ranks shard the image indices, produce rank-tagged stand-in outputs and exchange them
with the single all_gather of free-hunch_amd/pipeline.py; plus the flag parser of the CLI."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from free_hunch_amd.pipeline import gather_images, shard_indices
    mine = shard_indices(total, rank, world)
    local = torch.stack([torch.full((3, 4, 4), (7 * i + 1) % 256, dtype=torch.uint8) for i in mine]) if mine else \
        torch.zeros((0, 3, 4, 4), dtype=torch.uint8)
    out = gather_images(local, mine, total, torch.device("cpu"))
    q.put((rank, mine, out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_all_gather_world2():
    total, world = 7, 2  # uneven: rank 0 gets 4 images, rank 1 gets 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = sorted(i for _r, mine, _o in res for i in mine)
    assert seen == list(range(total))
    for _r, _mine, out in res:
        assert out.shape == (total, 3, 4, 4)
        for i in range(total):
            assert (out[i] == (7 * i + 1) % 256).all()


def test_cli_flag_parser():
    sys.path.insert(0, ROOT)
    from free_hunch_amd.config import load_config
    o = load_config(["--outdir=/tmp/x", "--num_steps=30", "--do_space_updates=false", "--seeds=1,2",
                     "--space_step_update_lower_threshold=1000.0", "--scale_factor=4.0"])
    assert o.num_steps == 30 and o.do_space_updates is False and o.seeds == [1, 2]
    assert o.space_step_update_lower_threshold == 1000.0 and o.scale_factor == 4
    assert o.conditioning_mechanism == "online_covariance"

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
    # the metric partial sums ride in the header of the same (single) all_gather
    part = torch.tensor([float(sum(mine)), float(len(mine)), 0.25 * (rank + 1)], dtype=torch.float64)
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    real_ar = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real_ar(*a, **k))[1]
    out, sums = gather_images(local, mine, total, torch.device("cpu"), partial_sums=part)
    dist.all_gather, dist.all_reduce = real, real_ar
    q.put((rank, mine, out.numpy().copy(), sums.numpy().copy(), calls))
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
    seen = sorted(i for _r, mine, _o, _s, _c in res for i in mine)
    assert seen == list(range(total))
    for _r, _mine, out, sums, calls in res:
        assert out.shape == (total, 3, 4, 4)
        for i in range(total):
            assert (out[i] == (7 * i + 1) % 256).all()
        # partial sums of both ranks, summed: image indices 0..6, image count, and the rank tags 0.25 + 0.5
        assert sums.tolist() == [float(sum(range(total))), float(total), 0.75]
        assert calls == [1], calls  # exactly ONE collective: no second all_gather for the indices, no all_reduce


def test_launcher_caps_cpu_threads_per_rank(monkeypatch):
    """bench.launch_ranks gives every rank its share of the host's cores (OMP / MKL pools) unless the caller chose."""
    sys.path.insert(0, ROOT)
    import bench
    assert 1 <= bench.rank_cpu_threads(8) <= bench.rank_cpu_threads(1) <= 32
    seen = {}

    class P:
        def __init__(self, argv, env):
            seen.setdefault("envs", []).append(env)

        def poll(self):
            return 0

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", P)
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    monkeypatch.delenv("MKL_NUM_THREADS", raising=False)
    assert bench.launch_ranks(2, []) == 0
    for e in seen["envs"]:
        assert e["OMP_NUM_THREADS"] == e["MKL_NUM_THREADS"] == str(bench.rank_cpu_threads(2))
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["MASTER_ADDR"] == "127.0.0.1"
    seen.clear()
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    assert bench.launch_ranks(2, []) == 0
    assert all(e["OMP_NUM_THREADS"] == "3" for e in seen["envs"])


def _run_launcher(extra):
    """`python bench.py --gpus 2`'s launcher (bench.launch_ranks) on the stub rank, in a child interpreter"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(2, ['--gpus', '2'] + %r, script=%r, timeout=120))"
            % (ROOT, extra, os.path.join(ROOT, "tests", "_bench_stub_rank.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launch_world2():
    """bench.py --gpus 2 without torchrun: two child ranks, gloo rendezvous on 127.0.0.1, one all_gather per step inside
    the barrier-bracketed timed loop, MAX over ranks, exactly one JSON line from rank 0, exit code 0."""
    import json
    r = _run_launcher(["--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3
    assert rec["elapsed"] >= rec["min_expected"]  # the slowest rank sets the time


def test_bench_self_launch_propagates_failure():
    r = _run_launcher(["--fail-rank", "1"])
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])


def test_cli_flag_parser():
    sys.path.insert(0, ROOT)
    from free_hunch_amd.config import load_config
    o = load_config(["--outdir=/tmp/x", "--num_steps=30", "--do_space_updates=false", "--seeds=1,2",
                     "--space_step_update_lower_threshold=1000.0", "--scale_factor=4.0"])
    assert o.num_steps == 30 and o.do_space_updates is False and o.seeds == [1, 2]
    assert o.space_step_update_lower_threshold == 1000.0 and o.scale_factor == 4
    # defaults are the reference's config/config.yaml:62-124
    assert o.conditioning_mechanism == "dps" and o.image_base_covariance == "identity" and o.max_batch_size == 2
    assert o.S_min == 0.0 and o.S_max == float("inf") and o.noise_name == "gaussian" and o.subdirs is False


README_FREE_HUNCH = """--conditioning_mechanism=online_covariance --do_space_updates=true
    --use_analytical_score_time_update=true --project_to_diagonal=false --image_base_covariance=dct_diagonal
    --space_step_update_lower_threshold=1000.0 --space_step_update_threshold=5.0 --scale_factor=4.0 --cond_scaling=1
    --S_churn=0 --num_steps=30 --solver=heun --max_batch_size=1 --total_images=10 --save_other_images=true
    --operator_name=gaussian_blur --noise_sigma=0.1 --num_other_images_to_save=5 --pigdm_posthoc_scaling=false
    --clip_x0_mean=false --scale_factor=4 --inpainting_type=random --inpainting_prob_lower=0.6
    --inpainting_prob_upper=0.8 --dataset=imagenet --dataset_path=data/imagenet
    --openai_state_dict_path=models/256x256_diffusion_uncond.pt
    --openai_setup_path=models/256x256_diffusion_uncond_setup.txt --outdir=results/free-hunch-spaceupdate_gaussian_blur"""


def test_cli_parses_readme_free_hunch_command_unchanged():
    """The reference README's "With space updates" command line (README.md:166-196), token for token."""
    sys.path.insert(0, ROOT)
    from free_hunch_amd.config import load_config
    o = load_config(README_FREE_HUNCH.split())
    assert o.conditioning_mechanism == "online_covariance" and o.image_base_covariance == "dct_diagonal"
    assert o.space_step_update_lower_threshold == 1000.0 and o.space_step_update_threshold == 5.0
    assert o.scale_factor == 4 and o.S_churn == 0.0 and o.num_steps == 30 and o.max_batch_size == 1
    assert o.save_other_images is True and o.num_other_images_to_save == 5 and o.clip_x0_mean is False
    assert o.outdir == "results/free-hunch-spaceupdate_gaussian_blur"
    # reference keys the path does not read are accepted; keys outside the schema are kept as strings
    o = load_config(["--outdir=x", "--S_noise=1.003", "--subdirs=true", "--noise_name=gaussian", "--class_idx=3",
                     "--guidance=1", "--some_future_key=abc"])
    assert o.S_noise == 1.003 and o.subdirs is True and o.class_idx == 3 and o.some_future_key == "abc"

#!/usr/bin/env python3
"""Free Hunch conditional generation on MI355X - the CLI of the reference (generate_conditional.py:434-598):

    python generate_conditional.py --outdir=DIR [--key=value ...]
    torchrun --nproc-per-node 8 generate_conditional.py --outdir=DIR ...

Same flags (free-hunch_amd/config.py), same outputs: DIR/images/{idx:06d}_{seed:06d}.png, cond_images/,
forward_images/, results.txt.  Differences by design: images are sharded i -> rank i mod world with no per-image
barrier, each rank runs `max_batch_size` images in lock-step, outputs are exchanged with one all_gather at the end,
RNG is keyed by (seed, image index).  LPIPS needs a network download and is omitted; PSNR and SSIM are computed on device.
`--synthetic_weights=ffhq|imagenet` runs with seeded random weights when no checkpoint is present."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    from free_hunch_amd import unet as hu
    from free_hunch_amd.config import load_config
    from free_hunch_amd.measurements import get_operator
    from free_hunch_amd.pipeline import gather_images, list_images, load_image_u8, metrics_u8, shard_indices
    from free_hunch_amd.precond import iDDPMLinearPrecond
    from free_hunch_amd.sampler import StandardRGBEncoder, conditional_sampler, conditional_sampler_grouped

    o = load_config(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from free_hunch_amd.conditioning_mechanisms import choose_conditioning_mechanism
    choose_conditioning_mechanism(o.conditioning_mechanism)  # raises for unknown / unsupported names up front
    if o.iddpm_preconditioning != "linear":
        raise NotImplementedError("cosine preconditioning is incompatible with the plugin API in the reference too")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    os.makedirs(o.outdir, exist_ok=True)

    if o.synthetic_weights:
        cfg = {"ffhq": hu.FFHQ256, "imagenet": hu.IMAGENET256}[o.synthetic_weights]
        model = hu.UNetModel(cfg, backend=o.unet_backend, dtype=o.unet_dtype)
        model.load_state_dict(hu.seeded_state(cfg, 0))
    else:
        model, cfg = hu.load_model(o.openai_state_dict_path, o.openai_setup_path, backend=o.unet_backend,
                                   dtype=None if o.unet_dtype == "fp32" else o.unet_dtype)
    net = iDDPMLinearPrecond(model.to(device).eval(), cfg.image_size, 3).to(device)
    S = cfg.image_size

    files = list_images(o.dataset_path)[: o.total_images]
    if not files:
        raise SystemExit(f"no images under {o.dataset_path}")
    total = len(files)
    mine = shard_indices(total, rank, world)
    enc = StandardRGBEncoder()
    data_dir = o.dataset_path if os.path.exists(os.path.join(o.dataset_path, "dct_variance.pt")) else \
        os.path.join(ROOT, "free-hunch_amd", "data")
    # every option the reference forwards to its sampler / plugin (generate_conditional.py:121-130, 495)
    kw = dict(conditioning_mechanism=o.conditioning_mechanism, cond_scaling=o.cond_scaling, clip_x0_mean=o.clip_x0_mean,
              pigdm_posthoc_scaling=o.pigdm_posthoc_scaling, max_vector_count=o.max_vector_count, dataset_path=data_dir,
              image_base_covariance=o.image_base_covariance, pca_component_count=o.pca_component_count,
              denoiser_mean_error_threshold=o.denoiser_mean_error_threshold,
              use_analytical_score_time_update=o.use_analytical_score_time_update,
              project_to_diagonal=o.project_to_diagonal, space_step_update_threshold=o.space_step_update_threshold,
              space_step_update_lower_threshold=o.space_step_update_lower_threshold, max_rtol=o.max_rtol,
              do_space_updates=o.do_space_updates, use_analytic_var_at_end=o.use_analytic_var_at_end,
              solver_type=o.solver_type, use_rtol_func=o.use_rtol_func, diffpir_lambda=o.diffpir_lambda)
    loop = dict(num_steps=o.num_steps, sigma_min=o.sigma_min, sigma_max=o.sigma_max, rho=o.rho, solver=o.solver,
                discretization=o.discretization, schedule=o.schedule, scaling=o.scaling)
    churn = dict(S_churn=o.S_churn, S_min=o.S_min, S_max=o.S_max, S_noise=o.S_noise)
    # the lock-step sampler covers the deterministic loop; stochastic churn runs image by image like the reference
    lockstep = o.conditioning_mechanism == "online_covariance" and o.S_churn == 0
    outs, conds, fwds = [], [], []
    # one unit = (image, seed): the reference repeats every image once per seed (generate_conditional.py:378-390)
    units = [(i, sd) for i in mine for sd in o.seeds]
    for s in range(0, len(units), max(1, o.max_batch_size)):
        chunk = units[s: s + max(1, o.max_batch_size)]
        ops, ys, noise, imgs = [], [], [], []
        for b, (i, sd) in enumerate(chunk):
            key = (sd * 1000003 + i) % (1 << 31)  # RNG keyed by (seed, image index): independent of the world size
            np.random.seed(key)
            torch.manual_seed(key)
            op = get_operator(name=o.operator_name, device=device, sigma_s=o.noise_sigma, kernel_size=o.kernel_size,
                              intensity=o.intensity, scale_factor=o.scale_factor, in_shape=(1, 3, S, S),
                              mask_opt={"mask_type": o.inpainting_type, "mask_len_range": (64, 156),
                                        "mask_prob_range": (o.inpainting_prob_lower, o.inpainting_prob_upper)
                                        if o.inpainting_type == "random" else (0.1, 0.3), "image_size": S})
            op.ctx_slot = b
            img = load_image_u8(files[i], S)
            imgs.append(img)
            ops.append(op)
            ys.append(op.forward(enc.encode(img[None].to(device)), noiseless=False))
            noise.append(torch.randn((1, 3, S, S), generator=torch.Generator().manual_seed(key), dtype=torch.float32))
        if lockstep:
            # two lock-step groups per GPU from 4 images on (the Free Hunch phase of one overlaps the UNet phase of the other:
            # +7 % at batch 8 on MI355X); FH_LOCKSTEP_GROUPS=1 runs the batch as one group
            ng = int(os.environ.get("FH_LOCKSTEP_GROUPS", "2")) if len(ops) >= 4 else 1
            x = conditional_sampler_grouped(net, torch.cat(noise).to(device), ys, ops, groups=ng, **loop, **kw)
        else:  # comparison methods and churned runs go image by image (batch 1, as in the reference)
            gens = [torch.Generator(device).manual_seed((sd * 1000003 + i) % (1 << 31)) for i, sd in chunk]
            x = torch.cat([conditional_sampler(
                net, noise[b].to(device), None, None, **loop, **churn, measurement=ys[b], operator=ops[b],
                randn_like=lambda t, g=gens[b]: torch.randn(t.shape, generator=g, dtype=t.dtype, device=t.device),
                **kw)[0].detach() for b in range(len(ops))])
        outs.append(enc.decode(x))
        conds.append(torch.stack(imgs).to(device))
        fwds += [enc.decode(y) for y in ys]
        print(f"[rank {rank}] (image, seed) {chunk} done", flush=True)
    local_out = torch.cat(outs) if outs else torch.zeros((0, 3, S, S), dtype=torch.uint8, device=device)
    local_cond = torch.cat(conds) if conds else torch.zeros((0, 3, S, S), dtype=torch.uint8, device=device)
    ns = len(o.seeds)
    unit_ids = [i * ns + o.seeds.index(sd) for i, sd in units]  # global position of (image, seed)
    # metrics as in generate_conditional.py:539-569: per image on the device (fh_metrics_u8); the partial sums travel in the
    # header of the ONE all_gather that collects the images (the reference: a barrier per image + three all_reduces)
    sums = torch.zeros(3, dtype=torch.float64, device=device)
    if local_out.shape[0]:
        ps, ss = metrics_u8(local_out, local_cond)
        sums = torch.stack([ps.sum(), ss.sum(), torch.tensor(float(local_out.shape[0]), dtype=torch.float64, device=device)])
    all_out, sums = gather_images(local_out, unit_ids, total * ns, device, partial_sums=sums)  # the single exchange of the run
    psnr_mean, ssim_mean = float(sums[0] / sums[2]), float(sums[1] / sums[2])
    name = lambda u: f"{u // ns:06d}_{o.seeds[u % ns]:06d}.png"
    import PIL.Image
    for sub in ("images", "cond_images", "forward_images"):
        os.makedirs(os.path.join(o.outdir, sub), exist_ok=True)
    if rank == 0:
        for u in range(total * ns):
            PIL.Image.fromarray(all_out[u].permute(1, 2, 0).cpu().numpy(), "RGB").save(
                os.path.join(o.outdir, "images", name(u)))
            if u % ns == 0:  # the ground-truth images are not exchanged: rank 0 reads them from the dataset itself
                PIL.Image.fromarray(load_image_u8(files[u // ns], S).permute(1, 2, 0).numpy(), "RGB").save(
                    os.path.join(o.outdir, "cond_images", name(u)))
        with open(os.path.join(o.outdir, "results.txt"), "w") as f:
            f.write(f"PSNR: {psnr_mean:.4f}\nSSIM: {ssim_mean:.4f}\nimages: {total * ns}\n")
        print(f"PSNR {psnr_mean:.3f} dB, SSIM {ssim_mean:.4f} over {total * ns} images -> {o.outdir}", flush=True)
    for j, u in enumerate(unit_ids):  # forward (measurement) images are written by the owning rank
        if fwds[j].shape[-1] == S:
            PIL.Image.fromarray(fwds[j][0].permute(1, 2, 0).cpu().numpy(), "RGB").save(
                os.path.join(o.outdir, "forward_images", name(u)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

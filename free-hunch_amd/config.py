"""Flag system of the CLI (reference: config_utils.py:72-114 + config/config.yaml): `--outdir=DIR` plus any number of
`--key=value` overrides, coerced by the schema below.

Schema and defaults are the reference's `config/config.yaml:1-124`, key for key (so a bare run selects `dps` with the
`identity` base covariance and `max_batch_size=2`, exactly like the reference); keys the Free Hunch path does not read
(`net`, `gnet`, `subdirs`, `class_idx`, `guidance`, `ref_stats_name`, `save_videos`, ...) are accepted and carried
along.  A key outside the schema is kept as the string it was given, as `validate_and_convert` does (config_utils.py:
66-68).  Keys marked NEW are this build's additions."""
from __future__ import annotations

import sys
from types import SimpleNamespace

# name -> (type, default)      types: str / int / float / bool / "ints" (List[int])
SCHEMA = {
    "net": (str, None), "gnet": (str, None), "outdir": (str, None), "subdirs": (bool, False), "seeds": ("ints", [0]),
    "class_idx": (int, None), "total_images": (int, 10), "max_batch_size": (int, 2), "device": (str, "cuda"),
    "num_steps": (int, 50), "sigma_min": (float, 0.002), "sigma_max": (float, 80.0), "rho": (float, 7.0),
    "guidance": (float, None), "S_churn": (float, 0.0), "S_min": (float, 0.0), "S_max": (float, float("inf")),
    "S_noise": (float, 1.0), "solver": (str, "heun"), "discretization": (str, "edm"), "schedule": (str, "linear"),
    "scaling": (str, "none"), "architecture": (str, "openai"),
    "openai_state_dict_path": (str, "models/256x256_diffusion_uncond.pt"),
    "openai_setup_path": (str, "models/256x256_diffusion_uncond_setup.txt"),
    "iddpm_preconditioning": (str, "linear"), "conditional": (bool, True),
    "dataset_name": (str, "training.dataset.ImageFolderDataset"), "dataset": (str, "imagenet"),
    "data_subset": (str, "val"), "dataset_path": (str, "data/imagenet/"), "ref_stats_name": (str, "fid_ref.pkl"),
    "operator_name": (str, "gaussian_blur"), "kernel_size": (int, 61), "intensity": (float, 1.0),
    "noise_name": (str, "gaussian"), "noise_sigma": (float, 0.1), "cond_scaling": (float, 1.0),
    "save_videos": (bool, False), "conditioning_mechanism": (str, "dps"), "clip_x0_mean": (bool, False),
    "pigdm_posthoc_scaling": (bool, False), "max_vector_count": (int, 100000),
    "image_base_covariance": (str, "identity"), "pca_component_count": (int, 10),
    "denoiser_mean_error_threshold": (float, 0.2), "use_analytical_score_time_update": (bool, True),
    "project_to_diagonal": (bool, False), "space_step_update_threshold": (float, 10.0),
    "space_step_update_lower_threshold": (float, 1.0), "scale_factor": (int, 2), "do_space_updates": (bool, True),
    "num_other_images_to_save": (int, 200), "max_rtol": (float, 1.0), "use_analytic_var_at_end": (bool, False),
    "inpainting_type": (str, "random"), "inpainting_prob_lower": (float, 0.1), "inpainting_prob_upper": (float, 0.3),
    "solver_type": (str, "customcuda"), "use_rtol_func": (bool, False), "diffpir_lambda": (float, 10.0),
    "use_ddnm_kernel_params": (bool, False), "save_other_images": (bool, False),
    # NEW: seeded random weights of that architecture when no checkpoint is present ("ffhq" | "imagenet")
    "synthetic_weights": (str, ""),
    # NEW: "hip" (libfh_hip.so kernels) | "torch" (PyTorch-ROCm ops, for A/B runs)
    "unet_backend": (str, "hip"),
    # NEW: "fp32" | "bf16" | "fp16" - reduced-precision UNet torso (the reference's use_fp16 flag lives in the
    # checkpoint's setup file, training/openai_fp16_util.py:15-32); a non-parity speed mode
    "unet_dtype": (str, "fp32"),
}


def _coerce(kind, text):
    if kind is bool:
        return text.strip().lower() in ("true", "yes", "1", "on")
    if kind == "ints":
        return [int(v.strip()) for v in text.split(",") if v.strip() != ""]
    if kind is int:
        return int(float(text))  # the README passes --scale_factor=4.0
    return kind(text)


def load_config(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    cfg = {k: d for k, (_t, d) in SCHEMA.items()}
    for arg in argv:
        if not arg.startswith("--") or "=" not in arg:
            raise SystemExit(f"expected --key=value, got '{arg}'")
        key, value = arg[2:].split("=", 1)
        cfg[key] = _coerce(SCHEMA[key][0], value) if key in SCHEMA else value
    if cfg["outdir"] is None:
        raise SystemExit("--outdir=DIR is required")
    return SimpleNamespace(**cfg)

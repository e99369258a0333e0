"""Flag system of the CLI (reference: config_utils.py:72-114 + config/config.yaml): `--outdir=DIR` plus any number of
`--key=value` overrides, coerced by the schema below.  Only the keys the Free Hunch path reads are kept."""
from __future__ import annotations

import sys
from types import SimpleNamespace

# name -> (type, default)
SCHEMA = {
    "outdir": (str, None), "seeds": ("ints", [0]), "total_images": (int, 10), "max_batch_size": (int, 8),
    "device": (str, "cuda"), "num_steps": (int, 50), "sigma_min": (float, 0.002), "sigma_max": (float, 80.0),
    "rho": (float, 7.0), "S_churn": (float, 0.0), "solver": (str, "heun"), "discretization": (str, "edm"),
    "schedule": (str, "linear"), "scaling": (str, "none"), "architecture": (str, "openai"),
    "openai_state_dict_path": (str, "models/256x256_diffusion_uncond.pt"),
    "openai_setup_path": (str, "models/256x256_diffusion_uncond_setup.txt"),
    "synthetic_weights": (str, ""),  # "ffhq" | "imagenet": seeded random weights of that architecture (no checkpoint)
    "iddpm_preconditioning": (str, "linear"), "dataset": (str, "imagenet"), "dataset_path": (str, "data/imagenet/"),
    "operator_name": (str, "gaussian_blur"), "kernel_size": (int, 61), "intensity": (float, 1.0),
    "noise_sigma": (float, 0.1), "cond_scaling": (float, 1.0), "conditioning_mechanism": (str, "online_covariance"),
    "clip_x0_mean": (bool, False), "pigdm_posthoc_scaling": (bool, False), "max_vector_count": (int, 100000),
    "image_base_covariance": (str, "dct_diagonal"), "pca_component_count": (int, 10),
    "denoiser_mean_error_threshold": (float, 0.2), "use_analytical_score_time_update": (bool, True),
    "project_to_diagonal": (bool, False), "space_step_update_threshold": (float, 10.0),
    "space_step_update_lower_threshold": (float, 1.0), "scale_factor": (int, 2), "do_space_updates": (bool, True),
    "num_other_images_to_save": (int, 200), "max_rtol": (float, 1.0), "use_analytic_var_at_end": (bool, False),
    "inpainting_type": (str, "random"), "inpainting_prob_lower": (float, 0.1), "inpainting_prob_upper": (float, 0.3),
    "solver_type": (str, "customcuda"), "use_rtol_func": (bool, False), "diffpir_lambda": (float, 10.0),
    "save_other_images": (bool, False), "unet_backend": (str, "hip"),
}


def _coerce(kind, text):
    if kind is bool:
        return text.strip().lower() in ("true", "yes", "1", "on")
    if kind == "ints":
        return [int(v) for v in text.split(",") if v != ""]
    if kind is int:
        return int(float(text))
    return kind(text)


def load_config(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    cfg = {k: d for k, (_t, d) in SCHEMA.items()}
    for arg in argv:
        if not arg.startswith("--") or "=" not in arg:
            raise SystemExit(f"expected --key=value, got '{arg}'")
        key, value = arg[2:].split("=", 1)
        if key not in SCHEMA:
            raise SystemExit(f"unknown option --{key}")
        cfg[key] = _coerce(SCHEMA[key][0], value)
    if cfg["outdir"] is None:
        raise SystemExit("--outdir=DIR is required")
    return SimpleNamespace(**cfg)

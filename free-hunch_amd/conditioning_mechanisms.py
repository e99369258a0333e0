"""Free Hunch conditioning plugin on MI355X - drop-in for the `online_covariance` entry of the reference's
registry (conditioning_utils/conditioning_mechanisms.py:16-50, 190-294) with its three `customcuda` solvers
(:384-419, :489-527, :641-675) and `rtol_func` (:307-323).

Constructor and call signature are the reference's:
    cls(cond_scaling, forward_operator, clip_x0_mean, init_denoiser_variance, init_noise_variance, data_dim,
        pigdm_posthoc_scaling=False, **kw)           obj(x_t, net, y, sigma) -> x0_mean_new  (float64)
Differences by design: the covariance state, the trajectory history and every intermediate stay in HBM (the
reference round-trips them through the CPU each call), and the linear solve is one C-ABI call
(`fh_cg_solve`) that keeps CG scalars on the device.  The comparison plugins of the reference's registry (DPS, PiGDM, both
schedules, DiffPIR, Peng-analytic, Peng-convert, TMPD) and its `solver_type` variants (customscipy, scipy) run on the same
operator / UNet-VJP / CG kernels; only DDNM (a separate sampler) is not here.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from warnings import warn

import numpy as np
import torch
from torch.autograd import grad

from . import _lib
from .covariance import CovarianceHessianBFGS, CovarianceHessianBFGSDCT, ScalarCovariance, _load_cached

F64 = torch.float64
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def choose_conditioning_mechanism(name):
    if name == "online_covariance":
        return BFGSOnlineUpdate
    if name in _BASELINES:
        return _BASELINES[name]
    if name == "ddnm":
        raise ValueError("DDNM conditioning mechanism not implemented in this branch of the codebase")
    raise ValueError(f"Unknown conditioning mechanism: {name}")


class ConditioningMechanism:
    def __init__(self, cond_scaling, forward_operator, clip_x0_mean, init_denoiser_variance=None,
                 init_noise_variance=None, data_dim=None, pigdm_posthoc_scaling=False, **argv):
        self.cond_scaling = cond_scaling
        self.forward_operator = forward_operator
        self.clip_x0_mean = clip_x0_mean

    def __call__(self, x_t, x_0_mean, y, sigma):
        x_0_mean_new = self.x0_mean_update(x_t, x_0_mean, y, sigma)
        if self.clip_x0_mean:
            x_0_mean_new = x_0_mean_new.clip(-1, 1)
        return x_0_mean_new


# ---------------------------------------------------------------------------------------------- solvers
def rtol_func(sigma, rtol_max=1e0, rtol_min=1e-14):
    """CG tolerance schedule, reference :307-323 (the upper clamp at sigma=80 is ineffective there too)."""
    lo, hi = 0.1, 80.0
    sigma = max(min(sigma, hi), max(lo, sigma))
    frac = ((math.log10(sigma) - math.log10(lo)) / (math.log10(hi) - math.log10(lo))) ** 0.1
    return 10 ** (frac * (math.log10(rtol_max) - math.log10(rtol_min)) + math.log10(rtol_min))


_OP_CODE = {"inpainting": 0, "gaussian_blur": 1, "motion_blur": 1, "super_resolution": 2}


def rtol_func_2(sigma, rtol_max=1e0, rtol_min=1e-4):
    """Tolerance schedule of the reference's scipy solvers when `use_rtol_func` is set (TMPD), :325-343."""
    lo, hi = 0.1, 80.0
    sigma = max(min(sigma, hi), max(lo, sigma))
    frac = ((math.log10(sigma) - math.log10(lo)) / (math.log10(hi) - math.log10(lo))) ** 0.05
    return 10 ** (frac * (math.log10(rtol_max) - math.log10(rtol_min)) + math.log10(rtol_min))


def _problem(operator, cov, sigma_y2):
    p = _lib.FhProblem()
    p.op = _OP_CODE[operator.name]
    p.use_dct = int(cov.use_dct)
    p.planes = 3
    p.stride = int(operator.scale_factor) if operator.name == "super_resolution" else 1
    p.d = cov.data_dim
    p.sigma_y2 = sigma_y2
    m = cov.famC.m
    p.m, p.ldm = m, cov.C.M_dev.shape[1]
    p.D, p.r, p.B, p.M = (cov.C.D.data_ptr(), cov.C.r.data_ptr(), cov.famC.B.data_ptr(), cov.C.M_dev.data_ptr())
    keep = [cov.C.D, cov.C.r, cov.famC.B, cov.C.M_dev]
    if operator.name == "inpainting":
        mask = operator.mask.to(device=cov.device, dtype=F64).contiguous()
        p.mask = mask.data_ptr()
        keep.append(mask)
    else:
        t = operator.taps
        if t.sep is not None and operator.name != "super_resolution":
            t, t2 = t.sep
            p.ntaps2, p.halo2 = t2.n, t2.halo
            p.tap2_dy, p.tap2_dx, p.tap2_w = t2.dy.data_ptr(), t2.dx.data_ptr(), t2.w.data_ptr()
        p.ntaps, p.halo = t.n, t.halo
        p.tap_dy, p.tap_dx, p.tap_w = t.dy.data_ptr(), t.dx.data_ptr(), t.w.data_ptr()
        fold = operator.folded_bases() if (cov.use_dct and p.op == 1) else None
        if fold is not None:  # separable blur absorbed by the DCT passes of A C A^T (measurements.folded_dct_blur_bases)
            mats, sym = fold
            p.fold_fwd_w, p.fold_fwd_h, p.fold_inv_w, p.fold_inv_h = (f.data_ptr() for f in mats)
            p.fold_sym = int(sym)
            keep.extend(mats)
    return p, keep


def _sigma_y2(operator):
    """`sigma_s.clip(min=0.001)**2` evaluated in float32 like the reference (:386, :491), SR also clips at 1e-2 (:642)."""
    s = operator.sigma_s.detach().cpu().float().clip(min=0.001)
    if operator.name == "super_resolution":
        s = s.clip(min=1e-2)
    return float((s ** 2).item())


def solve_customcuda(operator, y, x0_mean, covariance_model, max_rtol, sigma_t, info_out=None, rtol=None, scipy_cg=False,
                     maxiter=None):
    """mat = A^T (A C A^T + sigma_y^2 I)^-1 (y - A x0_mean), float64, on the device.  `scipy_cg`: iterate like
    scipy.sparse.linalg.cg as the reference's scipy solver variants call it (x0 = 0, initial residual tested first,
    maxiter 1000) instead of like its own cg() (x0 = b, maxiter 5000)."""
    cov = covariance_model
    ctx = cov.ctx
    dev = cov.device
    name = operator.name
    if name not in _OP_CODE:
        raise ValueError("Invalid operator name. Please choose 'gaussian_blur', 'super_resolution', "
                         "'motion_blur', or 'inpainting'.")
    prob, keep = _problem(operator, cov, _sigma_y2(operator))
    prob.cg_scipy = int(bool(scipy_cg))
    y64 = y.detach().to(device=dev, dtype=F64).contiguous()
    x64 = x0_mean.detach().to(device=dev, dtype=F64).contiguous()
    # b = y - A x0_mean
    if name == "inpainting":
        mask = keep[-1].view_as(x64)
        b = ctx.axpby(1.0, (mask * y64).contiguous(), -1.0, (mask * x64).contiguous(), torch.empty_like(x64))
    else:
        ax = operator._conv(x64, stride=prob.stride)
        b = ctx.axpby(1.0, y64, -1.0, ax, ax)
    sol = torch.empty_like(b)
    info = _lib.FhCgInfo()
    rtol = rtol_func(sigma_t, max_rtol) if rtol is None else rtol
    if maxiter is None:  # the reference's caps (:401, :512, :660 / scipy: 1000)
        maxiter = 1000 if scipy_cg else 5000
    _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b.data_ptr(), sol.data_ptr(), rtol, 0.0, maxiter,
                                   C.byref(info), _lib.stream()), "fh_cg_solve")
    if info.niter == (1000 if scipy_cg else (5000 if name == "inpainting" else 2000)):  # the reference's (inconsistent) guards
        warn("CG not converge.")
    if info_out is not None:
        info_out.append({"niter": info.niter, "optimal": bool(info.optimal), "residual_norm": info.residual_norm,
                         "rtol": rtol})
    solve_customcuda.last_solution = sol  # the measurement-space CG solution u (mat = A^T u); read by the parity tests
    if name == "inpainting":
        return sol
    return operator._conv(sol, stride=prob.stride, adjoint=True)


solve_customcuda.last_solution = None


def solve_customcuda_batched(operators, ys, x0_means, covariance_models, max_rtol, sigma_t, infos_out=None,
                             exclusive=False):
    """The same solve for B independent images in ONE kernel sequence (`fh_cg_solve_batched`): all images share the
    operator type / taps / noise level and the number of factor columns (they advance in lock-step), each has its own
    covariance state, measurement and iteration count.  Returns mat [B,3,S,S] float64."""
    B = len(operators)
    op0, cov0 = operators[0], covariance_models[0]
    name = op0.name
    if name not in _OP_CODE:
        raise ValueError("Invalid operator name. Please choose 'gaussian_blur', 'super_resolution', "
                         "'motion_blur', or 'inpainting'.")
    dev, S = cov0.device, cov0.S
    # scratch sized for the whole batch; one context per lock-step GROUP (keyed by the group's first image slot), so
    # that equal-sized groups running concurrently from different host threads never share CG vectors or graph caches
    ctx = _lib.Context.get(S, 3 * B, 0, slot=5000 + 64 * int(getattr(op0, "ctx_slot", 0)) + B)
    # `exclusive`: nothing else synchronises across workgroups on this GPU while the solve runs (the lock-step sampler
    # has joined its per-image streams) -> the covariance apply inside CG reads each factor base once, not twice
    ctx.set_exclusive(exclusive)
    prob, keep = _problem(op0, cov0, _sigma_y2(op0))
    per = _lib.FhBatch()
    per.nimg = B
    m = cov0.famC.m
    for b, (op, cov) in enumerate(zip(operators, covariance_models)):
        assert op.name == name and cov.famC.m == m and cov.C.M_dev.shape[1] == cov0.C.M_dev.shape[1], \
            "lock-step images must share operator type and factor count"
        per.D[b], per.r[b], per.B[b], per.M[b] = (cov.C.D.data_ptr(), cov.C.r.data_ptr(), cov.famC.B.data_ptr(),
                                                  cov.C.M_dev.data_ptr())
        if name == "inpainting":
            mk = op.mask.to(device=dev, dtype=F64).contiguous()
            keep.append(mk)
            per.mask[b] = mk.data_ptr()
    y64 = torch.cat([y.detach().to(device=dev, dtype=F64) for y in ys], 0).contiguous()
    x64 = torch.cat([x.detach().to(device=dev, dtype=F64) for x in x0_means], 0).contiguous()
    planes = 3 * B

    def conv(v, adjoint):
        """op0's blur on the whole batch (planes = 3B)"""
        so = S if (adjoint or prob.stride == 1) else S // prob.stride
        out = torch.empty(B, 3, so, so, dtype=F64, device=dev)
        t = op0.taps
        if t.sep is not None and prob.stride == 1:
            first, second = t.sep if not adjoint else t.sep[::-1]
            tmp = torch.empty_like(out)
            ctx.conv(v, tmp, first, planes, 1, adjoint)
            ctx.conv(tmp, out, second, planes, 1, adjoint)
        else:
            ctx.conv(v, out, t, planes, prob.stride, adjoint)
        return out

    if name == "inpainting":
        mask = torch.cat([k.view(1, 3, S, S) for k in keep[-B:]], 0)
        b_vec = ctx.axpby(1.0, (mask * y64).contiguous(), -1.0, (mask * x64).contiguous(), torch.empty_like(x64))
    else:
        ax = conv(x64, False)
        b_vec = ctx.axpby(1.0, y64, -1.0, ax, ax)
    sol = torch.empty_like(b_vec)
    rtol = rtol_func(sigma_t, max_rtol)
    rtols = (C.c_double * B)(*([rtol] * B))
    infos = (_lib.FhCgInfo * B)()
    _lib.check(ctx.lib.fh_cg_solve_batched(ctx.h, C.byref(prob), C.byref(per), b_vec.data_ptr(), sol.data_ptr(), rtols,
                                           0.0, 5000, infos, _lib.stream()), "fh_cg_solve_batched")
    for b in range(B):
        if infos[b].niter == (5000 if name == "inpainting" else 2000):
            warn("CG not converge.")
        if infos_out is not None:
            infos_out.append({"niter": infos[b].niter, "optimal": bool(infos[b].optimal),
                              "residual_norm": infos[b].residual_norm, "rtol": rtol})
    return sol if name == "inpainting" else conv(sol, True)


def choose_solver(operator_name, operator, y, x0_mean, theta0_var=None, covariance_model=None, method="customcuda",
                  max_rtol=1, ortho_tf=None, sigma_t=None, use_rtol_func=False, info_out=None):
    if operator_name not in _OP_CODE:
        raise ValueError("Invalid operator name. Please choose 'gaussian_blur', 'super_resolution', "
                         "'motion_blur', or 'inpainting'.")
    if method == "customcuda":
        return solve_customcuda(operator, y, x0_mean, covariance_model, max_rtol, sigma_t, info_out)
    # The reference's two scipy variants run `scipy.sparse.linalg.cg` on the CPU in float32 with tol = 1e-4 (or rtol_func_2
    # when use_rtol_func) and maxiter 1000 (:360-381, :420-447, :449-484, :529-560, :602-639, :677-706).  Same linear systems
    # on the device CG here in its scipy mode (fh_problem.cg_scipy: x0 = 0, initial residual tested first, no pAp test), in
    # float64 where the reference's LinearOperator is float32.
    tol = 1e-4 if (sigma_t is None or not use_rtol_func) else rtol_func_2(float(sigma_t))
    if method == "customscipy":  # the Free Hunch covariance with scipy's tolerance rule
        return solve_customcuda(operator, y, x0_mean, covariance_model, max_rtol, sigma_t, info_out, rtol=tol, scipy_cg=True)
    if method == "scipy":        # scalar or per-pixel variance theta0_var in image space (no DCT)
        data_dim = int(np.prod(operator.in_shape[1:]))
        return _variance_mat(operator, y, x0_mean, theta0_var, data_dim, tol, info_out)
    raise ValueError(f"unknown solver_type '{method}' (customcuda, customscipy, scipy)")


# ---------------------------------------------------------------------------------------------- the plugin
class BFGSOnlineUpdate(ConditioningMechanism):
    def __init__(self, cond_scaling, forward_operator, clip_x0_mean, init_denoiser_variance, init_noise_variance,
                 data_dim, pigdm_posthoc_scaling=False, **argv):
        super().__init__(cond_scaling, forward_operator, clip_x0_mean)
        device = getattr(forward_operator, "device", torch.device("cuda"))
        self.solver_type, self.max_rtol = argv["solver_type"], argv["max_rtol"]
        self.project_to_diagonal = argv["project_to_diagonal"]
        s0 = float(init_noise_variance)
        base, common = argv["image_base_covariance"], dict(max_vector_count=argv["max_vector_count"],
                                                           project_to_diagonal=self.project_to_diagonal,
                                                           device=device,
                                                           ctx_slot=getattr(forward_operator, "ctx_slot", 0))
        if base == "identity":
            self.covariance_model = CovarianceHessianBFGS(init_denoiser_variance, s0, data_dim, **common)
        elif base in ("dct_diagonal", "dct_diagonal_noinfo"):
            self.covariance_model = CovarianceHessianBFGSDCT(argv["data_dir"], s0, data_dim,
                                                             use_precalculated_info=(base == "dct_diagonal"), **common)
        elif base == "pca_dct_diagonal":
            raise NotImplementedError("pca_dct_diagonal: the reference imports a class that does not exist "
                                      "(conditioning_mechanisms.py:188)")
        else:
            raise ValueError(f"unknown image_base_covariance: {base}")
        self.do_space_updates = argv["do_space_updates"]
        self.init_denoiser_variance, self.init_noise_variance, self.data_dim = \
            init_denoiser_variance, init_noise_variance, data_dim
        self.sigmas, self.xs, self.denoiser_means = [], [], []
        self.denoiser_mean_error_threshold = argv["denoiser_mean_error_threshold"]
        self.use_analytical_score_time_update = argv["use_analytical_score_time_update"]
        self.space_step_update_threshold = argv["space_step_update_threshold"]
        self.space_step_update_lower_threshold = argv["space_step_update_lower_threshold"]
        self.pigdm_posthoc_scaling = pigdm_posthoc_scaling
        self.use_analytic_var_at_end = argv.get("use_analytic_var_at_end", False)
        self.use_rtol_func = argv.get("use_rtol_func", False)
        if self.solver_type == "scipy":
            raise ValueError("solver_type=scipy solves with a scalar / per-pixel variance; online_covariance needs "
                             "customcuda or customscipy")
        # the reference loads this file unconditionally (:225-226)
        self.recon_mse = _load_cached(os.path.join(_DATA, "recon_mse.pt"))
        self.mle_sigma_thres = 0.2
        self.trace = []  # per call: niter, branch, k (not in the reference; used by the parity tests)

    def update_time_step(self, x_t, sigma_t, sigma_tnext, score_t):
        self.covariance_model.update_time_step(x_t, sigma_t, sigma_tnext, score_t)

    def update_space_step(self, denoiser_mean_at_x, denoiser_mean_at_xnext, sigma_t, x, xnext):
        self.covariance_model.update_space_step(denoiser_mean_at_x, denoiser_mean_at_xnext, sigma_t, x, xnext)

    def x0_mean_update(self, x_t, model, y, sigma):
        x_t = x_t.requires_grad_()
        x_0_mean, _ = model(x_t, sigma)
        mat = self.fh_solve(x_t.detach(), x_0_mean.detach(), y, sigma, model)
        p_y_xt_grad = grad((mat.detach() * x_0_mean).sum(), x_t)[0]
        return self.fh_finish(mat, p_y_xt_grad, x_t.detach(), x_0_mean.detach(), sigma)

    # The call is split in two so that a batch of independent images can share ONE UNet forward and ONE UNet
    # input-VJP (sampler.conditional_sampler_batched): fh_solve = covariance updates + CG solve for one image,
    # fh_finish = the 0.2-std branch and the history append.
    def analytic_now(self, sigma):
        return bool(self.use_analytic_var_at_end and float(sigma) < self.mle_sigma_thres)

    def fh_solve(self, x_det, m_det, y, sigma, model=None):
        self.fh_update(x_det, m_det, sigma, model)
        info = []
        if self.analytic_now(sigma):
            # :273-276 - scalar variance from recon_mse.pt; the reference's Fourier closed forms (:357, :454, :608) are
            # A^T (theta A A^T + s^2 I)^-1 (y - A x0) exactly; here the same system goes through CG with C = theta I
            idx = (self.recon_mse["sigmas"].double() - float(sigma)).abs().argmin()
            theta = float(self.recon_mse["mse_list"][idx])
            scal = ScalarCovariance(theta, self.data_dim, m_det.device, self.covariance_model.ctx_slot)
            mat = solve_customcuda(self.forward_operator, y, m_det, scal, 1.0, float(sigma), info, rtol=1e-10)
            self._rec = dict(info[0], analytic=True)
            return mat
        mat = choose_solver(self.forward_operator.name, self.forward_operator, y, m_det,
                            covariance_model=self.covariance_model, method=self.solver_type, max_rtol=self.max_rtol,
                            sigma_t=float(sigma), use_rtol_func=self.use_rtol_func, info_out=info)
        self._rec = dict(info[0])
        return mat

    def fh_update(self, x_det, m_det, sigma, model=None, x_changed=None):
        """the covariance part of a guidance call: time update (sigma changed) and space update (x changed).
        `x_changed`: the result of `not torch.allclose(x, x_prev)` (:250) when the caller has already evaluated it (the
        lock-step sampler does so for the whole batch with one device -> host transfer)."""
        cm = self.covariance_model
        s = float(sigma)
        if self.do_space_updates:
            pred = None
            if len(self.sigmas) != 0 and s != self.sigmas[-1]:
                score_previous = (self.denoiser_means[-1] - self.xs[-1]) / self.sigmas[-1] ** 2
                pred, _ = cm.update_time_step(self.xs[-1], self.sigmas[-1], s, score_previous)
            elif len(self.sigmas) != 0:  # second Heun evaluation at the same sigma: no time update
                pred = self.denoiser_means[-1]
            if len(self.xs) != 0 and (x_changed if x_changed is not None else not torch.allclose(x_det, self.xs[-1])):
                if not self.use_analytical_score_time_update:
                    with torch.no_grad():
                        pred, _ = model(self.xs[-1], sigma)
                if self.space_step_update_lower_threshold < s < self.space_step_update_threshold:
                    cm.update_space_step(pred, m_det, s, self.xs[-1], x_det)
        elif len(self.sigmas) != 0 and s != self.sigmas[-1]:
            score_previous = (self.denoiser_means[-1] - self.xs[-1]) / self.sigmas[-1] ** 2
            cm.update_time_step(self.xs[-1], self.sigmas[-1], s, score_previous, only_covariance=True)

    @staticmethod
    def fh_update_batched(mechs, x_all, m_all, sigma, prev_x_all, prev_m_all, changed, slot=0):
        """`fh_update` of the B plugin instances of a lock-step batch with ONE kernel sequence per covariance update
        (CovarianceHessianBFGS.update_*_step_batched) instead of one per image.  x_all, m_all: this call's x_t and denoiser
        output [B,3,S,S]; prev_x_all, prev_m_all: the previous call's (None on the first call); `changed`: per image
        `not torch.allclose(x, x_prev)` (:250).  Returns False - nothing done - when the batch does not qualify (different
        column counts, factor tracking, a network-score time update, images that did not move): the caller then falls
        back to per-image `fh_update`."""
        a = mechs[0]
        covs = [mm.covariance_model for mm in mechs]
        s = float(sigma)
        if not (a.do_space_updates and a.use_analytical_score_time_update and CovarianceHessianBFGS.can_batch(covs)):
            return False
        if any(mm.sigmas != a.sigmas or mm.do_space_updates != a.do_space_updates
               or mm.space_step_update_threshold != a.space_step_update_threshold
               or mm.space_step_update_lower_threshold != a.space_step_update_lower_threshold for mm in mechs):
            return False
        if len(a.sigmas) == 0:
            return True  # first call: no update
        if prev_x_all is None or prev_m_all is None or changed is None or not all(changed):
            return False
        pred = prev_m_all
        if s != a.sigmas[-1]:
            score_prev = (prev_m_all - prev_x_all) / a.sigmas[-1] ** 2
            pred, _ = CovarianceHessianBFGS.update_time_step_batched(covs, prev_x_all, a.sigmas[-1], s, score_prev, slot=slot)
        if a.space_step_update_lower_threshold < s < a.space_step_update_threshold:
            if not CovarianceHessianBFGS.can_batch(covs):  # (column count after the time update is unchanged; re-check capacity)
                return False
            CovarianceHessianBFGS.update_space_step_batched(covs, pred, m_all, s, prev_x_all, x_all, slot=slot)
        return True

    def fh_branch(self, p_y_xt_grad, sigma, std=None):
        """"cov" when the VJP guidance is judged unreliable (std(vjp * sigma^2) > threshold, :283), else "vjp" """
        if self._rec.get("analytic"):  # :277-278: always the VJP form
            return "vjp"
        if std is None:
            std = (p_y_xt_grad * torch.as_tensor(sigma, dtype=F64, device=p_y_xt_grad.device).pow(2)).std()
        return "cov" if std > self.denoiser_mean_error_threshold else "vjp"

    def fh_finish(self, mat, p_y_xt_grad, x_det, m_det, sigma, std=None, cov_mat=None):
        """`std`: (p_y_xt_grad * sigma^2).std() when the caller has already reduced it (the lock-step sampler does it for
        the whole batch with one device -> host transfer instead of one per image); `cov_mat`: C . mat when the caller has
        already applied the covariance (for the whole batch in one kernel sequence)."""
        cm, rec, s = self.covariance_model, self._rec, float(sigma)
        sig2 = torch.as_tensor(sigma, dtype=F64, device=m_det.device).pow(2)
        if self.fh_branch(p_y_xt_grad, sigma, std) == "vjp":
            p_y_xt_grad = p_y_xt_grad * self.cond_scaling
            rec["branch"] = "vjp"
        else:
            cv = cov_mat if cov_mat is not None else cm.denoiser_cov_vector_dot(mat.detach(), use_cuda=True)
            p_y_xt_grad = cv * self.cond_scaling / sig2
            rec["branch"] = "cov"
        x_0_mean_new = m_det + p_y_xt_grad * sig2
        self._finish_record(x_0_mean_new, x_det, m_det, s)
        return x_0_mean_new

    def _finish_record(self, x_0_mean_new, x_det, m_det, s):
        cm, rec = self.covariance_model, self._rec
        rec["k"], rec["sigma"] = cm.k, s
        if os.environ.get("FH_TRACE_SUMS"):  # debugging aid: costs a device sync per call
            rec["out_sum"] = float(x_0_mean_new.double().sum())
        self.trace.append(rec)
        self.sigmas.append(s)
        self.xs.append(x_det)
        self.denoiser_means.append(m_det)

    @staticmethod
    def fh_finish_batched(mechs, g, x_det, m_det, sigma, s, branch, cov_all=None):
        """`fh_finish` of a lock-step batch whose images all take the same branch and share `cond_scaling`: the three or four
        elementwise operations run once over [B,3,S,S] instead of once per image (the same arithmetic per element), the
        per-image bookkeeping stays on the host.  `s` = float(sigma) (no device read-back), `branch` in {"vjp", "cov"},
        `cov_all` = C . mat of the whole batch for the "cov" branch.  Returns x0_mean_new [B,3,S,S]."""
        cs = mechs[0].cond_scaling
        sig2 = torch.as_tensor(sigma, dtype=F64, device=m_det.device).pow(2)
        p = g * cs if branch == "vjp" else cov_all * cs / sig2
        out = m_det + p * sig2
        for b, mm in enumerate(mechs):
            mm._rec["branch"] = branch
            mm._finish_record(out[b:b + 1], x_det[b:b + 1], m_det[b:b + 1], s)
        return out



# ---------------------------------------------------------------------------------------------- comparison methods
# The scalar-variance methods of the reference on the same operator / UNet-VJP kernels.  Their mat solver
# is mat = A^T (theta A A^T + sigma_y^2 I)^-1 (y - A x0) with a scalar theta (the Fourier closed forms
# conditioning_mechanisms.py:357, :454, :608); here that system goes through the device CG with C = theta I to 1e-10.
def _scalar_mat(operator, y, x0_mean, theta, data_dim):
    scal = ScalarCovariance(float(theta), data_dim, x0_mean.device, getattr(operator, "ctx_slot", 0))
    return solve_customcuda(operator, y, x0_mean.detach(), scal, 1.0, 1.0, None, rtol=1e-10)


def _variance_mat(operator, y, x0_mean, theta, data_dim, rtol=1e-4, info_out=None):
    """mat = A^T (sigma_y^2 I + A diag(theta) A^T)^-1 (y - A x0) for a scalar or PER-PIXEL image-space variance theta
    (`_inpainting_mat` / `_deblur_mat` / `_super_resolution_mat`, :353-382, :449-484, :602-639).  A scalar goes through the
    closed form's system at 1e-10; a per-pixel field is the reference's scipy-CG branch, here the device CG with the
    covariance representation D = theta, no factor columns, identity basis."""
    theta = torch.as_tensor(theta)
    if theta.numel() == 1:
        return _scalar_mat(operator, y, x0_mean, theta, data_dim)
    cov = ScalarCovariance(1.0, data_dim, x0_mean.device, getattr(operator, "ctx_slot", 0))
    cov.C.D = theta.detach().to(device=x0_mean.device, dtype=F64).reshape(-1).contiguous()
    if cov.C.D.numel() != data_dim:
        raise ValueError(f"per-pixel variance has {cov.C.D.numel()} entries, expected {data_dim}")
    return solve_customcuda(operator, y, x0_mean.detach(), cov, 1.0, 1.0, info_out, rtol=rtol, scipy_cg=True)


class _ScalarVarianceMechanism(ConditioningMechanism):
    def __init__(self, cond_scaling, forward_operator, clip_x0_mean, init_denoiser_variance=None,
                 init_noise_variance=None, data_dim=None, pigdm_posthoc_scaling=True, **argv):
        super().__init__(cond_scaling, forward_operator, clip_x0_mean)
        self.pigdm_posthoc_scaling = pigdm_posthoc_scaling
        self.data_dim = data_dim if data_dim is not None else int(np.prod(forward_operator.in_shape[1:]))
        self.mle_sigma_thres = 0.2
        self.argv = argv

    def _variance(self, sigma):
        raise NotImplementedError

    def _scale(self, x0_var):
        return self.cond_scaling

    def x0_mean_update(self, x_t, model, y, sigma):
        x_t = x_t.requires_grad_()
        x_0_mean, _ = model(x_t, sigma)
        x0_var = self._variance(sigma)
        mat = _scalar_mat(self.forward_operator, y, x_0_mean, x0_var, self.data_dim)
        p_y_xt_grad = grad((mat.detach() * x_0_mean).sum(), x_t)[0] * self._scale(x0_var)
        return x_0_mean + p_y_xt_grad * sigma.pow(2)


class PiGDM(_ScalarVarianceMechanism):  # conditioning_mechanisms.py:134-152
    def _variance(self, sigma):
        return sigma.pow(2) / (1 + sigma.pow(2))

    def _scale(self, x0_var):
        return (x0_var if self.pigdm_posthoc_scaling else 1) * self.cond_scaling


class PiGDM_Videodiff_schedule(_ScalarVarianceMechanism):  # :154-171
    def _variance(self, sigma):
        return sigma.pow(2)


class PengAnalytic(_ScalarVarianceMechanism):  # :87-110
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.recon_mse = torch.load(os.path.join(_DATA, "recon_mse.pt"), weights_only=True)

    def _variance(self, sigma):
        if sigma < self.mle_sigma_thres:
            idx = (self.recon_mse["sigmas"].to(sigma.device) - sigma).abs().argmin()
            return self.recon_mse["mse_list"][idx].to(sigma.device)
        return sigma.pow(2) / (1 + sigma.pow(2))


class PengConvert(_ScalarVarianceMechanism):  # :64-85 - the network's learned per-pixel variance below sigma = 0.2
    def x0_mean_update(self, x_t, model, y, sigma):
        x_t = x_t.requires_grad_()
        x_0_mean, x0_var = model(x_t, sigma)
        if not bool(sigma < self.mle_sigma_thres):
            x0_var = sigma.pow(2) / (1 + sigma.pow(2))
        mat = _variance_mat(self.forward_operator, y, x_0_mean, x0_var.detach(), self.data_dim, 1e-4)
        p_y_xt_grad = grad((mat.detach() * x_0_mean).sum(), x_t)[0] * self.cond_scaling
        return x_0_mean + p_y_xt_grad * sigma.pow(2)


class TMPD(_ScalarVarianceMechanism):  # :112-133 - variance from the row sums of the denoiser Jacobian
    def x0_mean_update(self, x_t, model, y, sigma):
        x_t = x_t.requires_grad_()
        x_0_mean_, _ = model(x_t, sigma)
        x0_var = grad(x_0_mean_.sum(), x_t)[0] * sigma.pow(2)
        mat = _variance_mat(self.forward_operator, y, x_0_mean_.detach(), x0_var.detach(), self.data_dim,
                            rtol_func_2(float(sigma)))
        x_t = x_t.detach().requires_grad_()
        x_0_mean, _ = model(x_t, sigma)
        p_y_xt_grad = grad((mat.detach() * x_0_mean).sum(), x_t)[0] * self.cond_scaling
        return x_0_mean + p_y_xt_grad * sigma.pow(2)


class DiffPIR(ConditioningMechanism):  # :173-188
    def __init__(self, cond_scaling, forward_operator, clip_x0_mean, **argv):
        super().__init__(cond_scaling, forward_operator, clip_x0_mean)
        self.lambda_ = argv["diffpir_lambda"]
        self.data_dim = int(np.prod(forward_operator.in_shape[1:]))

    def x0_mean_update(self, x_t, model, y, sigma):
        assert self.lambda_ is not None, "lambda_ must be specified for DiffPIR guidance"
        with torch.no_grad():
            x0_mean, _ = model(x_t, sigma)
        x0_var = sigma.pow(2) / self.lambda_
        mat = _scalar_mat(self.forward_operator, y, x0_mean, x0_var, self.data_dim)
        return x0_mean + mat * x0_var


class DPS(ConditioningMechanism):  # :52-63
    def x0_mean_update(self, x_t, model, y, sigma):
        x_t = x_t.requires_grad_()
        x_0_mean, _ = model(x_t, sigma)
        op = self.forward_operator
        with torch.no_grad():
            difference = y.to(x_0_mean.dtype) - op.forward(x_0_mean.detach(), noiseless=True)
            # -d||difference|| / d x0 = A^T difference / ||difference||: the operators run on HIP kernels without autograd,
            # so their exact adjoint supplies the cotangent of the UNet's input-VJP
            cot = op.forward_adjoint(difference / torch.linalg.norm(difference))
        p_y_xt_grad = grad((cot * x_0_mean).sum(), x_t)[0] * self.cond_scaling
        return x_0_mean + p_y_xt_grad * sigma.pow(2)


_BASELINES = {"dps": DPS, "pigdm": PiGDM, "pigdm_videodiff_schedule": PiGDM_Videodiff_schedule,
              "peng_analytic": PengAnalytic, "peng_convert": PengConvert, "tmpd": TMPD, "diffpir": DiffPIR}

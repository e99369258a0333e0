"""CPU oracle: restatement of the Free Hunch guidance path (torch-CPU / numpy / scipy).

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  Citations are
``file:line`` inside the reference checkout.  The code keeps the reference's
number formats on purpose (complex128 low-rank factors, float32-rounded
scalar increments, c64 transfer functions) because the product is judged
against it; it does not try to be fast.
"""
from __future__ import annotations

import math
import os
import warnings

import numpy as np
import scipy.fft
import scipy.io
import scipy.linalg
import torch

C128 = torch.complex128


# --------------------------------------------------------------------------
# a2  sigma grid            generate_conditional.py:172-201 (edm branch :199-200)
#     sigma table / snap    training/openai_preconditioning.py:122-137, 203-207
# --------------------------------------------------------------------------
def linear_sigma_table(M=1000, beta_min=1e-4, beta_max=0.02):
    """u[0..M] of iDDPMLinearPrecond (float32, descending, u[M]=0)."""
    betas = torch.cat([torch.tensor([0.0]), torch.linspace(beta_min, beta_max, M)])
    abar = torch.cumprod(1 - betas, dim=0).flip(dims=[0])
    return torch.sqrt((1 - abar) / abar)


def round_sigma_index(u, sigma):
    """Nearest table entry in float32 (the reference uses cdist+argmin, :205)."""
    s = torch.as_tensor(sigma).to(torch.float32).reshape(-1, 1)
    return (s - u.reshape(1, -1)).abs().argmin(1)


def round_sigma(u, sigma):
    sigma = torch.as_tensor(sigma)
    return u[round_sigma_index(u, sigma)].to(sigma.dtype).reshape(sigma.shape)


def edm_sigma_steps(u, num_steps, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """t_steps of conditional_sampler (generate_conditional.py:73-112): float64 [N+1], last = 0."""
    net_min, net_max = float(u[-2]), float(u[0])
    smin, smax = max(sigma_min, net_min), min(sigma_max, net_max)
    i = torch.arange(num_steps, dtype=torch.float64)
    raw = (smax ** (1 / rho) + i / (num_steps - 1) * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho
    snapped = round_sigma(u, raw)
    return torch.cat([snapped, torch.zeros_like(snapped[:1])])


# --------------------------------------------------------------------------
# a3  iDDPMLinearPrecond.forward   training/openai_preconditioning.py:167-197
# --------------------------------------------------------------------------
class LinearPrecond:
    """net(x, sigma) -> (D_x, x0_var); `model(x32, timesteps_long)` is the raw UNet."""

    def __init__(self, model, img_channels=3, M=1000, beta_min=1e-4, beta_max=0.02):
        self.model, self.M, self.img_channels = model, M, img_channels
        self.u = linear_sigma_table(M, beta_min, beta_max)
        self.sigma_min, self.sigma_max = float(self.u[M - 1]), float(self.u[0])
        betas = np.concatenate([[0.0], torch.linspace(beta_min, beta_max, M).numpy()]).astype(np.float32)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        with np.errstate(divide="ignore", invalid="ignore"):
            self.posterior_variance = betas * (1.0 - ac_prev) / (1.0 - ac)
            self.posterior_mean_coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)

    def round_sigma(self, sigma):
        return round_sigma(self.u, sigma)

    def __call__(self, x, sigma):
        x = x.to(torch.float32)
        sigma = torch.as_tensor(sigma).to(torch.float64).reshape(-1, 1, 1, 1)
        c_in = 1 / (sigma ** 2 + 1).sqrt()
        idx = round_sigma_index(self.u, sigma.reshape(-1))
        c_noise = (self.M - idx.to(torch.float32)).to(torch.long)
        out = self.model(c_in.to(torch.float32) * x, c_noise.flatten().repeat(x.shape[0]))
        F_x, vars_ = out[:, : self.img_channels], out[:, self.img_channels:]
        pv = torch.from_numpy(np.asarray(self.posterior_variance))[c_noise].float().reshape(-1, 1, 1, 1)
        pc = torch.from_numpy(np.asarray(self.posterior_mean_coef1))[c_noise].float().reshape(-1, 1, 1, 1)
        x0_var = ((vars_ - pv) / pc.pow(2)).clip(min=1e-6)
        D_x = torch.clamp(x + (-sigma) * F_x.to(torch.float32), -1, 1)
        return D_x, x0_var


# --------------------------------------------------------------------------
# orthonormal 2-D DCT-II / DCT-III  (torch-dct==0.1.6 dct_2d/idct_2d norm='ortho',
# third-party, absent; call sites online_update_bfgs.py:352-374).  SciPy stand-in.
# --------------------------------------------------------------------------
def dct2(x):
    return torch.from_numpy(scipy.fft.dctn(x.detach().numpy(), type=2, norm="ortho", axes=(-2, -1))).to(x.dtype)


def idct2(x):
    return torch.from_numpy(scipy.fft.idctn(x.detach().numpy(), type=2, norm="ortho", axes=(-2, -1))).to(x.dtype)


# --------------------------------------------------------------------------
# a7-a10  CovarianceHessianBFGS(+DCT)   conditioning_utils/online_update_bfgs.py:7-374
# --------------------------------------------------------------------------
def _sqrtm(A):  # :67-71
    if A.shape[0] == 0:
        return torch.zeros_like(A)
    return torch.from_numpy(scipy.linalg.sqrtm(A.numpy()).astype(np.complex128))


def _invert_rep(diag, U, V):
    """(diag + U U^T - V V^T)^-1 as (diag_inv, U_inv, V_inv): plain transposes.  :87-119"""
    dinv = 1 / diag
    k = U.shape[1]
    eye = torch.eye(k, dtype=C128)
    m1 = torch.linalg.inv(eye + U.T @ (dinv[:, None] * U))
    m1 = (m1 + m1.T) / 2
    Vinv = dinv[:, None] * (U @ _sqrtm(m1))
    K = Vinv.T @ V
    m2 = torch.linalg.inv(eye - V.T @ (dinv[:, None] * V) + K.T @ K)
    VR = V @ _sqrtm(m2)
    Uinv = dinv[:, None] * VR - Vinv @ (Vinv.T @ VR)
    return dinv, Uinv, Vinv


def _apply_rep(rep, v):
    d, U, V = rep
    return d * v + U @ (U.T @ v) - V @ (V.T @ v)


class OracleCovariance:
    """State = four (diag, U, V) complex128 representations: cov, inv_cov, hess, inv_hess."""

    def __init__(self, init_var, init_noise_variance, data_dim, max_vector_count=None,
                 project_to_diagonal=False, use_dct=False):
        empty = lambda: torch.zeros(data_dim, 0, dtype=C128)
        self.d = data_dim
        self.use_dct = use_dct
        self.max_vector_count = max_vector_count
        self.project_to_diagonal = project_to_diagonal
        diag = torch.ones(data_dim, dtype=C128) * init_var  # :26
        self.cov = (diag, empty(), empty())
        self._derive_from_cov(np.sqrt(init_noise_variance))  # :36, :327-330

    # -- :327-330
    def _derive_from_cov(self, sigma):
        d, U, V = self.cov
        self.icov = _invert_rep(d, U, V)
        self.hess = ((d / sigma ** 2 - 1) / sigma ** 2, U / sigma ** 2, V / sigma ** 2)
        self.ihess = _invert_rep(*self.hess)

    def _fwd(self, x):
        return dct2(x) if self.use_dct else x

    def _bwd(self, x):
        return idct2(x) if self.use_dct else x

    @property
    def k(self):
        return self.cov[1].shape[1]

    # -- :194-204 / :370-374
    def denoiser_cov_vector_dot(self, v):
        z = self._fwd(v).to(C128).reshape(-1)
        out = _apply_rep(self.cov, z).real.reshape(v.shape).to(v.dtype)
        return self._bwd(out)

    # -- :153-192 (+ DCT wrapper :357-360)
    def update_time_step(self, x_t, sigma_t, sigma_next, score_t, only_covariance=False):
        shape = x_t.shape
        assert shape[0] == 1, "Batch size must be 1"
        x = self._fwd(x_t.detach()).to(C128).reshape(-1)
        s = self._fwd(score_t.detach()).to(C128).reshape(-1)
        d_i, U_i, V_i = self.icov
        d_i = d_i + float(np.float32(sigma_next ** (-2) - sigma_t ** (-2)))  # float32 increment :166
        self.icov = (d_i, U_i, V_i)
        self.cov = _invert_rep(d_i, U_i, V_i)  # same routine, fed the inverse rep  :167-168
        if only_covariance:
            mean = x.reshape(shape).real
            return self._bwd(mean), self._bwd(mean)
        dh_i, Uh_i, Vh_i = self.ihess
        new_dh_i = dh_i - float(np.float32(sigma_next ** 2 - sigma_t ** 2))  # :172
        new_hess = _invert_rep(new_dh_i, Uh_i, Vh_i)
        t = _apply_rep(self.ihess, s)
        new_score = _apply_rep(new_hess, t).real + 0j
        new_mean = (x + sigma_next ** 2 * new_score).real
        self.ihess = (new_dh_i, Uh_i, Vh_i)
        self.hess = new_hess
        return self._bwd(new_mean.reshape(shape)), self._bwd(new_score.real.reshape(shape))

    # -- :250-312 (+ DCT wrapper :362-368)
    def update_space_step(self, mean_x, mean_xnext, sigma_t, x, xnext):
        assert x.shape[0] == 1, "Batch size must be 1"
        cv = lambda a: self._fwd(a.detach()).to(C128).reshape(-1)
        x, xnext, m0, m1 = cv(x), cv(xnext), cv(mean_x), cv(mean_xnext)
        dx = xnext - x
        de = sigma_t ** 2 * (m1 - m0)
        gamma = 1 / (dx @ de)
        cdx = _apply_rep(self.cov, dx)
        v = cdx / torch.sqrt(cdx @ dx)
        u = de * torch.sqrt(gamma)
        d, U, V = self.cov
        if self.project_to_diagonal:
            new_cov = (d + u * u - v * v, U, V)
        else:
            new_cov = (d, torch.cat((U, u[:, None]), -1), torch.cat((V, v[:, None]), -1))
        new_icov = _invert_rep(*new_cov)
        dh, Uh, Vh = self.hess
        new_hess = ((new_cov[0] / sigma_t ** 2 - 1) / sigma_t ** 2,
                    torch.cat((Uh, (u / sigma_t ** 2)[:, None]), -1),
                    torch.cat((Vh, (v / sigma_t ** 2)[:, None]), -1))
        new_ihess = _invert_rep(*new_hess)
        self.cov, self.icov, self.hess, self.ihess = new_cov, new_icov, new_hess, new_ihess
        if self.max_vector_count is not None:  # :233-245
            n = self.max_vector_count
            d, U, V = self.cov
            if n == 0:
                self.cov = (d, U[:, :0], V[:, :0])
                self._derive_from_cov(sigma_t)
            elif U.shape[1] > n:
                self.cov = (d, U[:, -n:], V[:, -n:])
                self._derive_from_cov(sigma_t)

    def dense(self):  # :320-325
        def full(rep):
            d, U, V = rep
            return torch.diag(d) + U @ U.T - V @ V.T
        return full(self.cov), full(self.icov), full(self.hess), full(self.ihess)


def make_covariance(image_base_covariance, data_dir, init_noise_variance, data_dim,
                    max_vector_count=None, project_to_diagonal=False):
    """BFGSOnlineUpdate.__init__ dispatch, conditioning_mechanisms.py:197-212."""
    if image_base_covariance == "identity":
        return OracleCovariance(1.0, init_noise_variance, data_dim, max_vector_count, project_to_diagonal, False)
    if image_base_covariance in ("dct_diagonal", "dct_diagonal_noinfo"):
        if image_base_covariance == "dct_diagonal":
            var = torch.load(os.path.join(data_dir, "dct_variance.pt"), weights_only=True).reshape(-1)
            assert var.numel() == data_dim
        else:
            var = torch.ones(data_dim)
        return OracleCovariance(var, init_noise_variance, data_dim, max_vector_count, project_to_diagonal, True)
    raise ValueError(f"unsupported image_base_covariance: {image_base_covariance}")


# --------------------------------------------------------------------------
# a13  measurement operators   measurement_utils/measurements.py:87-246,
#      utils_sisr.py:9-96, resizer.py:8-160
# --------------------------------------------------------------------------
def p2o(psf, shape):  # utils_sisr.py:22-41
    otf = torch.zeros(psf.shape[:-2] + tuple(shape), dtype=psf.dtype)
    otf[..., : psf.shape[-2], : psf.shape[-1]] = psf
    otf = torch.roll(otf, (-int(psf.shape[-2] / 2), -int(psf.shape[-1] / 2)), dims=(-2, -1))
    return torch.fft.fftn(otf, dim=(-2, -1))


def zero_insert(x, sf):  # utils_sisr.py:44-52
    z = torch.zeros(x.shape[:-2] + (x.shape[-2] * sf, x.shape[-1] * sf), dtype=x.dtype)
    z[..., ::sf, ::sf] = x
    return z


def decimate(x, sf):  # utils_sisr.py:55-61
    return x[..., ::sf, ::sf]


def pre_calculate(x, k, sf):  # utils_sisr.py:79-96
    FB = p2o(k, (x.shape[-2] * sf, x.shape[-1] * sf))
    FBC = torch.conj(FB)
    F2B = torch.abs(FB) ** 2
    FBFy = FBC * torch.fft.fftn(zero_insert(x, sf), dim=(-2, -1))
    return FB, FBC, F2B, FBFy


def _cubic(x):  # resizer.py:165-170
    ax = np.abs(x)
    return ((1.5 * ax ** 3 - 2.5 * ax ** 2 + 1) * (ax <= 1)
            + (-0.5 * ax ** 3 + 2.5 * ax ** 2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2)))


def bicubic_matrix(n_in, scale):
    """Dense [n_out, n_in] float32 matrix of the antialiased cubic Resizer along one axis
    (resizer.py:103-160: stretched kernel, normalised weights, mirror padding)."""
    n_out = int(np.ceil(n_in * scale))
    kw = 4.0 / scale
    out = np.arange(1, n_out + 1) - (n_out - n_in * scale) / 2
    match = out / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(match - kw / 2)
    fov = (left[:, None] + np.arange(np.ceil(kw) + 2) - 1).astype(np.int16)
    w = scale * _cubic(scale * (match[:, None] - fov - 1))
    sw = w.sum(1)
    sw[sw == 0] = 1.0
    w = w / sw[:, None]
    mirror = np.concatenate((np.arange(n_in), np.arange(n_in - 1, -1, -1)))
    fov = mirror[np.mod(fov, mirror.shape[0])]
    keep = np.nonzero(np.any(w, axis=0))[0]  # drop taps that are zero for every output  :155-158
    w, fov = w[:, keep], fov[:, keep]
    R = np.zeros((n_out, n_in), dtype=np.float64)
    w32 = w.astype(np.float32)
    for o in range(n_out):
        for j in range(fov.shape[1]):
            R[o, fov[o, j]] += w32[o, j]
    return R, fov, w32


def resizer_apply(x, scale):
    """Resizer(in_shape, scale).forward(x) for NCHW x, same scale on H and W (resizer.py:55-74):
    float32 gather-multiply-sum along H then W (sorted_dims order for equal scales: dim 2, 3)."""
    for dim in (2, 3):
        _, fov, w32 = bicubic_matrix(x.shape[dim], scale)
        xt = torch.transpose(x, dim, 0)
        wt = torch.from_numpy(w32.T.copy()).reshape(w32.shape[1], w32.shape[0], 1, 1, 1)
        xt = torch.sum(xt[torch.from_numpy(fov.T.astype(np.int64))] * wt, dim=0)
        x = torch.transpose(xt, dim, 0)
    return x


class OracleOperator:
    """The four linear operators.  `noise` is drawn by the caller (RNG is an input, never reproduced)."""

    def __init__(self, name, in_shape, sigma_s, kernel=None, scale_factor=None, mask=None, otf_double=False):
        """`otf_double` (test-only, NOT the reference's arithmetic): build the blur OTF from the float32 kernel values in
        complex128 instead of the reference's complex64 (`p2o` of a float32 PSF, utils_sisr.py:22-41).  The operator then
        equals the exact circular convolution with those taps to 1e-16 instead of 1e-7; used by the tests to separate the
        OTF rounding from the other differences between the reference and the tap-list kernels."""
        self.otf_double = bool(otf_double)
        self.name, self.in_shape = name, tuple(in_shape)
        self.sigma_s = torch.tensor([sigma_s], dtype=torch.float32)
        self.scale_factor = scale_factor
        self.mask = mask
        self.kernel = None if kernel is None else torch.as_tensor(np.asarray(kernel)).to(torch.float32)
        self.pre_calculated = None

    def _k(self):
        k = self.kernel.view(1, 1, *self.kernel.shape)
        return k.double() if self.otf_double else k

    def forward(self, data, noise=None):
        """y = A x (+ sigma_s * noise); caches pre_calculated like measurements.py:109,146,186."""
        if self.name in ("gaussian_blur", "motion_blur"):  # :137-149, :175-191
            FB, FBC, F2B, _ = pre_calculate(data, self._k(), 1)
            y = torch.fft.ifft2(FB * torch.fft.fft2(data)).real
            if noise is not None:
                y = y + self.sigma_s * noise
            self.pre_calculated = (FB, FBC, F2B, FBC * torch.fft.fft2(y))
            return y
        if self.name == "super_resolution":  # :104-112
            y = resizer_apply(data, 1 / self.scale_factor)
            if noise is not None:
                y = y + self.sigma_s * noise
            self.pre_calculated = pre_calculate(y, self._k(), self.scale_factor)
            return y
        if self.name == "inpainting":  # :213-229  (noise added before masking)
            y = data.clone()
            if noise is not None:
                y = y + self.sigma_s * noise
            return y * self.mask
        raise NameError(f"Name {self.name} is not defined.")

    def transpose(self, y):
        if self.name in ("gaussian_blur", "motion_blur"):  # :151-157, :193-199
            _, FBC, _, _ = pre_calculate(y, self._k(), 1)
            return torch.fft.ifft2(FBC * torch.fft.fft2(y)).real
        if self.name == "super_resolution":  # :114-120
            return torch.fft.ifft2(pre_calculate(y, self._k(), self.scale_factor)[3]).real
        return y.clone()  # :231-241


def random_mask(image_size, prob_range, rng):
    """MaskGenerator._retrieve_random (measurements.py:287-299) with an explicit numpy Generator-like
    object exposing uniform() and choice(); returns [1,3,S,S] float32."""
    total = image_size ** 2
    prob = rng.uniform(*prob_range)
    vec = torch.ones(total)
    vec[torch.from_numpy(np.asarray(rng.choice(total, int(total * prob), replace=False)))] = 0
    return vec.view(1, 1, image_size, image_size).repeat(1, 3, 1, 1)


# --------------------------------------------------------------------------
# a11  cg()   conditioning_utils/cg.py:118-292  (M = I, x0 = b)
# --------------------------------------------------------------------------
def cg(A_mm, b, rtol=1e-3, atol=0.0, maxiter=5000):
    x = b
    r = b - A_mm(x)
    p = r.clone()
    r_norm = torch.norm(r)
    stop = max(rtol * torch.norm(b), atol)
    optimal, k = False, 0
    rz = torch.dot(r, r)
    for k in range(1, maxiter + 1):
        Ap = A_mm(p)
        pAp = torch.dot(p, Ap)
        if pAp <= 1e-16:
            break
        alpha = rz / pAp
        x = x + alpha * p
        r_new = r - alpha * Ap
        n_new = torch.norm(r_new)
        if n_new <= stop:
            optimal, r, r_norm = True, r_new, n_new
            break
        rz_new = torch.dot(r_new, r_new)
        p = r_new + (rz_new / rz) * p
        r, rz, r_norm = r_new, rz_new, n_new
    return x, {"niter": k, "optimal": optimal, "residual_norm": r_norm}


# --------------------------------------------------------------------------
# a12  rtol_func + the three customcuda solvers   conditioning_mechanisms.py:307-323,
#      384-419, 489-527, 641-675
# --------------------------------------------------------------------------
def rtol_func(sigma, rtol_max=1.0, rtol_min=1e-14):
    lo, hi = 0.1, 80.0
    sigma = max(min(sigma, hi), max(lo, sigma))  # (sic) :313 - only the lower clamp acts
    f = ((math.log10(sigma) - math.log10(lo)) / (math.log10(hi) - math.log10(lo))) ** 0.1
    return 10 ** (f * (math.log10(rtol_max) - math.log10(rtol_min)) + math.log10(rtol_min))


def system(op, y, x0_mean, cov):
    """The linear system of the three customcuda solvers (conditioning_mechanisms.py:384-419, 489-527, 641-675):
    returns (A_mm, b, back, shape) with A_mm(u) = sigma_y^2 u + A C A^T u on flattened measurement-shaped vectors,
    b = y - A x0_mean and mat = back(solution)."""
    name = op.name
    C = cov.denoiser_cov_vector_dot
    if name == "inpainting":
        s2 = op.sigma_s.clip(min=0.001) ** 2
        mask = op.mask

        def A_mm(u):
            u = u.reshape(x0_mean.shape)
            return (s2 * u + mask * C(mask * u)).flatten()

        return A_mm, (mask * y - mask * x0_mean).flatten(), (lambda sol: sol.reshape(x0_mean.shape)), x0_mean.shape
    if name in ("gaussian_blur", "motion_blur"):
        s2 = op.sigma_s.clip(min=0.001) ** 2
        FB, FBC, _, _ = op.pre_calculated
        blur = lambda v: torch.fft.ifft2(FB * torch.fft.fft2(v)).real
        blur_t = lambda v: torch.fft.ifft2(FBC * torch.fft.fft2(v)).real

        def A_mm(u):
            u = u.reshape(y.shape)
            return (s2 * u + blur(C(blur_t(u)))).flatten()

        return A_mm, (y - blur(x0_mean)).flatten(), (lambda sol: blur_t(sol.reshape(y.shape))), y.shape
    if name == "super_resolution":
        s2 = op.sigma_s.clip(min=0.001).clip(min=1e-2) ** 2
        sf = op.scale_factor
        FB, FBC, _, _ = op.pre_calculated
        blur = lambda v: torch.fft.ifft2(FB * torch.fft.fft2(v))
        blur_t = lambda v: torch.fft.ifft2(FBC * torch.fft.fft2(v)).real

        def A_mm(u):
            u = u.reshape(y.shape)
            return (s2 * u + decimate(blur(C(blur_t(zero_insert(u, sf)))).real, sf)).flatten()

        return (A_mm, (y - decimate(blur(x0_mean), sf)).real.flatten(),
                (lambda sol: blur_t(zero_insert(sol.reshape(y.shape), sf))), y.shape)
    raise ValueError("Invalid operator name. Please choose 'gaussian_blur', 'super_resolution', "
                     "'motion_blur', or 'inpainting'.")


def solve_mat(op, y, x0_mean, cov, max_rtol, sigma_t, info_out=None, maxiter=5000, rtol=None):
    """mat = A^T (A C A^T + sigma_y^2 I)^-1 (y - A x0_mean) by CG; returns same dtype flow as the reference.
    `maxiter` / `rtol` (the reference fixes 5000 and rtol_func) let the tests stop both sides after the same few iterations."""
    rtol = rtol_func(sigma_t, max_rtol) if rtol is None else rtol
    A_mm, b, back, _ = system(op, y, x0_mean, cov)
    sol, info = cg(A_mm, b, rtol=rtol, maxiter=maxiter)
    if info["niter"] == (5000 if op.name == "inpainting" else 2000):  # the reference's (inconsistent) guards :415, :521, :669
        warnings.warn("CG not converge.")
    mat = back(sol)
    if info_out is not None:
        info_out.append({"niter": info["niter"], "optimal": info["optimal"],
                         "residual_norm": float(info["residual_norm"]), "rtol": rtol})
    return mat


def analytic_mat(op, y, x0_mean, theta):
    """Scalar-variance closed forms used when `use_analytic_var_at_end` (conditioning_mechanisms.py:357-358,
    :454-455, :608-610): mat = A^T (theta A A^T + sigma_s^2 I)^-1 (y - A x0_mean) evaluated in Fourier space."""
    if op.name == "inpainting":
        s2 = op.sigma_s.clip(min=0.001).pow(2)
        return (op.mask * y - op.mask * x0_mean) / (s2 + theta)
    FB, FBC, F2B, _ = op.pre_calculated
    fft2, ifft2 = torch.fft.fft2, torch.fft.ifft2
    if op.name in ("gaussian_blur", "motion_blur"):
        s2 = op.sigma_s.clip(min=0.001).pow(2)
        return ifft2(fft2(y - ifft2(FB * fft2(x0_mean))) / (s2 + theta * F2B) * FBC).real
    s2 = op.sigma_s.clip(min=0.001).clip(min=1e-2).pow(2)
    sf = op.scale_factor
    blocks = torch.stack(torch.chunk(F2B, sf, dim=2), dim=4)          # utils_sisr.splits :9-19
    blocks = torch.cat(torch.chunk(blocks, sf, dim=3), dim=4)
    invW = torch.mean(blocks, dim=-1)
    res = fft2(y - decimate(ifft2(FB * fft2(x0_mean)), sf)) / (s2 + theta * invW)
    return ifft2(FBC * res.repeat(1, 1, sf, sf)).real


# --------------------------------------------------------------------------
# a6  BFGSOnlineUpdate   conditioning_mechanisms.py:190-294 (+ base class :38-50)
# --------------------------------------------------------------------------
class OracleFreeHunch:
    def __init__(self, cond_scaling, forward_operator, clip_x0_mean, init_noise_variance, data_dim,
                 image_base_covariance="dct_diagonal", data_dir=None, max_vector_count=100000,
                 project_to_diagonal=False, do_space_updates=True, denoiser_mean_error_threshold=0.2,
                 use_analytical_score_time_update=True, space_step_update_threshold=10.0,
                 space_step_update_lower_threshold=1.0, max_rtol=1.0, use_analytic_var_at_end=False, recon_mse=None):
        self.cond_scaling, self.op, self.clip = cond_scaling, forward_operator, clip_x0_mean
        self.cov = make_covariance(image_base_covariance, data_dir, float(init_noise_variance), data_dim,
                                   max_vector_count, project_to_diagonal)
        self.do_space_updates = do_space_updates
        self.err_thr = denoiser_mean_error_threshold
        self.analytic_score = use_analytical_score_time_update
        self.upper, self.lower = space_step_update_threshold, space_step_update_lower_threshold
        self.max_rtol = max_rtol
        self.analytic_end, self.recon_mse = use_analytic_var_at_end, recon_mse  # :224-226, threshold 0.2 (:227)
        self.sigmas, self.xs, self.means = [], [], []
        self.trace = []  # one dict per call: niter, branch, k ...

    def __call__(self, x_t, net, y, sigma):
        out = self._update(x_t, net, y, sigma)
        return out.clip(-1, 1) if self.clip else out

    def _update(self, x_t, net, y, sigma):
        rec = {"sigma": float(sigma), "time_update": False, "space_update": False}
        x_t = x_t.detach().requires_grad_()
        with torch.enable_grad():
            x0, _ = net(x_t, sigma)
        s = float(sigma)
        if self.do_space_updates:
            pred = None
            if self.sigmas and s != self.sigmas[-1]:
                score_prev = (self.means[-1] - self.xs[-1]) / self.sigmas[-1] ** 2
                pred, _ = self.cov.update_time_step(self.xs[-1], self.sigmas[-1], s, score_prev)
                rec["time_update"] = True
            elif self.sigmas:
                pred = self.means[-1]
            if self.xs and not torch.allclose(x_t.detach(), self.xs[-1]):
                if not self.analytic_score:
                    with torch.no_grad():
                        pred, _ = net(self.xs[-1], sigma)
                if self.lower < s < self.upper:
                    self.cov.update_space_step(pred, x0.detach(), s, self.xs[-1], x_t.detach())
                    rec["space_update"] = True
        elif self.sigmas and s != self.sigmas[-1]:
            score_prev = (self.means[-1] - self.xs[-1]) / self.sigmas[-1] ** 2
            self.cov.update_time_step(self.xs[-1], self.sigmas[-1], s, score_prev, only_covariance=True)
            rec["time_update"] = True
        info = []
        with torch.no_grad():
            mat = solve_mat(self.op, y, x0.detach(), self.cov, self.max_rtol, s, info)
        rec.update(info[0])
        analytic = self.analytic_end and s < 0.2
        if analytic:  # :273-278
            idx = (self.recon_mse["sigmas"].double() - s).abs().argmin()
            mat = analytic_mat(self.op, y, x0.detach(), self.recon_mse["mse_list"][idx].double())
        (g,) = torch.autograd.grad((mat.detach() * x0).sum(), x_t)
        sig2 = torch.as_tensor(sigma, dtype=torch.float64) ** 2
        if analytic:
            g = g * self.cond_scaling
            rec["branch"] = "vjp"
        elif rec.setdefault("std", float((g * sig2).std())) > self.err_thr:  # :283-285
            g = self.cov.denoiser_cov_vector_dot(mat.detach()) * self.cond_scaling / sig2
            rec["branch"] = "cov"
        else:
            g = g * self.cond_scaling
            rec["branch"] = "vjp"
        new = x0.detach() + g * sig2
        rec["k"] = self.cov.k
        self.trace.append(rec)
        self.sigmas.append(s)
        self.xs.append(x_t.detach())
        self.means.append(x0.detach())
        return new


# --------------------------------------------------------------------------
# a1  conditional_sampler   generate_conditional.py:38-169  (edm / linear / none, S_churn = 0)
# --------------------------------------------------------------------------
# --------------------------------------------------------------------------
# scalar-variance comparison methods   conditioning_mechanisms.py:52-63 (DPS), :87-110 (PengAnalytic), :134-152 (PiGDM),
# :154-171 (PiGDM video-diffusion schedule), :173-188 (DiffPIR); their mat solvers are the closed forms of analytic_mat
# --------------------------------------------------------------------------
class OracleBaseline:
    def __init__(self, kind, cond_scaling, forward_operator, clip_x0_mean, pigdm_posthoc_scaling=False,
                 diffpir_lambda=None, recon_mse=None):
        assert kind in ("dps", "pigdm", "pigdm_videodiff_schedule", "diffpir", "peng_analytic")
        self.kind, self.cond_scaling, self.op, self.clip = kind, cond_scaling, forward_operator, clip_x0_mean
        self.posthoc, self.lam, self.recon_mse = pigdm_posthoc_scaling, diffpir_lambda, recon_mse
        self.out_sums = []

    def _forward(self, x):  # forward(noiseless=True) without touching the cached pre_calculated of the measurement
        op = self.op
        if op.name in ("gaussian_blur", "motion_blur"):
            FB = op.pre_calculated[0]
            return torch.fft.ifft2(FB * torch.fft.fft2(x)).real
        if op.name == "super_resolution":
            return resizer_apply(x, 1 / op.scale_factor)
        return x * op.mask

    def __call__(self, x_t, net, y, sigma):
        sigma = torch.as_tensor(sigma, dtype=torch.float64)
        x_t = x_t.detach().requires_grad_()
        x0, _ = net(x_t, sigma)
        s2 = sigma.pow(2)
        if self.kind == "dps":
            norm = torch.linalg.norm(y - self._forward(x0))
            g = -torch.autograd.grad(norm, x_t)[0] * self.cond_scaling
            out = x0 + g * s2
        elif self.kind == "diffpir":
            var = s2 / self.lam
            out = x0 + analytic_mat(self.op, y, x0.detach(), var) * var
        else:
            if self.kind == "pigdm_videodiff_schedule":
                var = s2
            elif self.kind == "peng_analytic" and float(sigma) < 0.2:
                var = self.recon_mse["mse_list"][(self.recon_mse["sigmas"] - sigma).abs().argmin()]
            else:
                var = s2 / (1 + s2)
            mat = analytic_mat(self.op, y, x0.detach(), var)
            scale = (var if (self.kind == "pigdm" and self.posthoc) else 1) * self.cond_scaling
            g = torch.autograd.grad((mat.detach() * x0).sum(), x_t)[0] * scale
            out = x0 + g * s2
        out = out.detach()
        self.out_sums.append(float(out.double().sum()))
        return out.clip(-1, 1) if self.clip else out


def conditional_sampler(net, noise, y, operator, num_steps=30, sigma_min=0.002, sigma_max=80.0, rho=7.0,
                        solver="heun", mechanism_factory=None):
    """Euler/Heun loop.  `y` (the measurement) and `noise` are inputs; returns (x_final f64, mechanism)."""
    assert solver in ("euler", "heun")
    t = edm_sigma_steps(net.u, num_steps, sigma_min, sigma_max, rho)
    x_next = noise.to(torch.float64) * t[0]
    mech = mechanism_factory(operator, t[0] ** 2, x_next.shape[1:].numel())
    for i, (t_cur, t_nxt) in enumerate(zip(t[:-1], t[1:])):
        x_hat = x_next  # gamma = 0: t_hat = round_sigma(t_cur) = t_cur, churn term is exactly 0
        h = t_nxt - t_cur
        den = mech(x_hat, net, y, t_cur)
        score = -(x_hat - den) / t_cur ** 2           # :146-147, kept literal for rounding
        d_cur = -score * t_cur
        x_prime = x_hat + h * d_cur
        t_prime = t_cur + h                           # :150 - may differ from t_nxt by one ulp
        if solver == "euler" or i == num_steps - 1:
            x_next = x_hat + h * d_cur
        else:
            den2 = mech(x_prime, net, y, t_prime)
            d_prime = (1 / t_prime) * x_prime - (1 / t_prime) * den2
            x_next = x_hat + h * (0.5 * d_cur + 0.5 * d_prime)
    return x_next, mech


# --------------------------------------------------------------------------
# a14  StandardRGBEncoder   training/encoders.py:61-73
# --------------------------------------------------------------------------
def encode_rgb(u8):
    return u8.to(torch.float32) / 127.5 - 1


def decode_rgb(x):
    return (x.to(torch.float32) * 127.5 + 128).clip(0, 255).to(torch.uint8)


# --------------------------------------------------------------------------
# dense known-answer helpers  online_update_bfgs.py:377-463 (analytic cross-check for a7-a10)
# --------------------------------------------------------------------------
def dense_time_update(x, cov, icov, hess, ihess, score, s, s_next):
    eye = torch.eye(x.shape[-1], dtype=cov.dtype)
    n_icov = icov + (s_next ** -2 - s ** -2) * eye
    n_ihess = ihess - (s_next ** 2 - s ** 2) * eye
    n_hess = torch.linalg.inv(n_ihess)
    n_score = n_hess @ ihess @ score
    return torch.linalg.inv(n_icov), n_icov, n_hess, n_ihess, n_score, x + s_next ** 2 * n_score


def dense_space_update(cov, icov, mean_x, mean_xn, s, dx):
    d = dx.shape[-1]
    eye = torch.eye(d, dtype=cov.dtype)
    de = s ** 2 * (mean_xn - mean_x)
    g = 1 / (dx @ de)
    cdx = cov @ dx
    n_cov = cov - torch.outer(cdx, cdx) / (dx @ cdx) + torch.outer(de, de) * g
    n_icov = (eye - torch.outer(dx, de) * g) @ icov @ (eye - torch.outer(de, dx) * g) + torch.outer(dx, dx) * g
    n_hess = (n_cov / s ** 2 - eye) / s ** 2
    return n_cov, n_icov, n_hess, torch.linalg.inv(n_hess + 1e-10 * eye)

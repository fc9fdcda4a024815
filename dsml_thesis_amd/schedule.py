"""Noise schedules and DDIM coefficient tables (host side, computed once per sampler call).

Restates the arithmetic of the reference including its mixed float32/float64 evaluation, because
the fp32 parity bar applies to the coefficients too:
  make_beta_schedule            ldm/modules/diffusionmodules/util.py:21-25
  DDPM.register_schedule        ldm/models/diffusion/ddpm.py:117-169
  make_ddim_timesteps           util.py:46-60
  make_ddim_sampling_parameters util.py:63-74  (+ DDIMSampler.make_schedule, ddim.py:42-50)
"""
import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        al = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        al = al / al[0]
        return np.clip(1 - al[1:] / al[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def schedule_buffers(betas, v_posterior=0.0):
    """float64 math, float32 buffers, names as registered by DDPM.register_schedule."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    pv = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    return dict(
        betas=f32(betas), alphas_cumprod=f32(ac), alphas_cumprod_prev=f32(ac_prev),
        sqrt_alphas_cumprod=f32(np.sqrt(ac)), sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - ac)),
        log_one_minus_alphas_cumprod=f32(np.log(1.0 - ac)), sqrt_recip_alphas_cumprod=f32(np.sqrt(1.0 / ac)),
        sqrt_recipm1_alphas_cumprod=f32(np.sqrt(1.0 / ac - 1)), posterior_variance=f32(pv),
        posterior_log_variance_clipped=f32(np.log(np.maximum(pv, 1e-20))),
        posterior_mean_coef1=f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        posterior_mean_coef2=f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)))


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    return steps + 1


def make_ddim_timesteps_strength(num_ddim_timesteps, num_ddpm_timesteps, strength=1.0):
    """Strength-scaled 'uniform' schedule of the latent-manipulation scripts (compute_latents.py:52-73,
    latent_manipulation_tuned.py:52-72): [1] + int(linspace(0,1,S) * int(T*strength))[1:]."""
    ts = np.linspace(0, 1, num_ddim_timesteps) * int(num_ddpm_timesteps * strength)
    ts = [int(s) for s in list(ts)]
    return np.asarray([1] + ts[1:])


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """alphacums: float32 CPU tensor.  Returns (sigmas f64 ndarray, alphas f32 ndarray, alphas_prev f64 ndarray)
    with the reference's promotion rules: `ndarray / Tensor` is reciprocal(float32)*float64, `Tensor / ndarray`
    promotes to float64 first (see oracle/ldm_oracle.py::make_ddim_tables for the derivation)."""
    ac = torch.as_tensor(alphacums, dtype=torch.float32).cpu()
    if int(np.max(ddim_timesteps)) >= ac.shape[0]:
        raise IndexError(f"index {int(np.max(ddim_timesteps))} is out of bounds for dimension 0 with size {ac.shape[0]}")
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0].item()] + ac[ddim_timesteps[:-1]].tolist())
    a64 = alphas.double().numpy()
    recip = (1 - alphas).reciprocal().double().numpy()
    sigmas = eta * np.sqrt((1 - alphas_prev) * recip * (1 - a64 / alphas_prev))
    return sigmas, alphas.numpy(), alphas_prev


def ddim_inversion_table(alphacums, ddim_timesteps):
    """[S][4] rows for the forward (inversion) DDIM update q_sample_ddim (compute_latents.py:364-406), in the slot
    order of the sampling table so that the same update kernel serves both directions:
      slot a_t <- alphas_prev[i], slot a_prev <- alphas[i] (the 'next' level), sigma <- 0,
      slot sqrt(1-a_t) <- sqrt(1 - alphas_prev[i])  (float64 ndarray math, rounded once like torch.full does)."""
    _, alphas, alphas_prev = make_ddim_sampling_parameters(alphacums, ddim_timesteps, 0.0)
    tab = np.stack([alphas_prev.astype(np.float32), alphas.astype(np.float32), np.zeros(len(alphas), np.float32),
                    np.sqrt(1.0 - alphas_prev).astype(np.float32)], axis=1)
    return np.ascontiguousarray(tab)


def ddim_step_table(alphacums, ddim_timesteps, eta):
    """[S][4] float32 rows (a_t, a_prev, sigma_t, sqrt(1-a_t)): what torch.full() receives at ddim.py:188-191."""
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta)
    tab = np.stack([alphas.astype(np.float32), alphas_prev.astype(np.float32), np.asarray(sigmas).astype(np.float32),
                    np.sqrt(1.0 - alphas).astype(np.float32)], axis=1)
    return np.ascontiguousarray(tab)

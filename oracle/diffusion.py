"""Oracle: noise schedules, time embedding and the three reverse updates (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch


# ----------------------------------------------------------------------------
# time embedding
# ----------------------------------------------------------------------------
def gaussian_fourier_projection(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """cat(sin, cos)(x * w * 2*pi); x [N,1], w [E] -> [N,2E].

    diffusion/diffusion_helpers.py:23-25 (multiplication order kept: x*w, *2, *pi).
    """
    proj = x * w[None, :] * 2 * np.pi
    return torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)


# ----------------------------------------------------------------------------
# VE (fractional coordinates, periodic)   diffusion/diffusion_helpers.py:28-81
# ----------------------------------------------------------------------------
def ve_sigmas(T: int, sigma_min: float, sigma_max: float, dtype=None) -> torch.Tensor:
    """Geometric sigma ladder with T+1 points (:38-41); reference uses the default dtype."""
    return torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), T + 1, dtype=dtype))


def ve_reverse(sigmas, xt, eps_x, t, z):
    """One ancestral VE step followed by wrap to [0,1) (:65-81).

    ``z`` is the injected standard-normal noise the reference draws with
    torch.randn_like(xt) at :79.
    """
    s = sigmas[t].view(-1, 1)
    s_prev = torch.where((t == 0).view(-1, 1), torch.zeros_like(s), sigmas[t - 1].view(-1, 1))
    mean = xt - eps_x * (s ** 2 - s_prev ** 2)
    std = torch.sqrt((s_prev ** 2 * (s ** 2 - s_prev ** 2)) / (s ** 2))
    return (mean + std * z) % 1


# ----------------------------------------------------------------------------
# VP (lattice lengths)   diffusion/diffusion_helpers.py:134-199
# ----------------------------------------------------------------------------
def vp_schedule(T: int, s: float = 0.0001, power: float = 2, clipmax: float = 0.999, dtype=None):
    """Cosine schedule of :139-154.  Returns (alpha_bars, betas, sigmas), each [T+1].

    dtype quirk kept from the reference: ``t`` is float32 (:141), so alpha_bars
    is ALWAYS float32; betas/sigmas are concatenated with torch.zeros([1]) of
    the DEFAULT dtype (:146-151) and therefore take that dtype (float64 in the
    reference's runs) while holding float32-computed values.  ``dtype`` stands
    for that default dtype (None = torch's current default).
    """
    t = torch.arange(0, T + 1, dtype=torch.float)
    f_t = torch.cos((np.pi / 2) * ((t / T) + s) / (1 + s)) ** power
    alpha_bars = f_t / f_t[0]
    betas = torch.cat([torch.zeros([1], dtype=dtype), 1 - (alpha_bars[1:] / alpha_bars[:-1])], dim=0)
    betas = betas.clamp_max(clipmax)
    sigmas = torch.sqrt(betas[1:] * ((1 - alpha_bars[:-1]) / (1 - alpha_bars[1:])))
    sigmas = torch.cat([torch.zeros([1], dtype=dtype), sigmas], dim=0)
    return alpha_bars, betas, sigmas


def vp_reverse_given_x0(alpha_bars, betas, xt, pred_x0, t, z):
    """Posterior step from predicted x0; adds ``variance * z`` (not sqrt) and
    z = 0 when t <= 1, exactly as :185-199.  ``t`` is a 1-element long tensor,
    ``z`` the injected randn_like(xt) (drawn by the reference even when masked).
    """
    denom = 1 - alpha_bars[t]
    alpha_t = 1 - betas[t]
    x0_term = torch.sqrt(alpha_bars[t - 1]) * betas[t] * pred_x0
    xt_term = torch.sqrt(alpha_t) * (1 - alpha_bars[t - 1]) * xt
    mean = (x0_term + xt_term) / denom
    variance = (1 - alpha_bars[t - 1]) * betas[t] / denom
    z = torch.where((t > 1)[:, None].expand_as(xt), z, torch.zeros_like(xt))
    return mean + variance * z


# ----------------------------------------------------------------------------
# D3PM, absorbing ("mask") chain   diffusion/d3pm.py
# ----------------------------------------------------------------------------
D3PM_EPS = 1e-6  # d3pm.py:23
MASK_PROB = 0.02  # d3pm.py:34


def d3pm_buffers(T: int, S: int, dtype=None):
    """(q_one_step_transposed [T,S,S], q_mats [T,S,S]) for forward_type="mask".

    d3pm.py:33-54: every one-step matrix is (1-p) I with column S-1 set to p and
    the mask row absorbing; q_mats are the running left-to-right products.  All
    one-step matrices are identical (beta_t is unused by the mask branch).
    """
    one = torch.zeros(S, S, dtype=dtype)
    one[:, -1] = MASK_PROB
    one.diagonal().fill_(1 - MASK_PROB)
    one[-1, -1] = 1
    q_one = one.unsqueeze(0).repeat(T, 1, 1)
    mats = [one]
    cur = one
    for _ in range(1, T):
        cur = cur @ one
        mats.append(cur)
    return q_one.transpose(1, 2).contiguous(), torch.stack(mats, 0)


def d3pm_q_posterior_logits(q_one_step_transposed, q_mats, x0_logits, x_t, t):
    """d3pm.py:74-110 with float x_0 logits (the sampler always passes logits).

    fact1 = Q_t^T[x_t, :], fact2 = softmax(x0_logits) @ Qbar_{t-1} with the
    reference's index ``t-2`` (wraps to -1 at t=1, then masked by the t==1
    branch that returns the raw logits).
    """
    fact1 = q_one_step_transposed[t - 1, x_t, :]
    soft = torch.softmax(x0_logits, dim=-1)
    fact2 = torch.einsum("bc,bcd->bd", soft, q_mats[t - 2])
    out = torch.log(fact1 + D3PM_EPS) + torch.log(fact2 + D3PM_EPS)
    return torch.where((t == 1).view(-1, 1), x0_logits, out)


def d3pm_reverse(q_one_step_transposed, q_mats, x_t, x0_logits, t, u):
    """Gumbel-argmax reverse step, d3pm.py:198-215.  ``u`` is the injected
    uniform noise [N,S] (torch.rand at :206)."""
    post = d3pm_q_posterior_logits(q_one_step_transposed, q_mats, x0_logits, x_t, t)
    u = torch.clip(u, D3PM_EPS, 1.0)
    scale = 0.2 + (t != 1).float().view(-1, 1) * 0.8  # float32 on purpose (d3pm.py:209)
    gumbel = -torch.log(-torch.log(u))
    return torch.argmax(post + gumbel * scale, dim=-1)


def sample_monoclinic_angles(rng: np.random.RandomState, B: int) -> np.ndarray:
    """[90, U(90,180), 90] per crystal, in DEGREES, as
    diffusion/diffusion_helpers.py:752-755 does for "monoclinic" (the sampler
    then consumes them as radians, diffusion_loss.py:294-296 + lattice_helpers.py:76)."""
    out = np.empty((B, 3), dtype=np.float64)
    for i in range(B):
        out[i] = (90, rng.uniform(90, 180), 90)
    return out

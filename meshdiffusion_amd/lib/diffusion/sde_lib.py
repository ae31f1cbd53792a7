"""VP SDE tables -- mirror of the reference's lib/diffusion/sde_lib.py (VPSDE :176-232).

The reference hard-codes `.cuda()`; here the tables live on `device` (default: cuda if
present).  Table construction uses the same float32 torch ops so the values are identical.
"""
import numpy as np
import torch


def _default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class SDE:
    def __init__(self, N):
        self.N = N


class VPSDE(SDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000, device=None):
        super().__init__(N)
        dev = device if device is not None else _default_device()
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N).to(dev)
        self.alphas = 1.0 - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alphas_cumprod_ext = torch.cat(
            [torch.tensor([1.0 - 1e-4]).to(dev), torch.cumprod(self.alphas, dim=0)], dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        drift = -0.5 * beta_t[:, None, None, None, None] * x
        return drift, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_mean_coeff[:, None, None, None, None]) * x
        std = torch.sqrt(1.0 - torch.exp(2.0 * log_mean_coeff))
        return mean, std

    def prior_sampling(self, shape):
        # CPU generator, like the reference (sde_lib.py:216-217): keeps the prior draw
        # identical between a CPU reference run and a GPU run under the same seed.
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3, 4)) / 2.0

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[timestep]
        alpha = self.alphas.to(x.device)[timestep]
        f = torch.sqrt(alpha)[:, None, None, None, None] * x - x
        return f, torch.sqrt(beta)

"""Drop-in for protlig_dd/processing/noise_lib.py. sigma(t), dsigma(t) are O(B) scalar schedules evaluated with a
handful of elementwise torch ops on the [B] time vector (host-side glue; no kernel worth writing)."""
import abc
import math

import torch
import torch.nn as nn


def get_noise(config):
    """noise_lib.py:7-17"""
    if config.noise.type == "geometric":
        return GeometricNoise(config.noise.sigma_min, config.noise.sigma_max)
    elif config.noise.type == "loglinear":
        return LogLinearNoise(eps=config.noise.eps)
    elif config.noise.type == "cosine":
        return CosineNoise(sigma_min=config.noise.sigma_min, sigma_max=config.noise.sigma_max)
    raise ValueError(f"{config.noise.type} is not a valid noise")


class Noise(abc.ABC, nn.Module):
    """noise_lib.py:20-42"""

    def forward(self, t):
        return self.total_noise(t), self.rate_noise(t)

    @abc.abstractmethod
    def rate_noise(self, t):
        pass

    @abc.abstractmethod
    def total_noise(self, t):
        pass


class GeometricNoise(Noise):
    """noise_lib.py:45-57 (sigmas stays a plain attribute, moved lazily to t's device)."""

    def __init__(self, sigma_min=1e-3, sigma_max=1, learnable=False):
        super().__init__()
        self.sigmas = 1.0 * torch.tensor([sigma_min, sigma_max])
        if learnable:
            self.sigmas = nn.Parameter(self.sigmas)
        self.empty = nn.Parameter(torch.tensor(0.0))

    def _s(self, t):
        s = self.sigmas.to(t.device)
        return s[0], s[1]

    def rate_noise(self, t):
        s0, s1 = self._s(t)
        return s0 ** (1 - t) * s1 ** t * (s1.log() - s0.log())

    def total_noise(self, t):
        s0, s1 = self._s(t)
        return s0 ** (1 - t) * s1 ** t


class LogLinearNoise(Noise):
    """noise_lib.py:60-76"""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps
        self.empty = nn.Parameter(torch.tensor(0.0))

    def rate_noise(self, t):
        return (1 - self.eps) / (1 - (1 - self.eps) * t)

    def total_noise(self, t):
        return -torch.log1p(-(1 - self.eps) * t)


class CosineNoise(Noise):
    """noise_lib.py:79-94"""

    def __init__(self, sigma_min=1e-3, sigma_max=1.0, T=1.0, learnable=False):
        super().__init__()
        sigmas = torch.tensor([sigma_min, sigma_max], dtype=torch.float32)
        self.sigmas = nn.Parameter(sigmas) if learnable else sigmas
        self.T = T

    def total_noise(self, t):
        s = self.sigmas.to(t.device)
        return s[1] + 0.5 * (s[0] - s[1]) * (1 + torch.cos(math.pi * t / self.T))

    def rate_noise(self, t):
        s = self.sigmas.to(t.device)
        return (math.pi / (2 * self.T)) * (s[1] - s[0]) * torch.sin(math.pi * t / self.T)

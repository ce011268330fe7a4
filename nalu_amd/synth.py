"""Seeded synthetic fields of SURVEY.md section 8(d), shared by tests, smoke and bench."""
import numpy as np


def quadratic_coeffs(seed=0):
    rng = np.random.RandomState(seed)
    a = rng.uniform(-1, 1)
    b = rng.uniform(-1, 1, 3)
    H = rng.uniform(-1, 1, (3, 3))
    return a, b, 0.5 * (H + H.T)


def quadratic(xyz, seed=0):
    """a + b.x + 1/2 x^T H x with U(-1,1) coefficients (unit_tests/UnitTestUtils.C:549-588
    style); xyz is (n, 3)."""
    a, b, H = quadratic_coeffs(seed)
    return a + xyz @ b + 0.5 * np.einsum("ni,ij,nj->n", xyz, H, xyz)


def quadratic_grad(xyz, seed=0):
    """exact gradient of `quadratic`, (n, 3)"""
    _, b, H = quadratic_coeffs(seed)
    return b + xyz @ H


def velocity(xyz):
    x, y = xyz[:, 0], xyz[:, 1]
    u = np.sin(np.pi * x) * np.cos(np.pi * y)
    v = -np.cos(np.pi * x) * np.sin(np.pi * y)
    return np.stack([u, v, np.zeros_like(u)], axis=1)


def density(xyz, variable=True):
    return 1.0 + 0.1 * np.sin(2 * np.pi * xyz[:, 0]) if variable else np.ones(len(xyz))

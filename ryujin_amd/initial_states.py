"""Initial/Dirichlet state recipes used as synthetic inputs (formulas restated from
source/euler/initial_state_{uniform,radial_contrast,isentropic_vortex}.h and
source/euler/hyperbolic_system.h:1255-1272 `from_primitive_state`)."""
from __future__ import annotations

import numpy as np

from . import capi


def euler_from_primitive(rho, vel, p, gamma=1.4):
    """vel: [..., dim]. Returns conserved [..., dim+2] = (rho, rho v, p/(gamma-1) + rho |v|^2/2)."""
    rho = np.asarray(rho, dtype=np.float64)
    vel = np.asarray(vel, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    U = np.empty(vel.shape[:-1] + (vel.shape[-1] + 2,), dtype=np.float64)
    U[..., 0] = rho
    U[..., 1:-1] = rho[..., None] * vel
    U[..., -1] = p / (gamma - 1.0) + 0.5 * rho * np.sum(vel * vel, axis=-1)
    return U


def euler_uniform(positions, rho=1.4, u=3.0, p=1.0, gamma=1.4, direction=None):
    """initial_state_uniform.h:36-50: primitive (rho,u,p) along `direction` (default +x)."""
    n, dim = positions.shape
    vel = np.zeros((n, dim))
    d = np.zeros(dim)
    d[0] = 1.0
    if direction is not None:
        d = np.asarray(direction, dtype=np.float64)
        d = d / np.linalg.norm(d)
    vel[:] = u * d
    return euler_from_primitive(np.full(n, rho), vel, np.full(n, p), gamma)


def euler_radial_contrast(positions, inner=(1.0, 0.0, 100.0), outer=(1.0, 0.0, 0.1), radius=0.1,
                          gamma=1.4, center=None):
    """initial_state_radial_contrast.h:29-62 (|x| <= radius -> inner state)."""
    n, dim = positions.shape
    x = positions if center is None else positions - np.asarray(center)
    r = np.linalg.norm(x, axis=1)
    inside = r <= radius
    rho = np.where(inside, inner[0], outer[0])
    p = np.where(inside, inner[2], outer[2])
    vel = np.zeros((n, dim))
    return euler_from_primitive(rho, vel, p, gamma)


def euler_isentropic_vortex(positions, t, mach=1.0, beta=5.0, gamma=1.4, direction=(1.0, 1.0),
                            position=(-1.0, -1.0)):
    """initial_state_isentropic_vortex.h:54-92 composed with the affine transform of
    initial_values.template.h:66-148 (translate by `position`, rotate onto `direction`)."""
    n, dim = positions.shape
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    nx, ny = d[0], d[1]
    x = positions[:, 0] - position[0]
    y = positions[:, 1] - position[1]
    # affine_transform: rotate the point back into the vortex frame
    xr = nx * x + ny * y
    yr = -ny * x + nx * y
    xb = xr - mach * t
    yb = yr
    r2 = xb * xb + yb * yb
    factor = beta / (2.0 * np.pi) * np.exp(0.5 - 0.5 * r2)
    T = 1.0 - (gamma - 1.0) / (2.0 * gamma) * factor * factor
    u = mach - factor * yb
    v = factor * xb
    rho = T ** (1.0 / (gamma - 1.0))
    p = rho ** gamma
    E = p / (gamma - 1.0) + 0.5 * rho * (u * u + v * v)
    # affine_transform_vector: rotate the momentum into the lab frame
    mx, my = rho * u, rho * v
    U = np.zeros((n, dim + 2))
    U[:, 0] = rho
    U[:, 1] = nx * mx - ny * my
    U[:, 2] = ny * mx + nx * my
    U[:, -1] = E
    return U


def sw_circular_dam_break(positions, h_inner=2.5, h_outer=0.5, radius=2.5):
    """source/shallow_water/initial_state_circular_dam_break.h:48-54 (compares |x|^2 with `radius`,
    sic): h = h_inner where |x|^2 <= radius, else h_outer; zero momentum. Returns (h, q) states."""
    n, dim = positions.shape
    r2 = np.sum(positions * positions, axis=1)
    U = np.zeros((n, dim + 1))
    U[:, 0] = np.where(r2 <= radius, h_inner, h_outer)
    return U


def aeos_specific_internal_energy(params, rho, p):
    """EquationOfState::specific_internal_energy(rho, p) of the closed-form equations of state
    (source/euler_aeos/equation_of_state_*.h), vectorised; `params` is a capi.Params."""
    rho = np.asarray(rho, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    g = params.gamma
    if params.eos == capi.EOS_POLYTROPIC_GAS:
        return p / (rho * (g - 1.0))
    if params.eos == capi.EOS_NOBLE_ABEL_STIFFENED_GAS:
        b, q, pinf = params.eos_covolume_b, params.eos_q, params.eos_pinf
        return q + (p + g * pinf) * (1.0 - b * rho) / (rho * (g - 1.0))
    if params.eos == capi.EOS_VAN_DER_WAALS:
        a, b = params.eos_vdw_a, params.eos_covolume_b
        return (p + a * rho * rho) * (1.0 - b * rho) / (rho * (g - 1.0)) - a * rho
    ratio = rho / params.jwl_rho_0
    first = params.jwl_A * (1.0 - params.jwl_omega / params.jwl_R1 * ratio) * np.exp(-params.jwl_R1 / ratio)
    second = params.jwl_B * (1.0 - params.jwl_omega / params.jwl_R2 * ratio) * np.exp(-params.jwl_R2 / ratio)
    return (p - first - second) / (rho * params.jwl_omega)


def aeos_from_primitive(params, rho, vel, p):
    """HyperbolicSystemView::from_initial_state (source/euler_aeos/hyperbolic_system.h:1470-1512):
    primitive (rho, v, p) -> conserved state with the selected equation of state."""
    rho = np.asarray(rho, dtype=np.float64)
    vel = np.asarray(vel, dtype=np.float64)
    e = aeos_specific_internal_energy(params, rho, p)
    U = np.empty(vel.shape[:-1] + (vel.shape[-1] + 2,), dtype=np.float64)
    U[..., 0] = rho
    U[..., 1:-1] = rho[..., None] * vel
    U[..., -1] = rho * e + 0.5 * rho * np.sum(vel * vel, axis=-1)
    return U

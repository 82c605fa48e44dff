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


# ---------------------------------------------------------------------------
# Analytic solutions of the reference's verification configurations (1-D ones take x relative to
# the `position` of subsection "E - InitialValues", direction +1)

def euler_leblanc(positions, t, position=0.0):
    """source/euler/initial_state_leblanc.h:63-120: the Le Blanc shock tube (gamma = 5/3), conserved
    1-D states (rho, m, E) of the exact Riemann fan at time t."""
    x = np.asarray(positions, dtype=np.float64)[:, 0] - position
    rarefaction_speed = 0.49578489518897934
    contact_velocity = 0.62183867139173454
    right_shock_speed = 0.82911836253346982
    pre_contact_density = 5.4079335349316249e-02
    post_contact_density = 3.9999980604299963e-03
    contact_pressure = 0.51557792765096996e-03
    rho = np.full_like(x, 1.0e-3)
    u = np.zeros_like(x)
    p = np.full_like(x, 2.0 / 3.0 * 1.0e-10)
    with np.errstate(divide="ignore", invalid="ignore"):
        chi = x / t
    left = x <= -1.0 / 3.0 * t
    fan = ~left & (x < rarefaction_speed * t)
    pre = ~left & ~fan & (x < contact_velocity * t)
    post = ~left & ~fan & ~pre & (x < right_shock_speed * t)
    rho[left], u[left], p[left] = 1.0, 0.0, 2.0 / 3.0 * 1.0e-1
    rho[fan] = np.power(0.75 - 0.75 * chi[fan], 3.0)
    u[fan] = 0.75 * (1.0 / 3.0 + chi[fan])
    p[fan] = (1.0 / 15.0) * np.power(0.75 - 0.75 * chi[fan], 5.0)
    rho[pre], u[pre], p[pre] = pre_contact_density, contact_velocity, contact_pressure
    rho[post], u[post], p[post] = post_contact_density, contact_velocity, contact_pressure
    U = np.empty((x.size, 3))
    U[:, 0] = rho
    U[:, 1] = rho * u
    U[:, 2] = p / (5.0 / 3.0 - 1.0) + 0.5 * rho * u * u
    return U


def euler_rarefaction(positions, t, gamma=1.4, position=0.0):
    """source/euler/initial_state_rarefaction.h:40-160: a single 1-rarefaction that has been running
    for t_0 = 0.2 / (u_R - u_L) already; conserved 1-D states at time t_0 + t."""
    x = np.asarray(positions, dtype=np.float64)[:, 0] - position
    rho_l, p_l = 3.0, 1.0
    c_l = np.sqrt(gamma * p_l / rho_l)
    u_l = c_l
    rho_r = 0.5
    p_r = np.power(rho_r / rho_l, gamma) * p_l
    c_r = np.sqrt(gamma * p_r / rho_r)
    u_r = u_l + 2.0 * (c_l - c_r) / (gamma - 1.0)
    k1 = 2.0 / (gamma + 1.0)
    k2 = (gamma - 1.0) / ((gamma + 1.0) * c_l)
    density_exponent = 2.0 / (gamma - 1.0)
    k3 = c_l + ((gamma - 1.0) / 2.0) * u_l
    pressure_exponent = 2.0 * gamma / (gamma - 1.0)
    tt = 0.2 / (u_r - u_l) + t
    chi = x / tt
    left = x <= tt * (u_l - c_l)
    fan = ~left & (x <= tt * (u_r - c_r))
    rho = np.full_like(x, rho_r)
    u = np.full_like(x, u_r)
    p = np.full_like(x, p_r)
    rho[left], u[left], p[left] = rho_l, u_l, p_l
    base = k1 + k2 * (u_l - chi[fan])
    rho[fan] = rho_l * np.power(base, density_exponent)
    u[fan] = k1 * (k3 + chi[fan])
    p[fan] = p_l * np.power(base, pressure_exponent)
    U = np.empty((x.size, 3))
    U[:, 0] = rho
    U[:, 1] = rho * u
    U[:, 2] = p / (gamma - 1.0) + 0.5 * rho * u * u
    return U


def sw_paraboloid_1d(positions, t, gravity=9.81, manning=0.0, free_surface_radius=3000.0,
                     water_height=10.0, length=10000.0, speed=2.0):
    """source/shallow_water/initial_state_paraboloid.h:66-101 (dim 1): the oscillating lake over a
    parabolic bathymetry with wetting and drying; returns ((h, q) states, bathymetry)."""
    x = np.asarray(positions, dtype=np.float64)[:, 0]
    a, h0, B, g, k = free_surface_radius, water_height, speed, gravity, manning
    z = h0 / (a * a) * np.power(x - 0.5 * length, 2)
    p = np.sqrt(8.0 * g * h0) / a
    s = np.sqrt(p * p - k * k) / 2.0
    term1 = (a * a * B * B) / (8.0 * g * g * h0) * np.exp(-k * t)
    term1 *= (1.0 / 4.0 * k * k - s * s) * np.cos(2.0 * s * t) - s * k * np.sin(2.0 * s * t)
    term2 = -(B * B / (4.0 * g)) * np.exp(-k * t)
    term3 = -(B / g) * np.exp(-1.0 / 2.0 * k * t)
    term3 = term3 * (s * np.cos(s * t) + 1.0 / 2.0 * k * np.sin(s * t)) * (x - 1.0 / 2.0 * length)
    htilde = h0 - z
    htilde = htilde + (term1 + term2 + term3)
    h = np.maximum(htilde, 0.0)
    v = B * np.exp(-1.0 / 2.0 * k * t) * np.sin(s * t)
    return np.column_stack([h, h * v]), z


def sw_ritter_dam_break(positions, t, gravity=9.81, time_initial=0.1, left_depth=0.005, position=0.0):
    """source/shallow_water/initial_state_ritter_dam_break.h:58-80: dam break over a dry bed."""
    x = np.asarray(positions, dtype=np.float64)[:, 0] - position
    g = gravity
    aL = np.sqrt(g * left_depth)
    xA = -(t + time_initial) * aL
    xB = 2.0 * (t + time_initial) * aL
    tmp = aL - x / (2.0 * (t + time_initial))
    h_exp = 4.0 / (9.0 * g) * tmp * tmp
    v_exp = 2.0 / 3.0 * (x / (t + time_initial) + aL)
    U = np.zeros((x.size, 2))
    left = x <= xA
    fan = ~left & (x <= xB)
    U[left, 0] = left_depth
    U[fan, 0] = h_exp[fan]
    U[fan, 1] = h_exp[fan] * v_exp[fan]
    return U


def sw_smooth_vortex(positions, t, gravity=9.81, reference_depth=1.0, mach=2.0, beta=0.1,
                     direction=(1.0, 0.0), position=(0.0, 0.0)):
    """source/shallow_water/initial_state_smooth_vortex.h:55-85 (no bathymetry) composed with the affine
    transform of initial_values.template.h:66-148."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    nx, ny = d
    x = positions[:, 0] - position[0]
    y = positions[:, 1] - position[1]
    xr = nx * x + ny * y
    yr = -ny * x + nx * y
    xb = xr - mach * t
    yb = yr
    r2 = xb * xb + yb * yb
    factor = beta / (2.0 * np.pi) * np.exp(0.5 - 0.5 * r2)
    h = reference_depth - 1.0 / (2.0 * gravity) * factor * factor
    u = mach - factor * yb
    v = factor * xb
    mx, my = h * u, h * v
    return np.column_stack([h, nx * mx - ny * my, ny * mx + nx * my])


def sw_sloping_friction(positions, manning=0.01, ramp_slope=1.0, initial_discharge=0.1):
    """source/shallow_water/initial_state_sloping_friction.h:50-85: uniform flow down an incline in
    balance with Manning friction; returns ((h, q) states, bathymetry)."""
    x = np.asarray(positions, dtype=np.float64)[:, 0]
    exponent = 1.0 / (2.0 + 4.0 / 3.0)
    profile = manning * manning * initial_discharge * initial_discharge / ramp_slope
    h = np.power(profile, exponent)
    U = np.empty((x.size, 2))
    U[:, 0] = h
    U[:, 1] = initial_discharge
    return U, -ramp_slope * x

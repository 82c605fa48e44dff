// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C ABI of the CPU restatement: the same call surface as include/ryujin_hip.h with
// the prefix ryujin_oracle_ (so the parity tests drive both through one wrapper), plus
// function-level entry points used to pin the restatement against the reference's
// golden outputs. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg load this library.

#include <cstring>
#include <memory>
#include <string>

#include <xmmintrin.h>
#include <pmmintrin.h>

#include "aeos_module.hpp"
#include "scalar_module.hpp"
#include "hyperbolic_module.hpp"
#include "shallow_water.hpp"

using namespace oracle;

namespace
{
  thread_local std::string g_error;

  struct Ctx {
    std::unique_ptr<ModuleBase> m;
  };

  Ctx *C(void *p) { return static_cast<Ctx *>(p); }

  template <typename F>
  int guarded(F &&f)
  {
    try {
      return f();
    } catch (const std::exception &e) {
      g_error = e.what();
      return RYUJIN_ERR_ARG;
    }
  }
} // namespace

extern "C" {

void ryujin_oracle_default_params(ryujin_hip_params *p, int equation, int dim)
{
  std::memset(p, 0, sizeof(*p));
  p->equation = equation;
  p->dim = dim;
  p->gamma = 7. / 5.;
  p->reference_density = 1.;
  p->vacuum_state_relaxation_small = 1.e2;
  p->vacuum_state_relaxation_large = 1.e4;
  p->gravity = 9.81;
  p->manning_friction_coefficient = 0.;
  p->reference_water_depth = 1.;
  p->dry_state_relaxation_factor = 2.e-1;
  p->dry_state_relaxation_small = 1.e2;
  p->dry_state_relaxation_large = 1.e4;
  p->cfl = 0.2;
  p->id_violation_strategy = RYUJIN_IDV_WARN;
  p->indicator_evc_factor = 1.;
  p->limiter_iterations = 2;
  p->limiter_newton_tolerance = 1.e-10;
  p->limiter_newton_max_iterations = 2;
  p->limiter_relaxation_factor = 1.;
  p->limiter_limit_on_kinetic_energy = 0;
  p->limiter_limit_on_square_velocity = 1;
  p->riemann_newton_max_iterations = 0;
  p->riemann_newton_tolerance = 1.e-10;
  p->eos = RYUJIN_EOS_POLYTROPIC_GAS;
  p->compute_strict_bounds = 1;
  p->eos_covolume_b = 0.;
  p->eos_q = 0.;
  p->eos_pinf = 0.;
  p->eos_vdw_a = 0.;
  p->eos_gas_constant_R = 287.052874;
  p->jwl_A = 6.3207e13;
  p->jwl_B = -4.472e9;
  p->jwl_R1 = 11.3;
  p->jwl_R2 = 1.13;
  p->jwl_omega = 0.8938;
  p->jwl_rho_0 = 1895.;
  p->jwl_q_0 = 0.;
  p->jwl_cv = 2487. / 1895.;
  p->sc_flux = RYUJIN_FLUX_BURGERS;
  p->sc_flux_polynomial[0][2] = 0.5; /* "0.5*u*u", the default expression of flux_function.h:32 */
  p->sc_derivative_approximation_delta = 1.e-10;
  p->sc_use_greedy_wavespeed = 0;
  p->sc_use_averaged_entropy = 0;
  p->sc_random_entropies = 0;
}

/* FTZ/DAZ as the reference sets in main (source/main.cc:26-36) */
void ryujin_oracle_set_flush_denormals(int on)
{
  _MM_SET_FLUSH_ZERO_MODE(on ? _MM_FLUSH_ZERO_ON : _MM_FLUSH_ZERO_OFF);
  _MM_SET_DENORMALS_ZERO_MODE(on ? _MM_DENORMALS_ZERO_ON : _MM_DENORMALS_ZERO_OFF);
}

int ryujin_oracle_create(void **ctx, const ryujin_hip_offline *offline,
                         const ryujin_hip_params *params, void * /*comm*/, int /*device*/)
{
  return guarded([&]() {
    auto c = std::make_unique<Ctx>();
    const int dim = params->dim;
    if (params->equation == RYUJIN_EQ_EULER) {
      if (dim == 1)
        c->m = std::make_unique<EulerModule<1>>(*offline, *params);
      else if (dim == 2)
        c->m = std::make_unique<EulerModule<2>>(*offline, *params);
      else if (dim == 3)
        c->m = std::make_unique<EulerModule<3>>(*offline, *params);
    } else if (params->equation == RYUJIN_EQ_EULER_AEOS) {
      if (dim == 1)
        c->m = std::make_unique<EulerAeosModule<1>>(*offline, *params);
      else if (dim == 2)
        c->m = std::make_unique<EulerAeosModule<2>>(*offline, *params);
      else if (dim == 3)
        c->m = std::make_unique<EulerAeosModule<3>>(*offline, *params);
    } else if (params->equation == RYUJIN_EQ_SCALAR_CONSERVATION) {
      if (dim == 1)
        c->m = std::make_unique<ScalarConservationModule<1>>(*offline, *params);
      else if (dim == 2)
        c->m = std::make_unique<ScalarConservationModule<2>>(*offline, *params);
      else if (dim == 3)
        c->m = std::make_unique<ScalarConservationModule<3>>(*offline, *params);
    } else if (params->equation == RYUJIN_EQ_SHALLOW_WATER) {
      if (dim == 1)
        c->m = std::make_unique<ShallowWaterModule<1>>(*offline, *params);
      else if (dim == 2)
        c->m = std::make_unique<ShallowWaterModule<2>>(*offline, *params);
    }
    if (!c->m) {
      g_error = "unsupported equation/dimension";
      return RYUJIN_ERR_UNSUPPORTED;
    }
    *ctx = c.release();
    return RYUJIN_OK;
  });
}

void ryujin_oracle_destroy(void *ctx)
{
  delete C(ctx);
}

void ryujin_oracle_set_exchange(void *ctx, exchange_fn fn, void *user)
{
  C(ctx)->m->exchange = fn;
  C(ctx)->m->exchange_user = user;
}

void ryujin_oracle_set_expensive_bounds_check(void *ctx, int on)
{
  C(ctx)->m->expensive_bounds_check = on != 0;
}

int ryujin_oracle_state_alloc(void *ctx, int *handle)
{
  return guarded([&]() {
    *handle = C(ctx)->m->state_alloc();
    return RYUJIN_OK;
  });
}

int ryujin_oracle_state_free(void *ctx, int handle)
{
  return guarded([&]() {
    C(ctx)->m->state_free(handle);
    return RYUJIN_OK;
  });
}

int ryujin_oracle_state_upload(void *ctx, int handle, const double *U)
{
  return guarded([&]() {
    auto &m = *C(ctx)->m;
    std::memcpy(m.state_U(handle), U, sizeof(double) * (size_t)m.n_relevant * m.k());
    return RYUJIN_OK;
  });
}

int ryujin_oracle_state_download(void *ctx, int handle, double *U)
{
  return guarded([&]() {
    auto &m = *C(ctx)->m;
    std::memcpy(U, m.state_U(handle), sizeof(double) * (size_t)m.n_relevant * m.k());
    return RYUJIN_OK;
  });
}

int ryujin_oracle_state_download_precomputed(void *ctx, int handle, double *prec)
{
  return guarded([&]() {
    auto &m = *C(ctx)->m;
    std::memcpy(prec, m.state_prec(handle), sizeof(double) * (size_t)m.n_relevant * m.n_prec());
    return RYUJIN_OK;
  });
}

int ryujin_oracle_prepare_state_vector(void *ctx, int handle, double t, const double *dirichlet)
{
  return guarded([&]() {
    C(ctx)->m->prepare_state_vector(handle, t, dirichlet);
    return RYUJIN_OK;
  });
}

int ryujin_oracle_step(void *ctx, int h_old, int stages, const int *h_stage, const double *w,
                       int h_new, double tau_in, double tau_max_in, double *tau_out)
{
  return guarded([&]() {
    if (stages < 0 || stages > 4)
      return RYUJIN_ERR_ARG;
    return C(ctx)->m->step(h_old, stages, h_stage, w, h_new, tau_in, tau_max_in, tau_out);
  });
}

int ryujin_oracle_sadd(void *ctx, int h_dst, double s, double b, int h_src)
{
  return guarded([&]() {
    C(ctx)->m->sadd(h_dst, s, b, h_src);
    return RYUJIN_OK;
  });
}

int ryujin_oracle_set_cfl(void *ctx, double cfl)
{
  C(ctx)->m->params.cfl = cfl;
  return RYUJIN_OK;
}

int ryujin_oracle_get_cfl(void *ctx, double *cfl)
{
  *cfl = C(ctx)->m->params.cfl;
  return RYUJIN_OK;
}

int ryujin_oracle_set_id_violation_strategy(void *ctx, int s)
{
  C(ctx)->m->params.id_violation_strategy = s;
  return RYUJIN_OK;
}

int ryujin_oracle_get_alpha(void *ctx, double *alpha)
{
  auto &m = *C(ctx)->m;
  std::memcpy(alpha, m.alpha.data(), sizeof(double) * m.n_relevant);
  return RYUJIN_OK;
}

int ryujin_oracle_get_counters(void *ctx, unsigned *n_restarts, unsigned *n_warnings)
{
  *n_restarts = C(ctx)->m->n_restarts;
  *n_warnings = C(ctx)->m->n_warnings;
  return RYUJIN_OK;
}

int ryujin_oracle_debug_fetch(void *ctx, int what, double *out, size_t n)
{
  return guarded([&]() { return C(ctx)->m->debug_fetch(what, out, n); });
}

const char *ryujin_oracle_last_error(void)
{
  return g_error.c_str();
}

/* ------------------------------------------------------------------------
 * Function-level entry points (golden-vector tests)
 * ---------------------------------------------------------------------- */

/* Euler RiemannSolver::compute(riemann_data_i, riemann_data_j) with trace.
 * out[0..4] = p_star_two_rarefaction, p_star_failsafe, p*_tilde, phi(p*_tilde), lambda_max
 * out[5..8] = p_1, p_2, gap, lambda_max at start of the Newton iteration
 * out[9] = converged_after (-1 if never), out[10] = number of recorded iterations
 * iters[8*n..] = phi_p_1, phi_p_2, dphi_p_1, dphi_p_2, p_1, p_2, gap, lambda_max */
int ryujin_oracle_euler_riemann(const ryujin_hip_params *p, const double rd_i[4],
                                const double rd_j[4], double out[11], double *iters,
                                int max_iters)
{
  euler::RiemannSolver rs(*p);
  euler::RiemannTrace tr;
  const euler::primitive_type a{{rd_i[0], rd_i[1], rd_i[2], rd_i[3]}};
  const euler::primitive_type b{{rd_j[0], rd_j[1], rd_j[2], rd_j[3]}};
  rs.compute(a, b, &tr);
  out[0] = tr.p_star_two_rarefaction;
  out[1] = tr.p_star_failsafe;
  out[2] = tr.p_star_tilde;
  out[3] = tr.phi_p_star_tilde;
  out[4] = tr.lambda_max;
  out[5] = tr.p_1_start;
  out[6] = tr.p_2_start;
  out[7] = tr.gap_start;
  out[8] = tr.lambda_max_start;
  out[9] = tr.converged_after;
  out[10] = (double)tr.iterations.size();
  for (size_t n = 0; n < tr.iterations.size() && (int)n < max_iters; ++n) {
    const auto &it = tr.iterations[n];
    const double v[8] = {it.phi_p_1, it.phi_p_2, it.dphi_p_1, it.dphi_p_2,
                         it.p_1,     it.p_2,     it.gap,      it.lambda_max};
    std::memcpy(iters + 8 * n, v, sizeof(v));
  }
  return RYUJIN_OK;
}

/* lambda_max from conserved states and a unit normal (RiemannSolver::compute(U_i,U_j,i,js,n_ij)) */
double ryujin_oracle_euler_lambda_max(const ryujin_hip_params *p, const double *U_i,
                                      const double *U_j, const double *n_ij)
{
  euler::RiemannSolver rs(*p);
  const int dim = p->dim;
  if (dim == 1) {
    return rs.compute<1>({{U_i[0], U_i[1], U_i[2]}}, {{U_j[0], U_j[1], U_j[2]}}, {{n_ij[0]}});
  } else if (dim == 2) {
    return rs.compute<2>({{U_i[0], U_i[1], U_i[2], U_i[3]}}, {{U_j[0], U_j[1], U_j[2], U_j[3]}},
                         {{n_ij[0], n_ij[1]}});
  }
  return rs.compute<3>({{U_i[0], U_i[1], U_i[2], U_i[3], U_i[4]}},
                       {{U_j[0], U_j[1], U_j[2], U_j[3], U_j[4]}}, {{n_ij[0], n_ij[1], n_ij[2]}});
}

/* d_ij = |c_ij| lambda_max(U_i, U_j, c_ij / |c_ij|) for n independent pairs (hyperbolic_module.template.h:402-406;
 * dealii::Tensor / scalar multiplies by the inverse) */
int ryujin_oracle_euler_dij_batch(const ryujin_hip_params *p, size_t n, const double *U_i, const double *U_j,
                                  const double *c, double *out)
{
  const int dim = p->dim, k = dim + 2;
#pragma omp parallel for schedule(static)
  for (size_t q = 0; q < n; ++q) {
    double norm2 = 0.;
    for (int d = 0; d < dim; ++d)
      norm2 += c[q * dim + d] * c[q * dim + d];
    const double norm = std::sqrt(norm2), inverse = 1. / norm;
    double nrm[3] = {0., 0., 0.};
    for (int d = 0; d < dim; ++d)
      nrm[d] = c[q * dim + d] * inverse;
    out[q] = norm * ryujin_oracle_euler_lambda_max(p, U_i + q * k, U_j + q * k, nrm);
  }
  return RYUJIN_OK;
}

/* Euler Limiter::limit in 1-D (the reference's unit test is dim = 1).
 * out[0]=l, out[1]=success, out[2]=t_l start, out[3]=t_r start, out[4..7]= violation flags
 * (density low, density high, entropy low, entropy high), out[8] = n recorded iterations
 * iters[7*n..] = kind, psi_l, psi_r, dpsi_l, dpsi_r, t_l, t_r */
int ryujin_oracle_euler_limit_1d(const ryujin_hip_params *p, int expensive_bounds_check,
                                 const double bounds[3], const double U[3], const double P[3],
                                 double out[9], double *iters, int max_iters)
{
  euler::View<1> view(*p);
  euler::Limiter<1> lim(view, *p);
  lim.expensive_bounds_check = expensive_bounds_check != 0;
  euler::LimiterTrace tr;
  const auto [l, success] = lim.limit({{bounds[0], bounds[1], bounds[2]}}, {{U[0], U[1], U[2]}},
                                      {{P[0], P[1], P[2]}}, 0., 1., &tr);
  out[0] = l;
  out[1] = success ? 1. : 0.;
  out[2] = tr.t_l_start;
  out[3] = tr.t_r_start;
  out[4] = tr.density_violation_low_order;
  out[5] = tr.density_violation_high_order;
  out[6] = tr.entropy_violation_low_order;
  out[7] = tr.entropy_violation_high_order;
  out[8] = (double)tr.iterations.size();
  for (size_t n = 0; n < tr.iterations.size() && (int)n < max_iters; ++n) {
    const auto &it = tr.iterations[n];
    const double v[7] = {(double)it.kind, it.psi_l, it.psi_r, it.dpsi_l, it.dpsi_r, it.t_l, it.t_r};
    std::memcpy(iters + 7 * n, v, sizeof(v));
  }
  return RYUJIN_OK;
}

/* generic-dim limiter (parity tests of the HIP device function) */
int ryujin_oracle_euler_limit(const ryujin_hip_params *p, const double *bounds, const double *U,
                              const double *P, double *l, int *success)
{
  auto run = [&](auto dim_tag) {
    constexpr int dim = decltype(dim_tag)::value;
    euler::View<dim> view(*p);
    euler::Limiter<dim> lim(view, *p);
    typename euler::View<dim>::state_type u, pp;
    for (int q = 0; q < dim + 2; ++q) {
      u[q] = U[q];
      pp[q] = P[q];
    }
    const auto [t, s] = lim.limit({{bounds[0], bounds[1], bounds[2]}}, u, pp);
    *l = t;
    *success = s;
  };
  if (p->dim == 1)
    run(std::integral_constant<int, 1>{});
  else if (p->dim == 2)
    run(std::integral_constant<int, 2>{});
  else
    run(std::integral_constant<int, 3>{});
  return RYUJIN_OK;
}

/* generic-dim limiter with its trace (parity tests: classify HIP-vs-oracle differences of l_ij).
 * out[0] = l, out[1] = success, out[2] = t_r after the density clip, out[3] = psi_r of the first Newton
 * iteration (the sign of which decides "accept t_r" against "iterate from t_l = 0",
 * limiter.template.h:188-216), out[4] = number of recorded iterations */
int ryujin_oracle_euler_limit_trace(const ryujin_hip_params *p, const double *bounds, const double *U,
                                    const double *P, double out[5])
{
  auto run = [&](auto dim_tag) {
    constexpr int dim = decltype(dim_tag)::value;
    euler::View<dim> view(*p);
    euler::Limiter<dim> lim(view, *p);
    typename euler::View<dim>::state_type u, pp;
    for (int q = 0; q < dim + 2; ++q) {
      u[q] = U[q];
      pp[q] = P[q];
    }
    euler::LimiterTrace tr;
    const auto [t, s] = lim.limit({{bounds[0], bounds[1], bounds[2]}}, u, pp, 0., 1., &tr);
    out[0] = t;
    out[1] = s ? 1. : 0.;
    out[2] = tr.t_r_start;
    out[3] = tr.iterations.empty() ? 0. : tr.iterations[0].psi_r;
    out[4] = (double)tr.iterations.size();
  };
  if (p->dim == 1)
    run(std::integral_constant<int, 1>{});
  else if (p->dim == 2)
    run(std::integral_constant<int, 2>{});
  else
    run(std::integral_constant<int, 3>{});
  return RYUJIN_OK;
}

/* Euler HyperbolicSystemView scalar functions for state U in dimension p->dim:
 * out layout: internal_energy, internal_energy_derivative[k], pressure, specific_entropy,
 * harten_entropy, harten_entropy_derivative[k], mathematical_entropy,
 * mathematical_entropy_derivative[k], f[k*dim], speed_of_sound */
int ryujin_oracle_euler_view(const ryujin_hip_params *p, const double *U, double *out)
{
  auto run = [&](auto dim_tag) {
    constexpr int dim = decltype(dim_tag)::value;
    constexpr int k = dim + 2;
    euler::View<dim> v(*p);
    typename euler::View<dim>::state_type u;
    for (int q = 0; q < k; ++q)
      u[q] = U[q];
    double *o = out;
    *o++ = v.internal_energy(u);
    for (auto x : v.internal_energy_derivative(u))
      *o++ = x;
    *o++ = v.pressure(u);
    *o++ = v.specific_entropy(u);
    *o++ = v.harten_entropy(u);
    for (auto x : v.harten_entropy_derivative(u))
      *o++ = x;
    *o++ = v.mathematical_entropy(u);
    for (auto x : v.mathematical_entropy_derivative(u))
      *o++ = x;
    const auto f = v.f(u);
    for (int q = 0; q < k; ++q)
      for (int d = 0; d < dim; ++d)
        *o++ = f[q][d];
    *o++ = v.speed_of_sound(u);
  };
  if (p->dim == 1)
    run(std::integral_constant<int, 1>{});
  else if (p->dim == 2)
    run(std::integral_constant<int, 2>{});
  else
    run(std::integral_constant<int, 3>{});
  return RYUJIN_OK;
}

/* boundary conditions on a single state (dim from params) */
int ryujin_oracle_euler_apply_bc(const ryujin_hip_params *p, int id, const double *U,
                                 const double *normal, const double *U_dirichlet, double *out)
{
  auto run = [&](auto dim_tag) {
    constexpr int dim = decltype(dim_tag)::value;
    constexpr int k = dim + 2;
    euler::View<dim> v(*p);
    typename euler::View<dim>::state_type u, ud;
    std::array<double, dim> n;
    for (int q = 0; q < k; ++q) {
      u[q] = U[q];
      ud[q] = U_dirichlet[q];
    }
    for (int d = 0; d < dim; ++d)
      n[d] = normal[d];
    const auto r = v.apply_boundary_conditions(id, u, n, ud);
    for (int q = 0; q < k; ++q)
      out[q] = r[q];
  };
  if (p->dim == 1)
    run(std::integral_constant<int, 1>{});
  else if (p->dim == 2)
    run(std::integral_constant<int, 2>{});
  else
    run(std::integral_constant<int, 3>{});
  return RYUJIN_OK;
}

/* shallow water Riemann solver on 1-D riemann data {h, u, a}: out = {h_star, lambda_max} */
int ryujin_oracle_sw_riemann(const ryujin_hip_params *p, const double rd_i[3], const double rd_j[3],
                             double out[2])
{
  shallow_water::RiemannSolver rs(*p);
  double h_star = 0.;
  out[1] = rs.compute({{rd_i[0], rd_i[1], rd_i[2]}}, {{rd_j[0], rd_j[1], rd_j[2]}}, &h_star);
  out[0] = h_star;
  return RYUJIN_OK;
}

/* shallow water d_ij = |c_ij| lambda_max(U_i, U_j, c_ij / |c_ij|) for n independent pairs, dim = p->dim */
int ryujin_oracle_sw_dij_batch(const ryujin_hip_params *p, size_t n, const double *U_i, const double *U_j,
                               const double *c, double *out)
{
  const shallow_water::RiemannSolver rs(*p);
  const int dim = p->dim, k = dim + 1;
#pragma omp parallel for schedule(static)
  for (size_t q = 0; q < n; ++q) {
    double norm2 = 0.;
    for (int d = 0; d < dim; ++d)
      norm2 += c[q * dim + d] * c[q * dim + d];
    const double norm = std::sqrt(norm2), inverse = 1. / norm;
    if (dim == 1) {
      out[q] = norm * rs.compute<1>({{U_i[q * k], U_i[q * k + 1]}}, {{U_j[q * k], U_j[q * k + 1]}},
                                    {{c[q] * inverse}});
    } else {
      out[q] = norm * rs.compute<2>({{U_i[q * k], U_i[q * k + 1], U_i[q * k + 2]}},
                                    {{U_j[q * k], U_j[q * k + 1], U_j[q * k + 2]}},
                                    {{c[q * 2] * inverse, c[q * 2 + 1] * inverse}});
    }
  }
  return RYUJIN_OK;
}

/* import check: logical CSR view (ptr, col, transposed position) of a reference layout */
int ryujin_oracle_import_csr(const ryujin_hip_offline *o, uint64_t *ptr, uint32_t *col,
                             uint64_t *transpose, const double *data, uint32_t n_comp, double *out)
{
  return guarded([&]() {
    CSR csr;
    csr.import(*o);
    std::copy(csr.ptr.begin(), csr.ptr.end(), ptr);
    if (col)
      std::copy(csr.col.begin(), csr.col.end(), col);
    if (transpose)
      std::copy(csr.transpose.begin(), csr.transpose.end(), transpose);
    if (data && out) {
      const auto g = csr.gather(*o, data, n_comp);
      std::copy(g.begin(), g.end(), out);
    }
    return RYUJIN_OK;
  });
}

/* ---- EulerAEOS function-level entry points (the tests under tests/euler_aeos of the reference) ---- */

/* RiemannSolver::compute(riemann_data_i, riemann_data_j): rd = (rho, u, p, gamma, a).
 * trace[7] = RS p_1, RS p_2, SS p_1, SS p_2, interpolated p, p_star, phi(p_star) */
int ryujin_oracle_aeos_riemann(const ryujin_hip_params *p, const double rd_i[5], const double rd_j[5],
                               double *lambda_max, double *trace)
{
  return guarded([&]() {
    const aeos::EquationOfState eos(*p);
    aeos::RiemannSolver rs(eos.interpolation_b, eos.interpolation_pinfty, p->compute_strict_bounds != 0);
    aeos::RiemannTrace t;
    if (trace)
      rs.trace = &t;
    aeos::RiemannSolver::primitive_type a, b;
    for (int q = 0; q < 5; ++q) {
      a[q] = rd_i[q];
      b[q] = rd_j[q];
    }
    *lambda_max = rs.compute(a, b);
    if (trace) {
      const double v[7] = {t.rs_p_1, t.rs_p_2, t.ss_p_1, t.ss_p_2, t.interpolated, t.p_star, t.phi_p_star};
      std::copy(v, v + 7, trace);
    }
    return RYUJIN_OK;
  });
}

/* the whole pipeline from states: precomputed pressures -> riemann data -> lambda_max */
double ryujin_oracle_aeos_lambda_max(const ryujin_hip_params *p, const double *U_i, const double *U_j,
                                     const double *n_ij)
{
  auto run = [&](auto tag) {
    constexpr int dim = decltype(tag)::value;
    const aeos::View<dim> view(*p);
    const aeos::RiemannSolver rs(view.b(), view.pinf(), view.compute_strict_bounds);
    typename aeos::View<dim>::state_type a, b;
    std::array<double, dim> n;
    for (int q = 0; q < dim + 2; ++q) {
      a[q] = U_i[q];
      b[q] = U_j[q];
    }
    for (int d = 0; d < dim; ++d)
      n[d] = n_ij[d];
    return rs.template compute<dim>(view, a, view.eos_pressure_of_state(a), b,
                                    view.eos_pressure_of_state(b), n);
  };
  switch (p->dim) {
  case 1: return run(std::integral_constant<int, 1>{});
  case 2: return run(std::integral_constant<int, 2>{});
  default: return run(std::integral_constant<int, 3>{});
  }
}

/* Limiter<1>::limit(bounds[4], U[3], P[3]); trace_out: t_l_start, t_r_start, n_iter, then per
 * iteration (psi_l, psi_r, dpsi_l, dpsi_r, t_l, t_r, newton?) */
int ryujin_oracle_aeos_limit(const ryujin_hip_params *p, int expensive_bounds_check,
                             const double *bounds, const double *U, const double *P, double *l,
                             int *success, double *trace_out, int trace_cap)
{
  return guarded([&]() {
    auto run = [&](auto tag) {
      constexpr int dim = decltype(tag)::value;
      const aeos::View<dim> view(*p);
      aeos::Limiter<dim> limiter(view, *p);
      limiter.expensive_bounds_check = expensive_bounds_check != 0;
      aeos::LimiterTrace t;
      limiter.trace = &t;
      typename aeos::Limiter<dim>::Bounds bnd;
      typename aeos::View<dim>::state_type u, pp;
      for (int q = 0; q < 4; ++q)
        bnd[q] = bounds[q];
      for (int q = 0; q < dim + 2; ++q) {
        u[q] = U[q];
        pp[q] = P[q];
      }
      const auto [t_l, ok] = limiter.limit(bnd, u, pp);
      *l = t_l;
      *success = ok ? 1 : 0;
      if (trace_out && trace_cap >= 3) {
        trace_out[0] = t.t_l_start;
        trace_out[1] = t.t_r_start;
        int n = 0;
        for (const auto &it : t.iters) {
          if (3 + 7 * (n + 1) > trace_cap)
            break;
          double *o = trace_out + 3 + 7 * n;
          o[0] = it.psi_l; o[1] = it.psi_r; o[2] = it.dpsi_l; o[3] = it.dpsi_r;
          o[4] = it.t_l; o[5] = it.t_r; o[6] = it.newton ? 1. : 0.;
          ++n;
        }
        trace_out[2] = n;
      }
      return RYUJIN_OK;
    };
    switch (p->dim) {
    case 1: return run(std::integral_constant<int, 1>{});
    case 2: return run(std::integral_constant<int, 2>{});
    default: return run(std::integral_constant<int, 3>{});
    }
  });
}

/* tests/euler_aeos/hyperbolic_system.cc: out = internal_energy, internal_energy_derivative[k],
 * surrogate_specific_entropy, surrogate_harten_entropy, its derivative[k], surrogate_pressure,
 * surrogate_gamma(U, that pressure), f(U, p)[k*dim] -- all with gamma_min = gamma_in */
int ryujin_oracle_aeos_view(const ryujin_hip_params *p, const double *U, double gamma_in, double *out)
{
  return guarded([&]() {
    auto run = [&](auto tag) {
      constexpr int dim = decltype(tag)::value;
      constexpr int k = dim + 2;
      const aeos::View<dim> view(*p);
      typename aeos::View<dim>::state_type u;
      for (int q = 0; q < k; ++q)
        u[q] = U[q];
      int n = 0;
      out[n++] = view.internal_energy(u);
      for (double v : view.internal_energy_derivative(u))
        out[n++] = v;
      out[n++] = view.surrogate_specific_entropy(u, gamma_in);
      const double eta = view.surrogate_harten_entropy(u, gamma_in);
      out[n++] = eta;
      for (double v : view.surrogate_harten_entropy_derivative(u, eta, gamma_in))
        out[n++] = v;
      const double ps = view.surrogate_pressure(u, gamma_in);
      out[n++] = ps;
      out[n++] = view.surrogate_gamma(u, ps);
      const auto f = view.f(u, ps);
      for (int q = 0; q < k; ++q)
        for (int d = 0; d < dim; ++d)
          out[n++] = f[q][d];
      return RYUJIN_OK;
    };
    switch (p->dim) {
    case 1: return run(std::integral_constant<int, 1>{});
    case 2: return run(std::integral_constant<int, 2>{});
    default: return run(std::integral_constant<int, 3>{});
    }
  });
}

/* EquationOfState: out = pressure(rho,e), specific_internal_energy(rho,p_in), temperature(rho,e),
 * speed_of_sound(rho,e), interpolation b, pinfty, q */
int ryujin_oracle_aeos_eos(const ryujin_hip_params *p, double rho, double e, double p_in, double *out)
{
  return guarded([&]() {
    const aeos::EquationOfState eos(*p);
    out[0] = eos.pressure(rho, e);
    out[1] = eos.specific_internal_energy(rho, p_in);
    out[2] = eos.temperature(rho, e);
    out[3] = eos.speed_of_sound(rho, e);
    out[4] = eos.interpolation_b;
    out[5] = eos.interpolation_pinfty;
    out[6] = eos.interpolation_q;
    return RYUJIN_OK;
  });
}

/* ---- scalar conservation function-level entry points ------------------------------------------ */

/* RiemannSolver::compute(u_i, u_j, prec_i, prec_j, n_ij) with prec = (f, df) of the selected flux;
 * trace[11] = f_i, f_j, df_i, df_j, Roe average, second, third, k, f_k, left, right wavespeed */
int ryujin_oracle_scalar_riemann(const ryujin_hip_params *p, double u_i, double u_j, const double *n_ij,
                                 double *lambda_max, double *trace)
{
  return guarded([&]() {
    auto run = [&](auto tag) {
      constexpr int dim = decltype(tag)::value;
      const scalar::View<dim> view(*p);
      scalar::RiemannSolver<dim> rs(view, *p);
      scalar::RiemannTrace t{};
      rs.trace = &t;
      std::array<double, dim> n;
      for (int d = 0; d < dim; ++d)
        n[d] = n_ij[d];
      *lambda_max = rs.compute(u_i, u_j, view.precompute(u_i), view.precompute(u_j), n);
      if (trace) {
        const double v[11] = {t.f_i, t.f_j, t.df_i, t.df_j, t.roe, t.second, t.third, t.k, t.f_k, t.left, t.right};
        std::copy(v, v + 11, trace);
      }
      return RYUJIN_OK;
    };
    switch (p->dim) {
    case 1: return run(std::integral_constant<int, 1>{});
    case 2: return run(std::integral_constant<int, 2>{});
    default: return run(std::integral_constant<int, 3>{});
    }
  });
}

/* flux_function(u)[dim], flux_gradient_function(u)[dim] */
int ryujin_oracle_scalar_flux(const ryujin_hip_params *p, double u, double *out)
{
  return guarded([&]() {
    const scalar::Flux flux(*p);
    for (int d = 0; d < p->dim; ++d) {
      out[d] = flux.value(u, d);
      out[p->dim + d] = flux.gradient(u, d);
    }
    return RYUJIN_OK;
  });
}

/* Limiter::limit(bounds[2], u, p) */
int ryujin_oracle_scalar_limit(const ryujin_hip_params *p, int expensive_bounds_check, const double *bounds,
                               double u, double pp, double *l, int *success)
{
  return guarded([&]() {
    scalar::Limiter<1> limiter(*p);
    limiter.expensive_bounds_check = expensive_bounds_check != 0;
    const auto [t, ok] = limiter.limit({{bounds[0], bounds[1]}}, u, pp);
    *l = t;
    *success = ok ? 1 : 0;
    return RYUJIN_OK;
  });
}

} /* extern "C" */

// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (scalar double) of ryujin's shallow-water "Description" and of the module-level
// special cases of HyperbolicModule::step for it:
//   HyperbolicSystemView   source/shallow_water/hyperbolic_system.h:676-1248
//   RiemannSolver          source/shallow_water/riemann_solver.template.h:25-251
//   Indicator              source/shallow_water/indicator.h:155-222
//   Limiter                source/shallow_water/limiter.h:247-363, limiter.template.h:16-449
//   module special cases   source/hyperbolic_module.template.h:270-271,660-720,773-787,797-846
//
// Parity status: RiemannSolver PINNED against tests/shallow_water/riemann_solver.output
// (tests/test_oracle_golden_sw.py). The reference holds no unit golden for the SW limiter or
// indicator, and its SW integration tests use initial states outside the scope of this repository:
// whole-step level "parity unpinned" (checked through invariants: lake at rest, conservation,
// positivity of the water depth).

#pragma once

#include <array>
#include <atomic>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>

#include "hyperbolic_module.hpp"

namespace oracle
{
  namespace shallow_water
  {
    template <int dim>
    struct View {
      static constexpr int k = dim + 1;
      using state_type = std::array<double, k>;
      using flux_type = std::array<std::array<double, dim>, k>;
      using vec_type = std::array<double, dim>;

      double gravity, manning, reference_water_depth, dry_state_relaxation_factor,
          dry_state_relaxation_small, dry_state_relaxation_large;

      explicit View(const ryujin_hip_params &p)
          : gravity(p.gravity)
          , manning(p.manning_friction_coefficient)
          , reference_water_depth(p.reference_water_depth)
          , dry_state_relaxation_factor(p.dry_state_relaxation_factor)
          , dry_state_relaxation_small(p.dry_state_relaxation_small)
          , dry_state_relaxation_large(p.dry_state_relaxation_large)
      {
      }

      static double water_depth(const state_type &U) { return U[0]; }

      /* hyperbolic_system.h:729-742 */
      double inverse_water_depth_mollified(const state_type &U) const
      {
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double h_cutoff_mollified = reference_water_depth * dry_state_relaxation_large * eps;
        const double h = U[0];
        const double h_pos = positive_part(h);
        const double h_max = std::max(h, h_cutoff_mollified);
        const double denom = h * h + h_max * h_max;
        return 2. * h_pos / denom;
      }

      /* :747-758 */
      double water_depth_sharp(const state_type &U) const
      {
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double h_cutoff_small = reference_water_depth * dry_state_relaxation_small * eps;
        return std::max(U[0], h_cutoff_small);
      }
      double inverse_water_depth_sharp(const state_type &U) const
      {
        return 1. / water_depth_sharp(U);
      }

      /* :773-784 */
      double filter_dry_water_depth(const double h) const
      {
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double h_cutoff_large = reference_water_depth * dry_state_relaxation_large * eps;
        return std::abs(h) < h_cutoff_large ? 0. : h;
      }

      /* :801-810 */
      double kinetic_energy(const state_type &U) const
      {
        const double h = U[0];
        const double ih = inverse_water_depth_sharp(U);
        double v2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double v = U[1 + d] * ih;
          v2 += v * v;
        }
        return 0.5 * h * v2;
      }

      /* :815-822 */
      double pressure(const state_type &U) const
      {
        const double h_sqd = U[0] * U[0];
        return 0.5 * gravity * h_sqd;
      }

      double speed_of_sound(const state_type &U) const { return std::sqrt(gravity * U[0]); }

      /* :837-844 */
      double mathematical_entropy(const state_type &U) const
      {
        return pressure(U) + kinetic_energy(U);
      }

      /* :849-877 */
      state_type mathematical_entropy_derivative(const state_type &U) const
      {
        state_type result;
        const double h = U[0];
        const double ih = inverse_water_depth_sharp(U);
        double v2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double v = U[1 + d] * ih;
          v2 += v * v;
          result[1 + d] = v;
        }
        result[0] = gravity * h - 0.5 * v2;
        return result;
      }

      /* :882-902 */
      bool is_admissible(const state_type &U) const { return filter_dry_water_depth(U[0]) >= 0.; }

      /* :907-949 */
      template <int component>
      state_type prescribe_riemann_characteristic(const state_type &U, const state_type &U_bar,
                                                  const vec_type &normal) const
      {
        const double a = speed_of_sound(U);
        double mn = 0.;
        for (int d = 0; d < dim; ++d)
          mn += U[1 + d] * normal[d];
        const double vn = mn * inverse_water_depth_sharp(U);

        const double a_bar = speed_of_sound(U_bar);
        double mn_bar = 0.;
        for (int d = 0; d < dim; ++d)
          mn_bar += U_bar[1 + d] * normal[d];
        const double vn_bar = mn_bar * inverse_water_depth_sharp(U_bar);

        const double R_1 = component == 1 ? vn_bar - 2. * a_bar : vn - 2. * a;
        const double R_2 = component == 2 ? vn_bar + 2. * a_bar : vn + 2. * a;

        vec_type vperp;
        const double ih = inverse_water_depth_sharp(U);
        for (int d = 0; d < dim; ++d)
          vperp[d] = U[1 + d] * ih - vn * normal[d];

        const double vn_new = 0.5 * (R_1 + R_2);
        const double tmp = (R_2 - R_1) / 4.;
        const double h_new = tmp * tmp / gravity;

        state_type U_new;
        U_new[0] = h_new;
        for (int d = 0; d < dim; ++d)
          U_new[1 + d] = h_new * (vn_new * normal[d] + vperp[d]);
        return U_new;
      }

      /* :954-1017 */
      state_type apply_boundary_conditions(int id, const state_type &U, const vec_type &normal,
                                           const state_type &U_dirichlet) const
      {
        state_type result = U;
        if (id == RYUJIN_BC_DIRICHLET) {
          result = U_dirichlet;
        } else if (id == RYUJIN_BC_DIRICHLET_MOMENTUM) {
          for (int d = 0; d < dim; ++d)
            result[1 + d] = U_dirichlet[1 + d];
        } else if (id == RYUJIN_BC_SLIP) {
          double mn = 0.;
          for (int d = 0; d < dim; ++d)
            mn += U[1 + d] * normal[d];
          for (int d = 0; d < dim; ++d)
            result[1 + d] = U[1 + d] - 1. * mn * normal[d];
        } else if (id == RYUJIN_BC_NO_SLIP) {
          for (int d = 0; d < dim; ++d)
            result[1 + d] = 0.;
        } else if (id == RYUJIN_BC_DYNAMIC) {
          const double h_inverse = inverse_water_depth_sharp(U);
          const double a = speed_of_sound(U);
          double mn = 0.;
          for (int d = 0; d < dim; ++d)
            mn += U[1 + d] * normal[d];
          const double vn = mn * h_inverse;
          if (vn < -a)
            result = U_dirichlet;
          if (vn >= -a && vn <= 0.)
            result = prescribe_riemann_characteristic<2>(U_dirichlet, U, normal);
          if (vn > 0. && vn <= a)
            result = prescribe_riemann_characteristic<1>(U, U_dirichlet, normal);
        }
        return result;
      }

      /* f: :1022-1037, g: :1042-1055 */
      flux_type f(const state_type &U) const
      {
        flux_type result = g(U);
        const double p = pressure(U);
        for (int i = 0; i < dim; ++i)
          result[1 + i][i] += p;
        return result;
      }
      flux_type g(const state_type &U) const
      {
        const double h_inverse = inverse_water_depth_sharp(U);
        flux_type result;
        for (int d = 0; d < dim; ++d)
          result[0][d] = (U[1 + d] * h_inverse) * U[0];
        for (int i = 0; i < dim; ++i)
          for (int d = 0; d < dim; ++d)
            result[1 + i][d] = (U[1 + d] * h_inverse) * U[1 + i];
        return result;
      }

      /* :1060-1070 */
      state_type star_state(const state_type &U, const double Z_left, const double Z_right) const
      {
        const double Z_max = std::max(Z_left, Z_right);
        const double h = U[0];
        const double H_star = std::max(0., h + Z_left - Z_max);
        const double ihm = inverse_water_depth_mollified(U);
        state_type r;
        for (int q = 0; q < k; ++q)
          r[q] = U[q] * H_star * ihm;
        return r;
      }

      /* :1117-1144 */
      state_type flux_divergence(const state_type &U_i, const double Z_i, const state_type &U_j,
                                 const double Z_j, const vec_type &c_ij) const
      {
        const auto U_star_ij = star_state(U_i, Z_i, Z_j);
        const auto U_star_ji = star_state(U_j, Z_j, Z_i);
        const double H_i = U_i[0];
        const double H_star_ij = U_star_ij[0];
        const double H_star_ji = U_star_ji[0];
        const auto g_i = g(U_star_ij);
        const auto g_j = g(U_star_ji);
        flux_type result;
        for (int q = 0; q < k; ++q)
          for (int d = 0; d < dim; ++d)
            result[q][d] = -(g_i[q][d] + g_j[q][d]);
        const double factor =
            (0.5 * (H_star_ji * H_star_ji - H_star_ij * H_star_ij) + H_i * H_i) * gravity;
        for (int i = 0; i < dim; ++i)
          result[1 + i][i] -= factor;
        return contract(result, c_ij);
      }

      /* :1149-1171 */
      state_type high_order_flux_divergence(const state_type &U_i, const double Z_i,
                                            const state_type &U_j, const double Z_j,
                                            const vec_type &c_ij) const
      {
        const double H_i = U_i[0];
        const double H_j = U_j[0];
        const auto g_i = g(U_i);
        const auto g_j = g(U_j);
        flux_type result;
        for (int q = 0; q < k; ++q)
          for (int d = 0; d < dim; ++d)
            result[q][d] = -(g_i[q][d] + g_j[q][d]);
        const double factor = gravity * H_i * (H_j + Z_j - Z_i);
        for (int i = 0; i < dim; ++i)
          result[1 + i][i] -= factor;
        return contract(result, c_ij);
      }

      /* :1176-1191 */
      state_type affine_shift(const state_type &U_i, const double Z_i, const state_type & /*U_j*/,
                              const double Z_j, const vec_type &c_ij, const double d_ij) const
      {
        const auto U_star_ij = star_state(U_i, Z_i, Z_j);
        const double h_inverse = inverse_water_depth_sharp(U_i);
        double m_c = 0.;
        for (int d = 0; d < dim; ++d)
          m_c += U_i[1 + d] * c_ij[d];
        const double factor = 2. * (d_ij + h_inverse * m_c);
        state_type r;
        for (int q = 0; q < k; ++q)
          r[q] = -factor * (U_star_ij[q] - U_i[q]);
        return r;
      }

      /* :1196-1218 */
      state_type manning_friction(const state_type &U, const double h_star, const double tau) const
      {
        state_type result;
        result.fill(0.);
        const double h_inverse = inverse_water_depth_mollified(U);
        double v2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double v = U[1 + d] * h_inverse;
          v2 += v * v;
        }
        const double v_norm = std::sqrt(v2);
        const double factor = 2. * gravity * manning * manning * v_norm;
        const double denominator = h_star + std::max(h_star, tau * factor);
        const double denominator_inverse = 1. / denominator;
        for (int d = 0; d < dim; ++d)
          result[d + 1] = -factor * denominator_inverse * U[1 + d];
        return result;
      }

      static state_type contract(const flux_type &f, const vec_type &c)
      {
        state_type r;
        for (int q = 0; q < k; ++q) {
          double s = 0.;
          for (int d = 0; d < dim; ++d)
            s += f[q][d] * c[d];
          r[q] = s;
        }
        return r;
      }
    };


    /* riemann data = {h, u, a}; riemann_solver.template.h */
    struct RiemannSolver {
      double gravity, reference_water_depth, dry_small;

      explicit RiemannSolver(const ryujin_hip_params &p)
          : gravity(p.gravity)
          , reference_water_depth(p.reference_water_depth)
          , dry_small(p.dry_state_relaxation_small)
      {
      }

      using primitive_type = std::array<double, 3>;

      /* :25-44 */
      double f(const primitive_type &rd, const double h) const
      {
        const auto &[h_Z, u_Z, a_Z] = rd;
        (void)u_Z;
        const double left_value = 2. * (std::sqrt(gravity * h) - a_Z);
        const double radicand = 0.5 * gravity * (h + h_Z) / (h * h_Z);
        const double right_value = (h - h_Z) * std::sqrt(radicand);
        return h <= h_Z ? left_value : right_value;
      }
      /* :47-62 */
      double phi(const primitive_type &rd_i, const primitive_type &rd_j, const double h) const
      {
        return f(rd_i, h) + f(rd_j, h) + rd_j[1] - rd_i[1];
      }
      /* :65-94 */
      double lambda1_minus(const primitive_type &rd, const double h_star) const
      {
        const auto &[h, u, a] = rd;
        const double factor = positive_part((h_star - h) / h);
        const double half_factor = 0.5 * factor;
        return u - a * std::sqrt((1. + half_factor) * (1. + factor));
      }
      double lambda3_plus(const primitive_type &rd, const double h_star) const
      {
        const auto &[h, u, a] = rd;
        const double factor = positive_part((h_star - h) / h);
        const double half_factor = 0.5 * factor;
        return u + a * std::sqrt((1. + half_factor) * (1. + factor));
      }
      /* :97-108 */
      double compute_lambda(const primitive_type &rd_i, const primitive_type &rd_j,
                            const double h_star) const
      {
        const double lambda1 = lambda1_minus(rd_i, h_star);
        const double lambda3 = lambda3_plus(rd_j, h_star);
        return std::max(negative_part(lambda1), positive_part(lambda3));
      }
      /* :111-204; the first mask result (:195-197) is overwritten at :199-201 -- as written */
      double compute_h_star(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const double gravity_inverse = 1. / gravity;
        const auto &[h_i, u_i, a_i] = rd_i;
        const auto &[h_j, u_j, a_j] = rd_j;
        const double h_min = std::min(h_i, h_j);
        const double h_max = std::max(h_i, h_j);
        const double a_min = std::sqrt(gravity * h_min);
        const double a_max = std::sqrt(gravity * h_max);
        const double sqrt_two = std::sqrt(2.);
        const double x0 = 9. - 4. * sqrt_two;
        const double phi_value_min = phi(rd_i, rd_j, x0 * h_min);
        const double phi_value_max = phi(rd_i, rd_j, x0 * h_max);

        double tmp = positive_part(u_i - u_j + 2. * (a_i + a_j));
        const double h_star_left = 0.0625 * gravity_inverse * tmp * tmp;

        tmp = 1. + sqrt_two * (u_i - u_j) / (a_min + a_max);
        const double h_star_middle = std::sqrt(h_min * h_max) * tmp;

        const double left_radicand = 3. * h_min + 2. * sqrt_two * std::sqrt(h_min * h_max);
        const double right_radicand = sqrt_two * std::sqrt(gravity_inverse * h_min) * (u_i - u_j);
        tmp = std::sqrt(positive_part(left_radicand + right_radicand));
        tmp -= sqrt_two * std::sqrt(h_min);
        const double h_star_right = tmp * tmp;

        double h_star = 0. <= phi_value_min ? h_star_left : h_star_right;
        h_star = phi_value_max < 0. ? h_star_middle : h_star_right;
        return h_star;
      }

      double compute(const primitive_type &rd_i, const primitive_type &rd_j,
                     double *h_star_out = nullptr) const
      {
        const double h_star = compute_h_star(rd_i, rd_j);
        if (h_star_out)
          *h_star_out = h_star;
        return compute_lambda(rd_i, rd_j, h_star);
      }

      /* :207-223 */
      template <int dim>
      primitive_type riemann_data_from_state(const std::array<double, dim + 1> &U,
                                             const std::array<double, dim> &n_ij) const
      {
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double h = std::max(U[0], reference_water_depth * dry_small * eps);
        double projected_velocity = 0.;
        for (int d = 0; d < dim; ++d)
          projected_velocity += n_ij[d] * (U[1 + d] / h);
        const double a = std::sqrt(h * gravity);
        return {{h, projected_velocity, a}};
      }

      template <int dim>
      double compute(const std::array<double, dim + 1> &U_i, const std::array<double, dim + 1> &U_j,
                     const std::array<double, dim> &n_ij) const
      {
        return compute(riemann_data_from_state<dim>(U_i, n_ij), riemann_data_from_state<dim>(U_j, n_ij));
      }
    };


    /* indicator.h:155-222 */
    template <int dim>
    struct Indicator {
      using V = View<dim>;
      using state_type = typename V::state_type;
      const V &view;
      double evc_factor;
      double eta_i = 0.;
      state_type d_eta_i;
      typename V::flux_type f_i;
      double left = 0.;
      state_type right;

      Indicator(const V &view, const ryujin_hip_params &p)
          : view(view)
          , evc_factor(p.indicator_evc_factor)
      {
      }
      void reset(const state_type &U_i, const double eta_m)
      {
        eta_i = eta_m;
        d_eta_i = view.mathematical_entropy_derivative(U_i);
        f_i = view.f(U_i);
        left = 0.;
        right.fill(0.);
      }
      void accumulate(const state_type &U_j, const double eta_j, const std::array<double, dim> &c_ij)
      {
        const double ih = view.inverse_water_depth_sharp(U_j);
        const auto f_j = view.f(U_j);
        const double pressure_j = view.pressure(U_j);
        double v_c = 0.;
        for (int d = 0; d < dim; ++d)
          v_c += (U_j[1 + d] * ih) * c_ij[d];
        left += (eta_j + pressure_j) * v_c;
        for (int q = 0; q < V::k; ++q) {
          double s = 0.;
          for (int d = 0; d < dim; ++d)
            s += (f_j[q][d] - f_i[q][d]) * c_ij[d];
          right[q] += s;
        }
      }
      double alpha(const double hd_i) const
      {
        double my_sum = 0.;
        for (int q = 0; q < V::k; ++q)
          my_sum += d_eta_i[q] * right[q];
        const double numerator = std::abs(left - my_sum);
        const double denominator = std::abs(left) + std::abs(my_sum);
        const double regularization = 100. * std::numeric_limits<double>::min();
        const double quotient =
            std::abs(numerator) / (denominator + std::max(hd_i * std::abs(eta_i), regularization));
        return std::min(1., evc_factor * quotient);
      }
    };


    /* limiter.h:247-363, limiter.template.h:16-449 */
    template <int dim>
    struct Limiter {
      using V = View<dim>;
      using state_type = typename V::state_type;
      static constexpr int n_bounds = 5;
      using Bounds = std::array<double, n_bounds>;

      const V &view;
      double newton_tolerance, relaxation_factor;
      bool limit_on_kinetic_energy, limit_on_square_velocity;
      bool expensive_bounds_check = false;

      state_type U_i;
      Bounds bounds_;
      double h_relaxation_numerator = 0., kin_relaxation_numerator = 0., v2_relaxation_numerator = 0.,
             relaxation_denominator = 0.;

      Limiter(const V &view, const ryujin_hip_params &p)
          : view(view)
          , newton_tolerance(p.limiter_newton_tolerance)
          , relaxation_factor(p.limiter_relaxation_factor)
          , limit_on_kinetic_energy(p.limiter_limit_on_kinetic_energy != 0)
          , limit_on_square_velocity(p.limiter_limit_on_square_velocity != 0)
      {
      }

      void reset(const state_type &new_U_i)
      {
        U_i = new_U_i;
        bounds_ = {{std::numeric_limits<double>::max(), 0., 0., 0., 0.}};
        h_relaxation_numerator = kin_relaxation_numerator = v2_relaxation_numerator = 0.;
        relaxation_denominator = 0.;
      }

      void accumulate(const state_type &U_j, const state_type &U_star_ij, const state_type &U_star_ji,
                      const std::array<double, dim> &scaled_c_ij, const state_type &affine_shift)
      {
        const auto f_star_ij = view.f(U_star_ij);
        const auto f_star_ji = view.f(U_star_ji);
        state_type U_ij_bar;
        for (int q = 0; q < V::k; ++q) {
          double s = 0.;
          for (int d = 0; d < dim; ++d)
            s += (f_star_ij[q][d] + (-f_star_ji[q][d])) * scaled_c_ij[d];
          U_ij_bar[q] = 0.5 * (U_star_ij[q] + U_star_ji[q] + s) + affine_shift[q];
        }
        auto &[h_min, h_max, h_small, kin_max, v2_max] = bounds_;
        (void)h_small;
        const double h_bar_ij = U_ij_bar[0];
        h_min = std::min(h_min, h_bar_ij);
        h_max = std::max(h_max, h_bar_ij);
        kin_max = std::max(kin_max, view.kinetic_energy(U_ij_bar));
        {
          const double ihm = view.inverse_water_depth_mollified(U_ij_bar);
          double v2 = 0.;
          for (int d = 0; d < dim; ++d) {
            const double v = U_ij_bar[1 + d] * ihm;
            v2 += v * v;
          }
          v2_max = std::max(v2_max, v2);
        }
        const double beta_ij = 1.;
        relaxation_denominator += std::abs(beta_ij);
        h_relaxation_numerator += beta_ij * (U_i[0] + U_j[0]);
        kin_relaxation_numerator += beta_ij * (view.kinetic_energy(U_i) + view.kinetic_energy(U_j));
        double v2_i = 0., v2_j = 0.;
        {
          const double ihm_i = view.inverse_water_depth_mollified(U_i);
          const double ihm_j = view.inverse_water_depth_mollified(U_j);
          for (int d = 0; d < dim; ++d) {
            const double vi = U_i[1 + d] * ihm_i, vj = U_j[1 + d] * ihm_j;
            v2_i += vi * vi;
            v2_j += vj * vj;
          }
        }
        v2_relaxation_numerator += beta_ij * (-v2_i + v2_j);
      }

      Bounds bounds(const double hd_i) const
      {
        auto relaxed = bounds_;
        auto &[h_min, h_max, h_small, kin_max, v2_max] = relaxed;
        double r_i = std::sqrt(hd_i);
        if constexpr (dim == 2) {
          const double t = std::sqrt(r_i);
          r_i = t * t * t;
        } else if constexpr (dim == 1) {
          r_i = r_i * r_i * r_i;
        }
        r_i *= relaxation_factor;
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double h_relaxed = 2. * std::abs(h_relaxation_numerator) / (relaxation_denominator + eps);
        h_min = std::max((1. - r_i) * h_min, h_min - h_relaxed);
        h_max = std::min((1. + r_i) * h_max, h_max + h_relaxed);
        const double kin_relaxed =
            2. * std::abs(kin_relaxation_numerator) / (relaxation_denominator + eps);
        kin_max = std::min((1. + r_i) * kin_max, kin_max + kin_relaxed);
        const double v2_relaxed =
            2. * std::abs(v2_relaxation_numerator) / (relaxation_denominator + eps);
        v2_max = std::min((1. + r_i) * v2_max, v2_max + v2_relaxed);
        r_i = hd_i;
        if constexpr (dim == 2)
          r_i = std::sqrt(hd_i);
        r_i *= view.dry_state_relaxation_factor;
        h_small = view.reference_water_depth * r_i;
        return relaxed;
      }

      std::pair<double, bool> limit(const Bounds &bounds, const state_type &U, const state_type &P,
                                    const double t_min = 0., const double t_max = 1.) const
      {
        bool success = true;
        double t_l = t_min, t_r = t_max;
        const auto &[h_min, h_max, h_small, kin_max, v2_max] = bounds;
        constexpr double min = std::numeric_limits<double>::min();
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double relax_small = 1. + view.dry_state_relaxation_small * eps;
        const double relax = 1. + view.dry_state_relaxation_large * eps;

        auto q_dot = [](const state_type &a, const state_type &b) {
          double s = 0.;
          for (int d = 0; d < dim; ++d)
            s += a[1 + d] * b[1 + d];
          return s;
        };

        {
          const double h_U = U[0], h_P = P[0];
          const double test_min = view.filter_dry_water_depth(std::max(0., h_U - relax * h_max));
          const double test_max = view.filter_dry_water_depth(std::max(0., h_min - relax * h_U));
          if (!(test_min == 0. && test_max == 0.))
            success = false;
          const double denominator = 1. / (std::abs(h_P) + eps * h_max + min);
          t_r = h_max < h_U + t_r * h_P ? (h_max - h_U) * denominator : t_r;
          const double h_min_tilde = std::max(h_small, h_min);
          t_r = h_U + t_r * h_P < h_min_tilde ? (h_U - h_min_tilde) * denominator : t_r;
          t_r = std::min(t_r, t_max);
          t_r = std::max(t_r, t_min);
          if (expensive_bounds_check) {
            const double h_new = U[0] + t_r * P[0];
            const double a = view.filter_dry_water_depth(std::max(0., h_new - relax * h_max));
            const double b = view.filter_dry_water_depth(std::max(0., h_min - relax * h_new));
            if (!(a == 0. && b == 0.))
              success = false;
          }
        }

        if (!limit_on_square_velocity && !limit_on_kinetic_energy)
          return {t_l, success};

        if (limit_on_kinetic_energy) {
          state_type U_r;
          for (int q = 0; q < V::k; ++q)
            U_r[q] = U[q] + t_r * P[q];
          const double psi_r = relax_small * U_r[0] * kin_max - 0.5 * q_dot(U_r, U_r);
          t_l = psi_r > 0. ? t_r : t_l;
          if (!limit_on_square_velocity && t_l == t_r)
            return {t_l, success};
          state_type U_l;
          for (int q = 0; q < V::k; ++q)
            U_l[q] = U[q] + t_l * P[q];
          const double h_l = U_l[0];
          const double psi_l = relax_small * h_l * kin_max - 0.5 * q_dot(U_l, U_l);
          const double filtered_h_l = view.filter_dry_water_depth(h_l);
          const double lower_bound = (1. - relax) * filtered_h_l * kin_max - eps;
          if (!(std::min(0., psi_l - lower_bound) == 0.))
            success = false;
          if (!(std::max(0., t_r - t_l - newton_tolerance) == 0.)) {
            const double h_P = P[0];
            const double dpsi_l = h_P * kin_max - q_dot(U, P) - q_dot(P, P) * t_l;
            const double dpsi_r = h_P * kin_max - q_dot(U, P) - q_dot(P, P) * t_r;
            quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
          }
          if (expensive_bounds_check) {
            state_type U_new;
            for (int q = 0; q < V::k; ++q)
              U_new[q] = U[q] + t_l * P[q];
            const double psi_new = relax_small * U_new[0] * kin_max - 0.5 * q_dot(U_new, U_new);
            const double lb = (1. - relax) * U_new[0] * kin_max - eps;
            if (!(std::min(0., psi_new - lb) == 0.))
              success = false;
          }
          if (limit_on_square_velocity) {
            t_r = t_l;
            t_l = t_min;
          }
        }

        if (limit_on_square_velocity) {
          state_type U_r;
          for (int q = 0; q < V::k; ++q)
            U_r[q] = U[q] + t_r * P[q];
          const double h_r = U_r[0];
          const double psi_r = relax_small * h_r * h_r * v2_max - q_dot(U_r, U_r);
          t_l = psi_r > 0. ? t_r : t_l;
          if (t_l == t_r)
            return {t_l, success};
          state_type U_l;
          for (int q = 0; q < V::k; ++q)
            U_l[q] = U[q] + t_l * P[q];
          const double h_l = U_l[0];
          const double psi_l = relax_small * h_l * h_l * v2_max - q_dot(U_l, U_l);
          const double filtered_h_l = view.filter_dry_water_depth(h_l);
          const double lower_bound = (1. - relax) * filtered_h_l * filtered_h_l * v2_max - 100. * eps;
          if (!(std::min(0., psi_l - lower_bound) == 0.))
            success = false;
          if (!(std::max(0., t_r - t_l - newton_tolerance) == 0.)) {
            const double h_U = U[0], h_P = P[0];
            const double dpsi_l =
                (h_U + t_l * h_P) * h_P * v2_max - 2. * (q_dot(U, P) - q_dot(P, P) * t_l);
            const double dpsi_r =
                (h_U + t_r * h_P) * h_P * v2_max - 2. * (q_dot(U, P) - q_dot(P, P) * t_r);
            quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
          }
          if (expensive_bounds_check) {
            state_type U_new;
            for (int q = 0; q < V::k; ++q)
              U_new[q] = U[q] + t_l * P[q];
            const double h_new = U_new[0];
            const double psi_new = relax_small * h_new * h_new * v2_max - q_dot(U_new, U_new);
            const double lb = (1. - relax) * h_new * h_new * v2_max - 100. * eps;
            if (!(std::min(0., psi_new - lb) == 0.))
              success = false;
          }
        }
        return {t_l, success};
      }
    };
  } // namespace shallow_water


  template <int dim>
  struct ShallowWaterModule final : ModuleBase {
    using V = shallow_water::View<dim>;
    static constexpr int K = dim + 1;
    static constexpr int NB = 5;
    using state_type = typename V::state_type;
    using vec_type = std::array<double, dim>;

    V view;
    CSR csr;
    std::vector<double> cij, mij, mi, mi_inv, Z;
    double measure_of_omega;
    std::vector<uint32_t> b_i;
    std::vector<double> b_normal;
    std::vector<uint8_t> b_id;
    std::vector<double> dirichlet;
    std::vector<uint32_t> p_i, p_col, p_j;

    struct State {
      std::vector<double> U, prec;
      bool used = false;
    };
    std::vector<State> states;
    std::vector<double> bounds, r, dij, lij, lij_next, pij;

    int k() const override { return K; }
    int n_prec() const override { return 2; }
    int n_bounds() const override { return NB; }

    bool discontinuous_ansatz = false;
    std::vector<double> incidence, mass_matrix_inverse;

    ShallowWaterModule(const ryujin_hip_offline &o, const ryujin_hip_params &p)
        : view(p)
    {
      params = p;
      if (p.limiter_iterations < 0 || p.limiter_iterations > 2)
        throw std::runtime_error("The number of limiter iterations must be between [0,2]");
      n_export = o.n_export;
      n_owned = o.n_owned;
      n_relevant = o.n_relevant;
      csr.import(o);
      cij = csr.gather(o, o.cij, dim);
      mij = csr.gather(o, o.mij, 1);
      mi.assign(o.mi, o.mi + n_relevant);
      mi_inv.assign(o.mi_inv, o.mi_inv + n_relevant);
      if (o.initial_precomputed)
        Z.assign(o.initial_precomputed, o.initial_precomputed + n_relevant);
      else
        Z.assign(n_relevant, 0.);
      /* discontinuous ansatz (hyperbolic_module.template.h:287-293): incidence matrix, full inverse mass matrix */
      discontinuous_ansatz = o.discontinuous_ansatz != 0;
      if (discontinuous_ansatz) {
        if (!o.incidence || !o.mass_matrix_inverse)
          throw std::runtime_error("discontinuous ansatz without incidence / inverse mass matrix");
        incidence = csr.gather(o, o.incidence, 1);
        mass_matrix_inverse = csr.gather(o, o.mass_matrix_inverse, 1);
      }
      measure_of_omega = o.measure_of_omega;
      b_i.assign(o.b_i, o.b_i + o.n_bdry);
      b_normal.assign(o.b_normal, o.b_normal + (size_t)o.n_bdry * dim);
      b_id.assign(o.b_id, o.b_id + o.n_bdry);
      dirichlet.assign((size_t)o.n_bdry * K, 0.);
      p_i.assign(o.p_i, o.p_i + o.n_pairs);
      p_col.assign(o.p_col, o.p_col + o.n_pairs);
      p_j.assign(o.p_j, o.p_j + o.n_pairs);
      alpha.assign(n_relevant, 0.);
      bounds.assign((size_t)n_relevant * NB, 0.);
      r.assign((size_t)n_relevant * K, 0.);
      dij.assign(csr.nnz(), 0.);
      lij.assign(csr.nnz(), 0.);
      lij_next.assign(csr.nnz(), 0.);
      pij.assign(csr.nnz() * K, 0.);
    }

    int state_alloc() override
    {
      for (size_t h = 0; h < states.size(); ++h)
        if (!states[h].used) {
          states[h].used = true;
          return (int)h;
        }
      states.emplace_back();
      states.back().U.assign((size_t)n_relevant * K, 0.);
      states.back().prec.assign((size_t)n_relevant * 2, 0.);
      states.back().used = true;
      return (int)states.size() - 1;
    }
    void state_free(int h) override { states.at(h).used = false; }
    double *state_U(int h) override { return states.at(h).U.data(); }
    double *state_prec(int h) override { return states.at(h).prec.data(); }

    static state_type get_state(const std::vector<double> &U, uint32_t i)
    {
      state_type s;
      for (int q = 0; q < K; ++q)
        s[q] = U[(size_t)i * K + q];
      return s;
    }
    static void put_state(std::vector<double> &U, uint32_t i, const state_type &s)
    {
      for (int q = 0; q < K; ++q)
        U[(size_t)i * K + q] = s[q];
    }
    vec_type get_c(uint64_t e) const
    {
      vec_type c;
      for (int d = 0; d < dim; ++d)
        c[d] = cij[e * dim + d];
      return c;
    }
    void do_exchange(int what, double *data, int n_comp)
    {
      if (exchange)
        exchange(exchange_user, what, data, n_comp);
    }

    void prepare_state_vector(int h, double /*t*/, const double *dirichlet_in) override
    {
      auto &U = states.at(h).U;
      auto &prec = states.at(h).prec;
      if (dirichlet_in)
        dirichlet.assign(dirichlet_in, dirichlet_in + dirichlet.size());
      for (size_t b = 0; b < b_i.size(); ++b) {
        const int id = b_id[b];
        if (id == RYUJIN_BC_DO_NOTHING)
          continue;
        const uint32_t i = b_i[b];
        vec_type normal;
        for (int d = 0; d < dim; ++d)
          normal[d] = b_normal[b * dim + d];
        state_type U_D;
        for (int q = 0; q < K; ++q)
          U_D[q] = dirichlet[b * K + q];
        put_state(U, i, view.apply_boundary_conditions(id, get_state(U, i), normal, U_D));
      }
      do_exchange(EX_U, U.data(), K);
      /* precomputation_loop: shallow_water/hyperbolic_system.h:676-716 */
#pragma omp parallel for schedule(static)
      for (uint32_t i = 0; i < n_owned; ++i) {
        if (csr.ptr[i + 1] - csr.ptr[i] == 1)
          continue;
        const auto U_i = get_state(U, i);
        prec[(size_t)i * 2 + 0] = view.mathematical_entropy(U_i);
        prec[(size_t)i * 2 + 1] = std::pow(view.water_depth_sharp(U_i), 4. / 3.);
      }
      do_exchange(EX_PREC, prec.data(), 2);
    }

    int step(int h_old, int stages, const int *h_stage, const double *w, int h_new, double tau,
             double tau_max_in, double *tau_out) override
    {
      const auto &old_U = states.at(h_old).U;
      const auto &old_prec = states.at(h_old).prec;
      auto &new_U = states.at(h_new).U;
      const double measure_of_omega_inverse = 1. / measure_of_omega;
      std::atomic<bool> restart_needed{false};
      const shallow_water::RiemannSolver riemann_solver(params);

      auto dij_of = [&](const state_type &A, const state_type &B, const vec_type &c) {
        double norm2 = 0.;
        for (int d = 0; d < dim; ++d)
          norm2 += c[d] * c[d];
        const double norm = std::sqrt(norm2);
        vec_type n;
        const double inverse_norm = 1. / norm; /* dealii::Tensor / scalar multiplies by the inverse */
        for (int d = 0; d < dim; ++d)
          n[d] = c[d] * inverse_norm;
        return norm * riemann_solver.template compute<dim>(A, B, n);
      };

      /* Step 2 */
#pragma omp parallel
      {
        shallow_water::Indicator<dim> indicator(view, params);
#pragma omp for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          const auto U_i = get_state(old_U, i);
          indicator.reset(U_i, old_prec[(size_t)i * 2 + 0]);
          for (uint64_t e = rs; e < re; ++e) {
            const uint32_t j = csr.col[e];
            const auto U_j = get_state(old_U, j);
            const auto c_ij = get_c(e);
            indicator.accumulate(U_j, old_prec[(size_t)j * 2 + 0], c_ij);
            if (e == rs || j < i)
              continue;
            dij[e] = dij_of(U_i, U_j, c_ij);
          }
          alpha[i] = indicator.alpha(mi[i] * measure_of_omega_inverse);
        }
      }
      do_exchange(EX_ALPHA, alpha.data(), 1);

      /* Step 3 */
      for (size_t q = 0; q < p_i.size(); ++q) {
        const uint32_t i = p_i[q], col_idx = p_col[q], j = p_j[q];
        if (j < i)
          continue;
        const uint64_t e = csr.ptr[i] + col_idx;
        const double d_ji = dij_of(get_state(old_U, j), get_state(old_U, i), get_c(csr.transpose[e]));
        dij[e] = std::max(dij[e], d_ji);
      }
      double tau_max = tau_max_in;
      {
        double local_tau_max = std::numeric_limits<double>::max();
#pragma omp parallel for schedule(static) reduction(min : local_tau_max)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          double d_sum = 0.;
          for (uint64_t e = rs + 1; e < re; ++e) {
            if (csr.col[e] < i)
              dij[e] = dij[csr.transpose[e]];
            d_sum -= dij[e];
          }
          d_sum = std::min(d_sum, -1.e6 * std::numeric_limits<double>::min());
          dij[rs] = d_sum;
          local_tau_max = std::min(local_tau_max, params.cfl * mi[i] / (-2. * d_sum));
        }
        tau_max = std::min(tau_max, local_tau_max);
      }
      do_exchange(EX_MIN, &tau_max, 1);
      if (std::isnan(tau_max) || std::isinf(tau_max) || !(tau_max > 0.))
        return RYUJIN_ERR_TAU;
      tau = (tau == 0. ? tau_max : tau);

      /* Step 4 (hyperbolic_module.template.h:597-884 with the shallow-water branches) */
      double weight;
      {
        double acc = -1.;
        for (int s = 0; s < stages; ++s)
          acc += w[s];
        weight = -acc;
      }
#pragma omp parallel
      {
        shallow_water::Limiter<dim> limiter(view, params);
#pragma omp for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          const auto U_i = get_state(old_U, i);
          auto U_i_new = U_i;
          const double alpha_i = alpha[i], m_i = mi[i], m_i_inv = mi_inv[i];
          const double Z_i = Z[i];

          std::array<state_type, 4> U_iHs;
          state_type S_iH;
          S_iH.fill(0.);
          for (int s = 0; s < stages; ++s) {
            const auto &st = states.at(h_stage[s]);
            U_iHs[s] = get_state(st.U, i);
            const auto S = view.manning_friction(U_iHs[s], st.prec[(size_t)i * 2 + 1], tau);
            for (int q = 0; q < K; ++q)
              S_iH[q] += w[s] * S[q];
          }
          const auto S_i = view.manning_friction(U_i, old_prec[(size_t)i * 2 + 1], tau);
          state_type F_iH;
          F_iH.fill(0.);
          for (int q = 0; q < K; ++q) {
            S_iH[q] += weight * S_i[q];
            U_i_new[q] += tau * S_i[q];
            F_iH[q] += m_i * S_iH[q];
          }
          limiter.reset(U_i);

          state_type affine_shift;
          affine_shift.fill(0.);
          for (uint64_t e = rs; e < re; ++e) {
            const uint32_t j = csr.col[e];
            const auto B_ij =
                view.affine_shift(U_i, Z_i, get_state(old_U, j), Z[j], get_c(e), dij[e]);
            for (int q = 0; q < K; ++q)
              affine_shift[q] += B_ij[q];
          }
          for (int q = 0; q < K; ++q) {
            affine_shift[q] *= tau * m_i_inv;
            affine_shift[q] += tau * S_i[q];
          }

          for (uint64_t e = rs; e < re; ++e) {
            const uint32_t j = csr.col[e];
            const auto U_j = get_state(old_U, j);
            const double Z_j = Z[j];
            const double d_ij = dij[e];
            double factor = (alpha_i + alpha[j]) * .5;
            if (discontinuous_ansatz) /* :733-737 */
              factor = std::max(factor, incidence[e]);
            const double d_ijH = d_ij * factor;
            const auto c_ij = get_c(e);
            const double denom = std::max(d_ij, 100. * std::numeric_limits<double>::min());
            vec_type scaled_c_ij;
            const double inverse_denom = 1. / denom; /* dealii::Tensor / scalar multiplies by the inverse */
            for (int d = 0; d < dim; ++d)
              scaled_c_ij[d] = c_ij[d] * inverse_denom;
            const double m_ij = mij[e];

            const auto flux_ij = view.flux_divergence(U_i, Z_i, U_j, Z_j, c_ij);
            state_type P_ij;
            for (int q = 0; q < K; ++q) {
              U_i_new[q] += tau * m_i_inv * flux_ij[q];
              P_ij[q] = -flux_ij[q];
            }
            const auto U_star_ij = view.star_state(U_i, Z_i, Z_j);
            const auto U_star_ji = view.star_state(U_j, Z_j, Z_i);
            for (int q = 0; q < K; ++q) {
              const double dU = U_star_ji[q] - U_star_ij[q];
              U_i_new[q] += tau * m_i_inv * d_ij * dU;
              F_iH[q] += d_ijH * dU;
              P_ij[q] += (d_ijH - d_ij) * dU;
            }
            limiter.accumulate(U_j, U_star_ij, U_star_ji, scaled_c_ij, affine_shift);

            for (int q = 0; q < K; ++q) {
              F_iH[q] -= m_ij * S_iH[q];
              P_ij[q] -= m_ij * /*sic!*/ S_i[q];
            }
            const auto hof = view.high_order_flux_divergence(U_i, Z_i, U_j, Z_j, c_ij);
            for (int q = 0; q < K; ++q) {
              F_iH[q] += weight * hof[q];
              P_ij[q] += weight * hof[q];
            }
            const auto S_j = view.manning_friction(U_j, old_prec[(size_t)j * 2 + 1], tau);
            for (int q = 0; q < K; ++q) {
              F_iH[q] += weight * m_ij * S_j[q];
              P_ij[q] += weight * m_ij * S_j[q];
            }
            for (int s = 0; s < stages; ++s) {
              const auto &st = states.at(h_stage[s]);
              const auto U_jHs = get_state(st.U, j);
              const auto hof_s = view.high_order_flux_divergence(U_iHs[s], Z_i, U_jHs, Z_j, c_ij);
              const auto S_js = view.manning_friction(U_jHs, st.prec[(size_t)j * 2 + 1], tau);
              for (int q = 0; q < K; ++q) {
                F_iH[q] += w[s] * hof_s[q];
                P_ij[q] += w[s] * hof_s[q];
              }
              for (int q = 0; q < K; ++q) {
                F_iH[q] += w[s] * m_ij * S_js[q];
                P_ij[q] += w[s] * m_ij * S_js[q];
              }
            }
            for (int q = 0; q < K; ++q)
              pij[e * K + q] = P_ij[q];
          }
          if (expensive_bounds_check && !view.is_admissible(U_i_new))
            restart_needed = true;
          put_state(new_U, i, U_i_new);
          for (int q = 0; q < K; ++q)
            r[(size_t)i * K + q] = F_iH[q];
          const auto relaxed = limiter.bounds(m_i * measure_of_omega_inverse);
          for (int q = 0; q < NB; ++q)
            bounds[(size_t)i * NB + q] = relaxed[q];
        }
      }
      do_exchange(EX_R, r.data(), K);
      if (discontinuous_ansatz) /* the bounds are extended over the stencil below: ghost range (:603-612) */
        do_exchange(EX_BOUNDS, bounds.data(), NB);

      /* Step 5 */
      const int n_iterations = params.limiter_iterations;
      if (n_iterations != 0 && discontinuous_ansatz) {
        /* Extend the bounds over the stencil (:938-948) with Limiter::combine_bounds AS WRITTEN
         * (shallow_water/limiter.h:386-397): (min h_min, max h_max, min h_small, max(k_max_l, H_MAX_r) -- sic: the
         * fourth entry takes the water-depth bound of the right argument, not its kinetic-energy bound --,
         * max v2_max). The left argument is the running result, the right one a neighbour's ORIGINAL bounds (the
         * reference combines in place while other threads read; see hyperbolic_module.hpp). */
        const std::vector<double> original(bounds);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          double b[NB];
          for (int q = 0; q < NB; ++q)
            b[q] = original[(size_t)i * NB + q];
          for (uint64_t e = rs + 1; e < re; ++e) {
            const double *right = &original[(size_t)csr.col[e] * NB];
            b[0] = std::min(b[0], right[0]);
            b[1] = std::max(b[1], right[1]);
            b[2] = std::min(b[2], right[2]);
            b[3] = std::max(b[3], right[1]); /* sic */
            b[4] = std::max(b[4], right[4]);
          }
          for (int q = 0; q < NB; ++q)
            bounds[(size_t)i * NB + q] = b[q];
        }
      }
      if (n_iterations != 0) {
#pragma omp parallel
        {
          shallow_water::Limiter<dim> limiter(view, params);
          limiter.expensive_bounds_check = expensive_bounds_check;
#pragma omp for schedule(static)
          for (uint32_t i = 0; i < n_owned; ++i) {
            const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
            if (re - rs == 1)
              continue;
            typename shallow_water::Limiter<dim>::Bounds bnd;
            for (int q = 0; q < NB; ++q)
              bnd[q] = bounds[(size_t)i * NB + q];
            const double m_i_inv = mi_inv[i];
            const auto U_i_new = get_state(new_U, i);
            const auto F_iH = get_state(r, i);
            const double factor = tau * m_i_inv * double(re - rs - 1);
            for (uint64_t e = rs + 1; e < re; ++e) {
              const uint32_t j = csr.col[e];
              const auto F_jH = get_state(r, j);
              double b_ij, b_ji;
              if (discontinuous_ansatz) { /* full consistent mass matrix inverse (:976-986) */
                b_ij = mi[i] * mass_matrix_inverse[e] - 0.;
                b_ji = mi[j] * mass_matrix_inverse[e] - 0.;
              } else { /* Neumann series (:988-996) */
                b_ij = 0. - mij[e] * mi_inv[j];
                b_ji = 0. - mij[e] * m_i_inv;
              }
              state_type P_ij;
              for (int q = 0; q < K; ++q) {
                P_ij[q] = pij[e * K + q];
                P_ij[q] += b_ij * F_jH[q] - b_ji * F_iH[q];
                P_ij[q] *= factor;
                pij[e * K + q] = P_ij[q];
              }
              const auto [l_ij, success] = limiter.limit(bnd, U_i_new, P_ij);
              lij[e] = l_ij;
              if (!success)
                restart_needed = true;
            }
          }
        }
        do_exchange(EX_LIJ, lij.data(), 1);
      }

      /* Steps 6, 7 */
      for (int pass = 0; pass < n_iterations; ++pass) {
        const bool last_round = (pass + 1 == n_iterations);
        if (n_iterations == 2 && last_round)
          std::swap(lij, lij_next);
#pragma omp parallel
        {
          shallow_water::Limiter<dim> limiter(view, params);
          limiter.expensive_bounds_check = expensive_bounds_check;
          std::vector<double> lij_row;
#pragma omp for schedule(static)
          for (uint32_t i = 0; i < n_owned; ++i) {
            const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
            if (re - rs == 1)
              continue;
            auto U_i_new = get_state(new_U, i);
            const double lambda = 1. / double(re - rs - 1);
            lij_row.resize(re - rs);
            for (uint64_t e = rs + 1; e < re; ++e) {
              const double l_ij = std::min(lij[e], lij[csr.transpose[e]]);
              for (int q = 0; q < K; ++q)
                U_i_new[q] += l_ij * lambda * pij[e * K + q];
              if (!last_round)
                lij_row[e - rs] = l_ij;
            }
            if (expensive_bounds_check && !view.is_admissible(U_i_new))
              restart_needed = true;
            put_state(new_U, i, U_i_new);
            if (last_round)
              continue;
            typename shallow_water::Limiter<dim>::Bounds bnd;
            for (int q = 0; q < NB; ++q)
              bnd[q] = bounds[(size_t)i * NB + q];
            for (uint64_t e = rs + 1; e < re; ++e) {
              const double old_l_ij = lij_row[e - rs];
              state_type new_p_ij;
              for (int q = 0; q < K; ++q)
                new_p_ij[q] = (1. - old_l_ij) * pij[e * K + q];
              const auto [new_l_ij, success] = limiter.limit(bnd, U_i_new, new_p_ij);
              if (expensive_bounds_check && !success)
                restart_needed = true;
              lij_next[e] = (1. - old_l_ij) * new_l_ij;
            }
          }
        }
        if (!last_round)
          do_exchange(EX_LIJ_NEXT, lij_next.data(), 1);
      }

      double flag = restart_needed.load() ? 1. : 0.;
      do_exchange(EX_OR, &flag, 1);
      *tau_out = tau;
      if (flag != 0.) {
        if (params.id_violation_strategy == RYUJIN_IDV_WARN) {
          n_warnings++;
          return RYUJIN_WARN;
        }
        n_restarts++;
        return RYUJIN_RESTART;
      }
      return RYUJIN_OK;
    }

    void sadd(int h_dst, double s, double b, int h_src) override
    {
      auto &dst = states.at(h_dst).U;
      const auto &src = states.at(h_src).U;
#pragma omp parallel for schedule(static)
      for (size_t q = 0; q < dst.size(); ++q)
        dst[q] = s * dst[q] + b * src[q];
    }

    int debug_fetch(int what, double *out, size_t n) override
    {
      const uint64_t nnz_owned = csr.ptr[n_owned];
      const std::vector<double> *src = nullptr;
      size_t count = 0;
      switch (what) {
      case 0: src = &dij; count = nnz_owned; break;
      case 1: src = &lij; count = nnz_owned; break;
      case 2: src = &pij; count = nnz_owned * K; break;
      case 3: src = &bounds; count = (size_t)n_owned * NB; break;
      case 4: src = &r; count = (size_t)n_owned * K; break;
      case 5: src = &lij_next; count = nnz_owned; break;
      /* over all locally relevant rows (ghost rows / ghost range included): multi-rank parity tests */
      case 6: src = &dij; count = csr.ptr[n_relevant]; break;
      case 7: src = &lij; count = csr.ptr[n_relevant]; break;
      case 8: src = &lij_next; count = csr.ptr[n_relevant]; break;
      case 9: src = &r; count = (size_t)n_relevant * K; break;
      default: return RYUJIN_ERR_ARG;
      }
      if (n < count)
        return RYUJIN_ERR_ARG;
      std::copy(src->begin(), src->begin() + count, out);
      return RYUJIN_OK;
    }
  };
} // namespace oracle

// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// HyperbolicModule::prepare_state_vector / ::step<stages> (source/hyperbolic_module.template.h:96-193,
// 234-1211) instantiated for the EulerAEOS Description (source/euler_aeos/): four precomputed values
// (p, gamma_min, s, eta), two precomputation cycles, fluxes from the precomputed EOS pressure, four
// limiter bounds. Same loop structure as EulerModule (hyperbolic_module.hpp).
//
// Parity status: see euler_aeos.hpp.

#pragma once

#include "euler_aeos.hpp"
#include "hyperbolic_module.hpp"

namespace oracle
{
  template <int dim>
  struct EulerAeosModule final : ModuleBase {
    using V = aeos::View<dim>;
    static constexpr int K = dim + 2;
    static constexpr int NB = 4;
    static constexpr int NP = 4; /* p, gamma_min, s, eta */
    using state_type = typename V::state_type;
    using vec_type = std::array<double, dim>;

    V view;
    CSR csr;
    std::vector<double> cij, mij, mi, mi_inv;
    double measure_of_omega;
    /* discontinuous ansatz (hyperbolic_module.template.h:287-293): incidence matrix, full inverse mass matrix */
    bool discontinuous_ansatz = false;
    std::vector<double> incidence, mass_matrix_inverse;

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

    /* module-owned scratch (hyperbolic_module.h:319-333) */
    std::vector<double> bounds, r, dij, lij, lij_next, pij;

    int k() const override { return K; }
    int n_prec() const override { return NP; }
    int n_bounds() const override { return NB; }

    EulerAeosModule(const ryujin_hip_offline &o, const ryujin_hip_params &p)
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
      measure_of_omega = o.measure_of_omega;
      discontinuous_ansatz = o.discontinuous_ansatz != 0;
      if (discontinuous_ansatz) {
        if (!o.incidence || !o.mass_matrix_inverse)
          throw std::runtime_error("discontinuous ansatz without incidence / inverse mass matrix");
        incidence = csr.gather(o, o.incidence, 1);
        mass_matrix_inverse = csr.gather(o, o.mass_matrix_inverse, 1);
      }
      b_i.assign(o.b_i, o.b_i + o.n_bdry);
      b_normal.assign(o.b_normal, o.b_normal + (size_t)o.n_bdry * dim);
      b_id.assign(o.b_id, o.b_id + o.n_bdry);
      dirichlet.assign((size_t)o.n_bdry * K, 0.);
      p_i.assign(o.p_i, o.p_i + o.n_pairs);
      p_col.assign(o.p_col, o.p_col + o.n_pairs);
      p_j.assign(o.p_j, o.p_j + o.n_pairs);

      /* prepare(): hyperbolic_module.template.h:52-86 */
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
      states.back().prec.assign((size_t)n_relevant * NP, 0.);
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
    static typename V::precomputed_type get_prec(const std::vector<double> &prec, uint32_t i)
    {
      typename V::precomputed_type p;
      for (int q = 0; q < NP; ++q)
        p[q] = prec[(size_t)i * NP + q];
      return p;
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

    /* ---- Step 1: hyperbolic_module.template.h:96-193 --------------------- */
    void prepare_state_vector(int h, double /*t*/, const double *dirichlet_in) override
    {
      auto &U = states.at(h).U;
      auto &prec = states.at(h).prec;
      if (dirichlet_in)
        dirichlet.assign(dirichlet_in, dirichlet_in + dirichlet.size());

      /* serial loop over the boundary map (:123-144) */
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
        const auto U_i = get_state(U, i);
        put_state(U, i, view.apply_boundary_conditions(id, U_i, normal, U_D));
      }

      do_exchange(EX_U, U.data(), K); /* :148 */

      /* precomputation_loop, cycle 0: source/euler_aeos/hyperbolic_system.h:919-938
       * (the prefer_vector_interface() branch :865-917 computes the same values) */
#pragma omp parallel for schedule(static)
      for (uint32_t i = 0; i < n_owned; ++i) {
        if (csr.ptr[i + 1] - csr.ptr[i] == 1)
          continue;
        const auto U_i = get_state(U, i);
        const double p_i = view.eos_pressure_of_state(U_i);
        const double gamma_i = view.surrogate_gamma(U_i, p_i);
        prec[(size_t)i * NP + 0] = p_i;
        prec[(size_t)i * NP + 1] = gamma_i;
        prec[(size_t)i * NP + 2] = 0.;
        prec[(size_t)i * NP + 3] = 0.;
      }
      do_exchange(EX_PREC, prec.data(), NP); /* hyperbolic_module.template.h:155-158 */

      /* cycle 1 (:942-975): gamma_min over the stencil (gamma_j recomputed from U_j and p_j: the
       * neighbour's slot 1 may already hold ITS minimum), then surrogate s_i and eta_i.
       * Written to a copy so that the result does not depend on the thread schedule. */
      {
        std::vector<double> out(prec);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          const auto U_i = get_state(U, i);
          double gamma_min_i = prec[(size_t)i * NP + 1];
          for (uint64_t e = rs + 1; e < re; ++e) {
            const uint32_t j = csr.col[e];
            const auto U_j = get_state(U, j);
            const double p_j = prec[(size_t)j * NP + 0];
            const double gamma_j = view.surrogate_gamma(U_j, p_j);
            gamma_min_i = std::min(gamma_min_i, gamma_j);
          }
          out[(size_t)i * NP + 1] = gamma_min_i;
          out[(size_t)i * NP + 2] = view.surrogate_specific_entropy(U_i, gamma_min_i);
          out[(size_t)i * NP + 3] = view.surrogate_harten_entropy(U_i, gamma_min_i);
        }
        prec.swap(out);
      }
      do_exchange(EX_PREC, prec.data(), NP);
    }

    /* ---- Steps 2-7: hyperbolic_module.template.h:234-1211 ----------------- */
    int step(int h_old, int stages, const int *h_stage, const double *w, int h_new, double tau,
             double tau_max_in, double *tau_out) override
    {
      const auto &old_U = states.at(h_old).U;
      const auto &old_prec = states.at(h_old).prec;
      auto &new_U = states.at(h_new).U;

      const double measure_of_omega_inverse = 1. / measure_of_omega;
      std::atomic<bool> restart_needed{false};

      const aeos::RiemannSolver riemann_solver(view.b(), view.pinf(), view.compute_strict_bounds);

      /* Step 2: d_ij (upper triangle) and alpha_i  (:341-424) */
#pragma omp parallel
      {
        aeos::Indicator<dim> indicator(view, params);
#pragma omp for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          const auto U_i = get_state(old_U, i);
          indicator.reset(U_i, get_prec(old_prec, i));
          for (uint64_t e = rs; e < re; ++e) {
            const uint32_t j = csr.col[e];
            const auto U_j = get_state(old_U, j);
            const auto c_ij = get_c(e);
            indicator.accumulate(U_j, c_ij);
            if (e == rs)
              continue;
            if (j < i)
              continue;
            double norm2 = 0.;
            for (int d = 0; d < dim; ++d)
              norm2 += c_ij[d] * c_ij[d];
            const double norm = std::sqrt(norm2);
            vec_type n_ij;
            const double inverse_norm = 1. / norm; /* dealii::Tensor / scalar multiplies by the inverse */
            for (int d = 0; d < dim; ++d)
              n_ij[d] = c_ij[d] * inverse_norm;
            const double lambda_max = riemann_solver.template compute<dim>(view, U_i, old_prec[(size_t)i * NP], U_j, old_prec[(size_t)j * NP], n_ij);
            dij[e] = norm * lambda_max;
          }
          const double hd_i = mi[i] * measure_of_omega_inverse;
          alpha[i] = indicator.alpha(hd_i);
        }
      }
      do_exchange(EX_ALPHA, alpha.data(), 1); /* :344-347 */

      /* Step 3: boundary d_ij, symmetrise, diagonal, tau_max  (:432-564) */
      for (size_t q = 0; q < p_i.size(); ++q) {
        const uint32_t i = p_i[q], col_idx = p_col[q], j = p_j[q];
        if (j < i)
          continue;
        const auto U_i = get_state(old_U, i);
        const auto U_j = get_state(old_U, j);
        const uint64_t e = csr.ptr[i] + col_idx;
        const uint64_t et = csr.transpose[e];
        const auto c_ji = get_c(et);
        double norm2 = 0.;
        for (int d = 0; d < dim; ++d)
          norm2 += c_ji[d] * c_ji[d];
        const double norm_ji = std::sqrt(norm2);
        vec_type n_ji;
        const double inverse_norm_ji = 1. / norm_ji; /* dealii::Tensor / scalar multiplies by the inverse */
        for (int d = 0; d < dim; ++d)
          n_ji[d] = c_ji[d] * inverse_norm_ji;
        const double d_ij = dij[e];
        const double lambda_max = riemann_solver.template compute<dim>(view, U_j, old_prec[(size_t)j * NP], U_i, old_prec[(size_t)i * NP], n_ji);
        const double d_ji = norm_ji * lambda_max;
        dij[e] = std::max(d_ij, d_ji);
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
            const uint32_t j = csr.col[e];
            if (j < i)
              dij[e] = dij[csr.transpose[e]];
            d_sum -= dij[e];
          }
          d_sum = std::min(d_sum, -1.e6 * std::numeric_limits<double>::min());
          dij[rs] = d_sum;
          const double tau_i = params.cfl * mi[i] / (-2. * d_sum);
          local_tau_max = std::min(local_tau_max, tau_i);
        }
        tau_max = std::min(tau_max, local_tau_max);
      }
      do_exchange(EX_MIN, &tau_max, 1); /* :571 */

      if (std::isnan(tau_max) || std::isinf(tau_max) || !(tau_max > 0.)) /* :573-576 */
        return RYUJIN_ERR_TAU;

      tau = (tau == 0. ? tau_max : tau); /* :578 */

      /* Step 4: low-order update, bounds, r_i, p_ij (:597-884) */
      double weight = 1.;
      for (int s = 0; s < stages; ++s)
        weight -= w[s]; /* -accumulate(w, -1.) */
      {
        /* reference: -std::accumulate(begin, end, -1.) = -((-1 + w0) + w1 ...) */
        double acc = -1.;
        for (int s = 0; s < stages; ++s)
          acc += w[s];
        weight = -acc;
      }

#pragma omp parallel
      {
        aeos::Limiter<dim> limiter(view, params);
#pragma omp for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;

          const auto U_i = get_state(old_U, i);
          auto U_i_new = U_i;
          const double alpha_i = alpha[i];
          const double m_i = mi[i];
          const double m_i_inv = mi_inv[i];
          const auto flux_i = V::f(U_i, old_prec[(size_t)i * NP]); /* flux_contribution: f(U_i, p_i) */

          std::array<typename V::flux_type, 4> flux_iHs;
          for (int s = 0; s < stages; ++s)
            flux_iHs[s] = V::f(get_state(states.at(h_stage[s]).U, i), states.at(h_stage[s]).prec[(size_t)i * NP]);

          state_type F_iH;
          F_iH.fill(0.);
          limiter.reset(U_i, flux_i, old_prec[(size_t)i * NP + 1]);

          for (uint64_t e = rs; e < re; ++e) {
            const uint32_t j = csr.col[e];
            const auto U_j = get_state(old_U, j);
            const double alpha_j = alpha[j];
            const double d_ij = dij[e];
            double factor = (alpha_i + alpha_j) * .5;
            if (discontinuous_ansatz) /* hyperbolic_module.template.h:733-737 */
              factor = std::max(factor, incidence[e]);
            const double d_ijH = d_ij * factor;

            const auto c_ij = get_c(e);
            const double regularization = 100. * std::numeric_limits<double>::min();
            vec_type scaled_c_ij;
            const double denom = std::max(d_ij, regularization);
            const double inverse_denom = 1. / denom; /* dealii::Tensor / scalar multiplies by the inverse */
            for (int d = 0; d < dim; ++d)
              scaled_c_ij[d] = c_ij[d] * inverse_denom;

            const auto flux_j = V::f(U_j, old_prec[(size_t)j * NP]);
            const auto flux_ij = V::flux_divergence(flux_i, flux_j, c_ij);

            state_type P_ij;
            for (int q = 0; q < K; ++q) {
              U_i_new[q] += tau * m_i_inv * flux_ij[q];
              P_ij[q] = -flux_ij[q];
            }
            for (int q = 0; q < K; ++q) {
              const double dU = U_j[q] - U_i[q];
              U_i_new[q] += tau * m_i_inv * d_ij * dU;
              F_iH[q] += d_ijH * dU;
              P_ij[q] += (d_ijH - d_ij) * dU;
            }
            limiter.accumulate(U_j, flux_j, scaled_c_ij, old_prec[(size_t)j * NP + 2]);

            for (int q = 0; q < K; ++q) {
              F_iH[q] += weight * flux_ij[q];
              P_ij[q] += weight * flux_ij[q];
            }

            for (int s = 0; s < stages; ++s) {
              const auto U_jHs = get_state(states.at(h_stage[s]).U, j);
              const auto flux_jHs = V::f(U_jHs, states.at(h_stage[s]).prec[(size_t)j * NP]);
              const auto flux_ij_s = V::flux_divergence(flux_iHs[s], flux_jHs, c_ij);
              for (int q = 0; q < K; ++q) {
                F_iH[q] += w[s] * flux_ij_s[q];
                P_ij[q] += w[s] * flux_ij_s[q];
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

          const double hd_i = m_i * measure_of_omega_inverse;
          const auto relaxed_bounds = limiter.bounds(hd_i);
          for (int q = 0; q < NB; ++q)
            bounds[(size_t)i * NB + q] = relaxed_bounds[q];
        }
      }
      do_exchange(EX_R, r.data(), K); /* :601-613 */
      if (discontinuous_ansatz) /* the bounds are extended over the stencil below: ghost range (:603-612) */
        do_exchange(EX_BOUNDS, bounds.data(), NB);

      /* Step 5: second part of p_ij, first l_ij (:892-1041) */
      const int n_iterations = params.limiter_iterations;
      if (n_iterations != 0 && discontinuous_ansatz) {
        /* extend the bounds over the stencil (:938-948; Limiter::combine_bounds, euler_aeos/limiter.h:435-445: min,
         * max, min, min), every row from the ORIGINAL bounds of its stencil (see hyperbolic_module.hpp) */
        const std::vector<double> original(bounds);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n_owned; ++i) {
          const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
          if (re - rs == 1)
            continue;
          double b0 = original[(size_t)i * NB], b1 = original[(size_t)i * NB + 1], b2 = original[(size_t)i * NB + 2],
                 b3 = original[(size_t)i * NB + 3];
          for (uint64_t e = rs + 1; e < re; ++e) {
            const uint32_t j = csr.col[e];
            b0 = std::min(b0, original[(size_t)j * NB]);
            b1 = std::max(b1, original[(size_t)j * NB + 1]);
            b2 = std::min(b2, original[(size_t)j * NB + 2]);
            b3 = std::min(b3, original[(size_t)j * NB + 3]);
          }
          bounds[(size_t)i * NB] = b0;
          bounds[(size_t)i * NB + 1] = b1;
          bounds[(size_t)i * NB + 2] = b2;
          bounds[(size_t)i * NB + 3] = b3;
        }
      }
      if (n_iterations != 0) {
#pragma omp parallel
        {
          aeos::Limiter<dim> limiter(view, params);
          limiter.expensive_bounds_check = expensive_bounds_check;
#pragma omp for schedule(static)
          for (uint32_t i = 0; i < n_owned; ++i) {
            const uint64_t rs = csr.ptr[i], re = csr.ptr[i + 1];
            if (re - rs == 1)
              continue;
            typename aeos::Limiter<dim>::Bounds bnd;
            for (int q = 0; q < NB; ++q)
              bnd[q] = bounds[(size_t)i * NB + q];
            const double m_i_inv = mi_inv[i];
            const auto U_i_new = get_state(new_U, i);
            const auto F_iH = get_state(r, i);
            const double lambda_inv = double(re - rs - 1);
            const double factor = tau * m_i_inv * lambda_inv;

            for (uint64_t e = rs + 1; e < re; ++e) {
              const uint32_t j = csr.col[e];
              state_type P_ij;
              for (int q = 0; q < K; ++q)
                P_ij[q] = pij[e * K + q];
              const auto F_jH = get_state(r, j);

              const double kronecker_ij = 0.;
              double b_ij, b_ji;
              if (discontinuous_ansatz) { /* full consistent mass matrix inverse (:976-986) */
                const double m_ij_inv = mass_matrix_inverse[e];
                b_ij = mi[i] * m_ij_inv - kronecker_ij;
                b_ji = mi[j] * m_ij_inv - kronecker_ij;
              } else { /* Neumann series expansion (:988-996) */
                const double m_j_inv = mi_inv[j];
                const double m_ij = mij[e];
                b_ij = kronecker_ij - m_ij * m_j_inv;
                b_ji = kronecker_ij - m_ij * m_i_inv;
              }
              for (int q = 0; q < K; ++q) {
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
        do_exchange(EX_LIJ, lij.data(), 1); /* :895-898 */
      }

      /* Steps 6, 7: symmetrise l_ij, high-order update, next l_ij (:1053-1182) */
      for (int pass = 0; pass < n_iterations; ++pass) {
        const bool last_round = (pass + 1 == n_iterations);
        if (n_iterations == 2 && last_round)
          std::swap(lij, lij_next);

#pragma omp parallel
        {
          aeos::Limiter<dim> limiter(view, params);
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

            typename aeos::Limiter<dim>::Bounds bnd;
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
          do_exchange(EX_LIJ_NEXT, lij_next.data(), 1); /* :1066-1071 */
      }

      /* restart? (:1190-1207) */
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

    /* time_integrator.template.h:18-25 */
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

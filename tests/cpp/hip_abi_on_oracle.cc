// TEST DOUBLE (tests/test_binding_run.py, CPU leg): the part of the C ABI of libryujin_hip.so that the adapter
// contrib/hyperbolic_module_hip.h calls when it serves an UNMODIFIED TimeIntegrator -- create, state vectors,
// prepare_state_vector, step, accessors -- forwarded to the CPU oracle (oracle/oracle_capi.cc exports the same call
// surface with the prefix ryujin_oracle_). It lets the host logic of the adapter (twins that follow the storage through
// StateVector::swap, what is uploaded and written back when) run and be checked in a container without a GPU. Linked
// into tests/cpp/time_integrator_run INSTEAD of libryujin_hip.so, only by that test; nothing in the product can reach
// it. The device-resident driver (ryujin_hip_time_step_fn) has no counterpart here: it refuses.
#include <ryujin_hip.h>

#include <cstring>
#include <string>
#include <vector>

extern "C" {
void ryujin_oracle_default_params(ryujin_hip_params *, int, int);
int ryujin_oracle_create(void **, const ryujin_hip_offline *, const ryujin_hip_params *, void *, int);
void ryujin_oracle_destroy(void *);
int ryujin_oracle_state_alloc(void *, int *);
int ryujin_oracle_state_free(void *, int);
int ryujin_oracle_state_upload(void *, int, const double *);
int ryujin_oracle_state_download(void *, int, double *);
int ryujin_oracle_state_download_precomputed(void *, int, double *);
int ryujin_oracle_prepare_state_vector(void *, int, double, const double *);
int ryujin_oracle_step(void *, int, int, const int *, const double *, int, double, double, double *);
int ryujin_oracle_set_cfl(void *, double);
int ryujin_oracle_get_cfl(void *, double *);
int ryujin_oracle_set_id_violation_strategy(void *, int);
int ryujin_oracle_get_alpha(void *, double *);
int ryujin_oracle_get_counters(void *, unsigned *, unsigned *);
const char *ryujin_oracle_last_error(void);
}

/* what the double needs to know about the mesh to write back the rows the real entry points write back */
struct ryujin_hip_ctx {
  void *oracle = nullptr;
  unsigned n_owned = 0, n_relevant = 0;
  int k = 0;
  std::vector<unsigned> bc_rows;
  std::vector<double> scratch;
  unsigned long long bytes_up = 0, bytes_down = 0;
};

static std::string g_error;

extern "C" {

void ryujin_hip_default_params(ryujin_hip_params *p, int equation, int dim) { ryujin_oracle_default_params(p, equation, dim); }

int ryujin_hip_device_count(int *n)
{
  *n = 1;
  return RYUJIN_OK;
}

int ryujin_hip_create(ryujin_hip_ctx **ctx, const ryujin_hip_offline *o, const ryujin_hip_params *p, ryujin_hip_comm *, int)
{
  auto *c = new ryujin_hip_ctx;
  const int rc = ryujin_oracle_create(&c->oracle, o, p, nullptr, 0);
  if (rc < 0) {
    delete c;
    return rc;
  }
  c->n_owned = o->n_owned;
  c->n_relevant = o->n_relevant;
  c->k = p->equation == RYUJIN_EQ_SHALLOW_WATER ? p->dim + 1 : p->equation == RYUJIN_EQ_SCALAR_CONSERVATION ? 1 : p->dim + 2;
  for (unsigned q = 0; q < o->n_bdry; ++q)
    c->bc_rows.push_back(o->b_i[q]);
  *ctx = c;
  return RYUJIN_OK;
}

void ryujin_hip_destroy(ryujin_hip_ctx *c)
{
  if (c) {
    ryujin_oracle_destroy(c->oracle);
    delete c;
  }
}

void ryujin_hip_comm_destroy(ryujin_hip_comm *) {}
int ryujin_hip_comm_unique_id(char *) { return RYUJIN_ERR_UNSUPPORTED; }
int ryujin_hip_comm_init(ryujin_hip_comm **, const char *, int, int, int) { return RYUJIN_ERR_UNSUPPORTED; }

int ryujin_hip_state_alloc(ryujin_hip_ctx *c, int *h) { return ryujin_oracle_state_alloc(c->oracle, h); }
int ryujin_hip_state_free(ryujin_hip_ctx *c, int h) { return ryujin_oracle_state_free(c->oracle, h); }
int ryujin_hip_state_upload(ryujin_hip_ctx *c, int h, const double *U)
{
  c->bytes_up += 8ull * c->n_relevant * c->k;
  return ryujin_oracle_state_upload(c->oracle, h, U);
}
int ryujin_hip_state_download(ryujin_hip_ctx *c, int h, double *U)
{
  c->bytes_down += 8ull * c->n_relevant * c->k;
  return ryujin_oracle_state_download(c->oracle, h, U);
}
int ryujin_hip_state_download_precomputed(ryujin_hip_ctx *c, int h, double *prec)
{
  return ryujin_oracle_state_download_precomputed(c->oracle, h, prec);
}
int ryujin_hip_host_register(ryujin_hip_ctx *, const void *ptr, size_t bytes) { return ptr && bytes ? RYUJIN_OK : RYUJIN_ERR_ARG; }
int ryujin_hip_host_unregister(ryujin_hip_ctx *, const void *) { return RYUJIN_OK; }

/* as the real ones: ONLY the owned rows / ONLY the boundary rows and the ghost range are written */
int ryujin_hip_state_download_owned(ryujin_hip_ctx *c, int h, double *U)
{
  c->scratch.resize((size_t)c->n_relevant * c->k);
  const int rc = ryujin_oracle_state_download(c->oracle, h, c->scratch.data());
  std::memcpy(U, c->scratch.data(), sizeof(double) * c->n_owned * c->k);
  c->bytes_down += 8ull * c->n_owned * c->k;
  return rc;
}
int ryujin_hip_state_download_prepared(ryujin_hip_ctx *c, int h, double *U)
{
  c->scratch.resize((size_t)c->n_relevant * c->k);
  const int rc = ryujin_oracle_state_download(c->oracle, h, c->scratch.data());
  for (const unsigned row : c->bc_rows)
    std::memcpy(U + (size_t)row * c->k, c->scratch.data() + (size_t)row * c->k, sizeof(double) * c->k);
  std::memcpy(U + (size_t)c->n_owned * c->k, c->scratch.data() + (size_t)c->n_owned * c->k,
              sizeof(double) * (c->n_relevant - c->n_owned) * c->k);
  c->bytes_down += 8ull * (c->bc_rows.size() + c->n_relevant - c->n_owned) * c->k;
  return rc;
}

int ryujin_hip_prepare_state_vector(ryujin_hip_ctx *c, int h, double t, const double *dirichlet)
{
  return ryujin_oracle_prepare_state_vector(c->oracle, h, t, dirichlet);
}
int ryujin_hip_step(ryujin_hip_ctx *c, int h_old, int stages, const int *h_stage, const double *w, int h_new, double tau,
                    double tau_max, double *tau_out)
{
  return ryujin_oracle_step(c->oracle, h_old, stages, h_stage, w, h_new, tau, tau_max, tau_out);
}
int ryujin_hip_time_step_fn(ryujin_hip_ctx *, int, int, int, const int *, double, ryujin_hip_dirichlet_fn, void *, double, int,
                            double, double, double *)
{
  g_error = "test double: the device-resident Runge-Kutta driver exists in libryujin_hip.so only";
  return RYUJIN_ERR_UNSUPPORTED;
}
int ryujin_hip_set_cfl(ryujin_hip_ctx *c, double cfl) { return ryujin_oracle_set_cfl(c->oracle, cfl); }
int ryujin_hip_get_cfl(ryujin_hip_ctx *c, double *cfl) { return ryujin_oracle_get_cfl(c->oracle, cfl); }
int ryujin_hip_set_id_violation_strategy(ryujin_hip_ctx *c, int s) { return ryujin_oracle_set_id_violation_strategy(c->oracle, s); }
int ryujin_hip_get_alpha(ryujin_hip_ctx *c, double *alpha) { return ryujin_oracle_get_alpha(c->oracle, alpha); }
int ryujin_hip_get_counters(ryujin_hip_ctx *c, unsigned *r, unsigned *w) { return ryujin_oracle_get_counters(c->oracle, r, w); }
const char *ryujin_hip_last_error(void) { return g_error.empty() ? ryujin_oracle_last_error() : g_error.c_str(); }
}

// Model traits: thin static wrappers over the generated one-step maps (models_gen.h).
// A model is selected at compile time (template parameter of every kernel), so the
// per-step arithmetic is fully inlined into the scan loops.
//
// Reference: sde/example_models/fhn.py:10-65, sde/example_models/sir.py:9-93.
#pragma once
#include "models_gen.h"

namespace chmc {

struct FhnModel {
  static constexpr int ID = 0, X = CHMC_FHN_X, V = CHMC_FHN_V, Z = CHMC_FHN_Z, V0 = CHMC_FHN_V0, NK = CHMC_FHN_NK;
  static constexpr int NABS = 0;     // no absorbing components
  CHMC_HD static bool absorbed(double) { return false; }
  CHMC_HD static double abs_value() { return 0.0; }
  static constexpr int NXI = X + V + Z;
  static constexpr int U = Z;        // dim_u: the global parameters (fixed observation noise)
  static constexpr bool VS = false;  // variable observation noise
  CHMC_HD static void precompute(const double* z, double dl, double* k) { chmc_fhn_precompute(z, dl, k); }
  CHMC_HD static void step(const double* k, const double* x, const double* v, double* xn) { chmc_fhn_step(k, x, v, xn); }
  CHMC_HD static void jac(const double* k, const double* x, const double* v, double* A, double* B, double* Zf) {
    chmc_fhn_jac(k, x, v, A, B, Zf);
  }
  CHMC_HD static void jac_ab(const double* k, const double* x, const double* v, double* A, double* B) {
    chmc_fhn_jac_ab(k, x, v, A, B);
  }
  CHMC_HD static void hess(const double* k, const double* x, const double* v, const double* S, double* out) {
    chmc_fhn_step_hess(k, x, v, S, out);
  }
  CHMC_HD static void gz(const double* u, double* z) { chmc_fhn_gz(u, z); }
  CHMC_HD static void gz_jac(const double* u, double* G) { chmc_fhn_gz_jac(u, G); }
  CHMC_HD static void gz_hess(const double* u, const double* ud, const double* zb, double* o) { chmc_fhn_gz_hess(u, ud, zb, o); }
  CHMC_HD static void gx0(const double* z, const double* v0, double* x0) { chmc_fhn_gx0(z, v0, x0); }
  CHMC_HD static void gx0_jac(double* dz, double* dv0) { chmc_fhn_gx0_jac(dz, dv0); }
  CHMC_HD static double obs(const double* x) { return chmc_fhn_obs(x); }
  CHMC_HD static void obs_grad(const double* x, double* g) { chmc_fhn_obs_grad(x, g); }
  CHMC_HD static void obs_hess_vec(const double* x, const double* xd, double* o) { chmc_fhn_obs_hess_vec(x, xd, o); }
};

// FitzHugh-Nagumo with the priors of the reference's notebook (FitzHugh-Nagumo_example.ipynb cells 7-18): same
// one-step map, different generate_z / generate_x_0.
struct FhnNbModel {
  static constexpr int ID = 2, X = CHMC_FHNNB_X, V = CHMC_FHNNB_V, Z = CHMC_FHNNB_Z, V0 = CHMC_FHNNB_V0, NK = CHMC_FHNNB_NK;
  static constexpr int NABS = 0;     // no absorbing components
  CHMC_HD static bool absorbed(double) { return false; }
  CHMC_HD static double abs_value() { return 0.0; }
  static constexpr int NXI = X + V + Z;
  static constexpr int U = Z;        // dim_u: the global parameters (fixed observation noise)
  static constexpr bool VS = false;  // variable observation noise
  CHMC_HD static void precompute(const double* z, double dl, double* k) { chmc_fhnnb_precompute(z, dl, k); }
  CHMC_HD static void step(const double* k, const double* x, const double* v, double* xn) { chmc_fhnnb_step(k, x, v, xn); }
  CHMC_HD static void jac(const double* k, const double* x, const double* v, double* A, double* B, double* Zf) {
    chmc_fhnnb_jac(k, x, v, A, B, Zf);
  }
  CHMC_HD static void jac_ab(const double* k, const double* x, const double* v, double* A, double* B) {
    chmc_fhnnb_jac_ab(k, x, v, A, B);
  }
  CHMC_HD static void hess(const double* k, const double* x, const double* v, const double* S, double* out) {
    chmc_fhnnb_step_hess(k, x, v, S, out);
  }
  CHMC_HD static void gz(const double* u, double* z) { chmc_fhnnb_gz(u, z); }
  CHMC_HD static void gz_jac(const double* u, double* G) { chmc_fhnnb_gz_jac(u, G); }
  CHMC_HD static void gz_hess(const double* u, const double* ud, const double* zb, double* o) { chmc_fhnnb_gz_hess(u, ud, zb, o); }
  CHMC_HD static void gx0(const double* z, const double* v0, double* x0) { chmc_fhnnb_gx0(z, v0, x0); }
  CHMC_HD static void gx0_jac(double* dz, double* dv0) { chmc_fhnnb_gx0_jac(dz, dv0); }
  CHMC_HD static double obs(const double* x) { return chmc_fhnnb_obs(x); }
  CHMC_HD static void obs_grad(const double* x, double* g) { chmc_fhnnb_obs_grad(x, g); }
  CHMC_HD static void obs_hess_vec(const double* x, const double* xd, double* o) { chmc_fhnnb_obs_hess_vec(x, xd, o); }
};

// SIR in (log S, log I, log-contact-rate) coordinates.  The first two components are clipped
// below at -500 before a step and a clipped component keeps its (clipped) value
// (sde/example_models/sir.py:54-70); derivatives through a clipped component are zero.
struct SirModel {
  static constexpr int ID = 1, X = CHMC_SIR_X, V = CHMC_SIR_V, Z = CHMC_SIR_Z, V0 = CHMC_SIR_V0, NK = CHMC_SIR_NK;
  static constexpr int NXI = X + V + Z;
  static constexpr int U = Z;        // dim_u: the global parameters (fixed observation noise)
  static constexpr bool VS = false;  // variable observation noise
  // the first NABS components are absorbing: once a state component is at or below the floor (or NaN: clip() below), every
  // later state has it at abs_value() whatever the other components do (used by the time-parallel scan, k_fwd_par)
  static constexpr int NABS = 2;
  CHMC_HD static bool absorbed(double xa) { return !(xa > -500.0); }
  CHMC_HD static double abs_value() { return -500.0; }
  CHMC_HD static void precompute(const double* z, double dl, double* k) { chmc_sir_precompute(z, dl, k); }
  CHMC_HD static void clip(const double* x, double* xc, bool* fr) {
    for (int a = 0; a < 2; ++a) {
      fr[a] = !(x[a] > -500.0);
      xc[a] = fr[a] ? -500.0 : x[a];
    }
    fr[2] = false;
    xc[2] = x[2];
  }
  CHMC_HD static void step(const double* k, const double* x, const double* v, double* xn) {
    double xc[3];
    bool fr[3];
    clip(x, xc, fr);
    chmc_sir_step(k, xc, v, xn);
    if (fr[0]) xn[0] = xc[0];
    if (fr[1]) xn[1] = xc[1];
  }
  CHMC_HD static void jac(const double* k, const double* x, const double* v, double* A, double* B, double* Zf) {
    double xc[3];
    bool fr[3];
    clip(x, xc, fr);
    chmc_sir_jac(k, xc, v, A, B, Zf);
    for (int a = 0; a < 2; ++a)
      if (fr[a]) {
        for (int c = 0; c < 3; ++c) A[a * 3 + c] = 0.0, A[c * 3 + a] = 0.0, B[a * 3 + c] = 0.0;
        for (int c = 0; c < 4; ++c) Zf[a * 4 + c] = 0.0;
      }
  }
  CHMC_HD static void jac_ab(const double* k, const double* x, const double* v, double* A, double* B) {
    double Zf[X * Z];
    jac(k, x, v, A, B, Zf);
  }
  CHMC_HD static void hess(const double* k, const double* x, const double* v, const double* S, double* out) {
    double xc[3], S2[X * NXI];
    bool fr[3];
    clip(x, xc, fr);
    for (int i = 0; i < X * NXI; ++i) S2[i] = S[i];
    for (int a = 0; a < 2; ++a)
      if (fr[a]) {
        for (int m = 0; m < NXI; ++m) S2[a * NXI + m] = 0.0;
        for (int b = 0; b < X; ++b) S2[b * NXI + a] = 0.0;
      }
    chmc_sir_step_hess(k, xc, v, S2, out);
    for (int a = 0; a < 2; ++a)
      if (fr[a]) out[a] = 0.0;
  }
  CHMC_HD static void gz(const double* u, double* z) { chmc_sir_gz(u, z); }
  CHMC_HD static void gz_jac(const double* u, double* G) { chmc_sir_gz_jac(u, G); }
  CHMC_HD static void gz_hess(const double* u, const double* ud, const double* zb, double* o) { chmc_sir_gz_hess(u, ud, zb, o); }
  CHMC_HD static void gx0(const double* z, const double* v0, double* x0) { chmc_sir_gx0(z, v0, x0); }
  CHMC_HD static void gx0_jac(double* dz, double* dv0) { chmc_sir_gx0_jac(dz, dv0); }
  CHMC_HD static double obs(const double* x) { return chmc_sir_obs(x); }
  CHMC_HD static void obs_grad(const double* x, double* g) { chmc_sir_obs_grad(x, g); }
  CHMC_HD static void obs_hess_vec(const double* x, const double* xd, double* o) { chmc_sir_obs_hess_vec(x, xd, o); }
};

// Variable observation noise: generate_sigma(u) = exp(u[dim_z]) (fhn.py:46-47, sir.py:92-93), one more global parameter.
// Everything that depends on the state dimension only is inherited; the kernels size their dc/du arrays with U and add
// the d(sigma n_i)/du_sigma column and the sigma-dependent Gram diagonal where VS is set.
template <class Base>
struct VarSigma : Base {
  static constexpr int U = Base::Z + 1;
  static constexpr bool VS = true;
};
using FhnVsModel = VarSigma<FhnModel>;
using SirVsModel = VarSigma<SirModel>;

}  // namespace chmc

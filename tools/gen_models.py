#!/usr/bin/env python3
"""Generate C code for the SDE one-step maps and their derivatives (sympy -> C).

Restates, in plain sympy, what the reference obtains by SymNum code generation:

* strong-order-1.5 Taylor step for additive noise  (sde/integrators.py:43-63, 95-149)
* Euler-Maruyama step                              (sde/integrators.py:8-14)
* Ito-lemma change of variables                    (sde/transforms.py:9-63)
* FitzHugh-Nagumo drift / diffusion                (sde/example_models/fhn.py:17-34)
* SIR drift / diffusion, log transform             (sde/example_models/sir.py:19-51)
* generate_z / generate_x_0 / obs_func             (fhn.py:37-51, sir.py:73-93)

For every model it emits `static inline` C functions (usable from plain C, C++ and
HIP device code through the CHMC_HD macro):

  <m>_precompute(z, dl, k)            constants depending on (z, dl) only
  <m>_step(k, x, v, xn)               x_{s+1} = f(z, x_s, v_s)
  <m>_step_jac(k, x, v, xn, A, B, Zf) f and df/dx [X*X], df/dv [X*V], df/dz [X*Z]
  <m>_step_hess(k, x, v, S, out)      out_k = sum_{a,m} S[a][m] d2 f_a / d xi_m d xi_k,
                                      xi = (x, v, z), S is X x (X+V+Z) row-major
  <m>_gz(u, z), <m>_gz_jac(u, G), <m>_gz_hess(u, ud, zb, out)
  <m>_gx0(z, v0, x0), <m>_gx0_jac(dz, dv0)   (affine in (z, v0) for both models)
  <m>_obs(x), <m>_obs_grad(x, g), <m>_obs_hess_vec(x, xd, out)

The file written is committed, so neither the GPU box nor the tests need sympy to
build.  Run:  python tools/gen_models.py
"""
import os
import sys
import sympy as sp
from sympy.printing.c import C99CodePrinter

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ----------------------------------------------------------------------------- printer
class Printer(C99CodePrinter):
    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer:
            n = int(e)
            bs = self._print(b)
            if not (b.is_Symbol or b.is_Number):
                bs = "(" + bs + ")"
            if 1 <= n <= 6:
                return "(" + "*".join([bs] * n) + ")"
            if -4 <= n <= -1:
                return "(1.0/(" + "*".join([bs] * (-n)) + "))"
        if e == sp.Rational(1, 2):
            return "sqrt(%s)" % self._print(b)
        if e == sp.Rational(-1, 2):
            return "(1.0/sqrt(%s))" % self._print(b)
        if e == sp.Rational(3, 2):
            bs = self._print(b)
            return "((%s)*sqrt(%s))" % (bs, bs)
        return super()._print_Pow(expr)

    def _print_Rational(self, expr):
        return "(%d.0/%d.0)" % (expr.p, expr.q)

    def _print_Integer(self, expr):
        return "%d.0" % int(expr)


PR = Printer()


def cc(e):
    return PR.doprint(e)


# ----------------------------------------------------------------------------- hoisting
class Hoister:
    """Pull sub-expressions that depend on (z, dl) only into a constants array."""

    def __init__(self, params):
        self.params = set(params)
        self.consts = []  # list of (symbol, expr)
        self.cache = {}

    def is_param(self, e):
        return e.free_symbols <= self.params

    def new_const(self, e):
        e = sp.nsimplify(e) if False else e
        if e in self.cache:
            return self.cache[e]
        s = sp.Symbol("k[%d]" % len(self.consts))
        self.consts.append((s, e))
        self.cache[e] = s
        return s

    def hoist_poly(self, e, gens):
        """Polynomial in `gens` with (z, dl)-only coefficients: one constant per monomial."""
        p = sp.Poly(sp.expand(e), *gens)
        terms = []
        for monom, coeff in p.terms():
            coeff = sp.factor(sp.simplify(coeff))
            c = coeff if coeff.is_Number else self.new_const(coeff)
            t = c
            for g, mm in zip(gens, monom):
                t = t * g ** mm
            terms.append(t)
        if not terms:
            return sp.Integer(0)
        return sp.horner(sp.Add(*terms), *gens)

    def hoist(self, e):
        if e.is_Number:
            return e
        if e.is_Symbol:
            if e in self.params:
                return self.new_const(e)
            return e
        if self.is_param(e):
            return self.new_const(e)
        if e.is_Mul or e.is_Add:
            pa = [a for a in e.args if self.is_param(a)]
            ot = [a for a in e.args if not self.is_param(a)]
            new_ot = [self.hoist(a) for a in ot]
            if pa:
                comb = e.func(*pa)
                if comb.is_Number:
                    return e.func(comb, *new_ot)
                return e.func(self.new_const(comb), *new_ot)
            return e.func(*new_ot)
        return e.func(*[self.hoist(a) for a in e.args])


def factor_state_exps(lines, outputs, nx):
    """exp(sum_i c_i x[i]) with half-integer c_i (the log-transformed SIR state: every exponential of the step, its
    Jacobian and its Hessian contraction is of this form) as a product of powers of e_i = exp(x[i] / 2) and
    ie_i = 1 / e_i: one exponential per state component instead of one per distinct argument (ten in the SIR step;
    a transcendental costs ~40 dependent fp64 instructions, which is what paces the sequential scans)."""
    xs = [sp.Symbol("x[%d]" % i) for i in range(nx)]
    atoms = set()
    for e in outputs:
        atoms |= e.atoms(sp.exp)
    E = [sp.Symbol("e%d" % i, positive=True) for i in range(nx)]
    IE = [sp.Symbol("ie%d" % i, positive=True) for i in range(nx)]
    rep, need_e, need_ie = {}, set(), set()
    for a in atoms:
        arg = sp.expand(a.args[0])
        if not arg.free_symbols or not arg.free_symbols <= set(xs):
            continue
        pol = sp.Poly(arg, *xs)
        if pol.total_degree() != 1 or pol.coeff_monomial(1) != 0:
            continue
        co = [pol.coeff_monomial(x) for x in xs]
        if not all((2 * c).is_Integer for c in co):
            continue
        term = sp.Integer(1)
        for i, c in enumerate(co):
            n = int(2 * c)
            if n > 0:
                term *= E[i] ** n
                need_e.add(i)
            elif n < 0:
                term *= IE[i] ** (-n)
                need_e.add(i), need_ie.add(i)
        rep[a] = term
    if len(rep) < 3:
        return outputs
    for i in sorted(need_e):
        lines.append("  const double e%d = exp(0.5*x[%d]);" % (i, i))
    for i in sorted(need_ie):
        lines.append("  const double ie%d = 1.0/e%d;" % (i, i))
    return [e.xreplace(rep) for e in outputs]


def emit_assignments(lines, outputs, names, tmp_prefix="t", nx=0):
    """CSE `outputs` and append C statements to `lines`."""
    if nx:
        outputs = factor_state_exps(lines, list(outputs), nx)
    repl, red = sp.cse(list(outputs), symbols=sp.numbered_symbols(tmp_prefix), optimizations="basic")
    for s, e in repl:
        lines.append("  const double %s = %s;" % (s, cc(e)))
    for n, e in zip(names, red):
        lines.append("  %s = %s;" % (n, cc(e)))


# ----------------------------------------------------------------------------- SDE schemes
def jvp(f, x, a):
    return sp.Matrix(f).jacobian(sp.Matrix(x)) * sp.Matrix(a)


def mhp(f, x, M):
    """matrix_hessian_product: out_i = sum_kl d2 f_i/dx_k dx_l M_kl"""
    out = []
    for fi in f:
        H = sp.hessian(fi, list(x))
        out.append(sum(H[k, l] * M[k, l] for k in range(len(x)) for l in range(len(x))))
    return sp.Matrix(out)


def strong_order_1p5_additive(drift, diff, x, z, v, dl):
    """sde/integrators.py:43-63 with diffusion_operator :95-127 and Lj_operator :130-149."""
    a = sp.Matrix(drift(x, z))
    B = sp.Matrix(diff(x, z))
    dim_noise = len(v) // 2
    dw = [sp.sqrt(dl) * v[j] for j in range(dim_noise)]
    dzeta = [dl * sp.sqrt(dl) * (v[j] + v[dim_noise + j] / sp.sqrt(3)) / 2 for j in range(dim_noise)]
    L0a = jvp(a, x, a) + mhp(a, x, B * B.T) / 2
    xn = sp.Matrix(x) + dl * a + B * sp.Matrix(dw) + (dl ** 2 / 2) * L0a
    for j in range(dim_noise):
        xn = xn + jvp(a, x, B[:, j]) * dzeta[j]
    return [sp.simplify(e) for e in xn]


def euler_maruyama(a, B, x, v, dl):
    """sde/integrators.py:8-14."""
    xn = sp.Matrix(x) + dl * sp.Matrix(a) + sp.sqrt(dl) * sp.Matrix(B) * sp.Matrix(v)
    return list(xn)


def transform_sde(fwd, bwd, drift, diff, y, z):
    """sde/transforms.py:9-63 (Ito's lemma)."""
    xs = sp.symbols("xx0:%d" % len(y), real=True)
    a = sp.Matrix(drift(xs, z))
    B = sp.Matrix(diff(xs, z))
    f = sp.Matrix(fwd(xs))
    x_y = bwd(y)
    sub = list(zip(xs, x_y))
    a_y = (jvp(f, xs, a) + mhp(f, xs, B * B.T) / 2).subs(sub)
    B_y = (f.jacobian(sp.Matrix(xs)) * B).subs(sub)
    a_y = sp.Matrix([sp.simplify(e) for e in a_y])
    B_y = B_y.applyfunc(sp.simplify)
    return a_y, B_y


# ----------------------------------------------------------------------------- models
def model_fhn():
    X, V, Z, U, V0 = 2, 2, 4, 4, 2
    x = sp.symbols("x0:2", real=True)
    v = sp.symbols("v0:2", real=True)
    z = sp.symbols("z0:4", real=True)  # sigma, eps, gamma, beta
    u = sp.symbols("u0:4", real=True)
    v0 = sp.symbols("w0:2", real=True)
    dl = sp.Symbol("dl", positive=True)

    def drift(x, z):
        s, e, g, b = z
        return [(x[0] - x[0] ** 3 - x[1]) / e, g * x[0] - x[1] + b]

    def diff(x, z):
        s, e, g, b = z
        return [[0], [s]]

    f = strong_order_1p5_additive(drift, diff, x, z, v, dl)
    gz = [sp.exp(u[0]), sp.exp(u[1]), sp.exp(u[2]), u[3]]  # fhn.py:41-43
    gx0 = [v0[0], v0[1] - z[3]]  # fhn.py:50-51
    obs = x[0]  # fhn.py:37-38
    return dict(name="fhn", X=X, V=V, Z=Z, U=U, V0=V0, x=x, v=v, z=z, u=u, v0=v0, dl=dl, f=f, gz=gz,
                gx0=gx0, obs=obs)


def model_fhn_nb():
    """FitzHugh-Nagumo with the priors of the reference's notebook (FitzHugh-Nagumo_example.ipynb, cells 7-18): the
    same drift / diffusion / integrator as model_fhn, sigma = exp(u0/2 - 1), eps = exp(u1/2 - 2), gamma = u2/2 + 1,
    beta = u3/2 + 1, x_0 = (-1/2, -1/2) + v_0.  Used to reproduce the notebook's posterior table (its only
    known-answer material)."""
    m = model_fhn()
    u, v0 = m["u"], m["v0"]
    h = sp.Rational(1, 2)
    m["name"] = "fhnnb"
    m["gz"] = [sp.exp(h * u[0] - 1), sp.exp(h * u[1] - 2), h * u[2] + 1, h * u[3] + 1]
    m["gx0"] = [v0[0] - h, v0[1] - h]
    return m


def model_sir():
    X, V, Z, U, V0 = 3, 3, 4, 4, 1
    y = sp.symbols("x0:3", real=True)
    v = sp.symbols("v0:3", real=True)
    z = sp.symbols("z0:4", positive=True)  # beta, gamma, zeta, eps
    u = sp.symbols("u0:4", real=True)
    v0 = sp.symbols("w0:1", real=True)
    dl = sp.Symbol("dl", positive=True)
    N = 763

    def drift(x, z):
        al = sp.exp(x[2])
        b, g, ze, ep = z
        return [-al * x[0] * x[1] / N, al * x[0] * x[1] / N - b * x[1], g * (ze - x[2])]

    def diff(x, z):
        al = sp.exp(x[2])
        b, g, ze, ep = z
        return [[sp.sqrt(al * x[0] * x[1] / N), 0, 0],
                [-sp.sqrt(al * x[0] * x[1] / N), sp.sqrt(b * x[1]), 0],
                [0, 0, ep]]

    # the transform is applied for positive S, I
    def drift_p(x, z):
        return drift(x, z)

    xs_pos = None
    a_y, B_y = transform_sde(lambda x: [sp.log(x[0]), sp.log(x[1]), x[2]],
                             lambda yy: [sp.exp(yy[0]), sp.exp(yy[1]), yy[2]],
                             drift, diff, y, z)
    # square roots of exp(...) products: write with half-exponents (valid for real y)
    B_y = B_y.applyfunc(lambda e: sp.powdenest(sp.powsimp(sp.expand_power_base(e, force=True), force=True), force=True))
    f = euler_maruyama(a_y, B_y, y, v, dl)
    # zeta (z[2]) is real-valued in the model; positivity assumption above is only used
    # to simplify sqrt(beta * ...) and does not enter any emitted formula for zeta.
    gz = [sp.exp(u[0]), sp.exp(u[1]), u[2], sp.exp(sp.sqrt(sp.Rational(3, 4)) * u[3] + u[1] / 2 - 3)]  # sir.py:77-85
    gx0 = [sp.log(762), sp.Integer(0), v0[0]]  # sir.py:88-89
    obs = sp.exp(y[1])  # sir.py:73-74
    return dict(name="sir", X=X, V=V, Z=Z, U=U, V0=V0, x=y, v=v, z=z, u=u, v0=v0, dl=dl, f=f, gz=gz,
                gx0=gx0, obs=obs, a_y=a_y, B_y=B_y)


# ----------------------------------------------------------------------------- emission
def gen_model(m):
    nm = m["name"]
    X, V, Z, U, V0 = m["X"], m["V"], m["Z"], m["U"], m["V0"]
    x, v, z, u, v0, dl = m["x"], m["v"], m["z"], m["u"], m["v0"], m["dl"]
    f = sp.Matrix(m["f"])
    xi = list(x) + list(v) + list(z)
    NXI = X + V + Z
    A = f.jacobian(sp.Matrix(x))
    Bm = f.jacobian(sp.Matrix(v))
    Zf = f.jacobian(sp.Matrix(z))
    S = sp.Matrix(X, NXI, lambda a, mm: sp.Symbol("S[%d]" % (a * NXI + mm)))
    full_jac = f.jacobian(sp.Matrix(xi))  # X x NXI
    psi = sum(S[a, mm] * full_jac[a, mm] for a in range(X) for mm in range(NXI))
    hess_out = [sp.diff(psi, xi_k) for xi_k in xi]

    h = Hoister(list(z) + [dl])
    gens = list(x) + list(v)
    poly = all(sp.expand(e).is_polynomial(*gens) for e in f)

    def hoist(e, extra=()):
        if poly:
            return h.hoist_poly(e, list(extra) + gens)
        return h.hoist(e)

    f_h = [hoist(e) for e in f]
    A_h = [hoist(e) for e in A]
    B_h = [hoist(e) for e in Bm]
    Z_h = [hoist(e) for e in Zf]
    H_h = [hoist(e, list(S)) for e in hess_out]
    NK = len(h.consts)

    def subs_arr(e):
        # replace x0.. v0.. by array refs
        rep = {}
        for i, s in enumerate(x):
            rep[s] = sp.Symbol("x[%d]" % i)
        for i, s in enumerate(v):
            rep[s] = sp.Symbol("v[%d]" % i)
        for i, s in enumerate(z):
            rep[s] = sp.Symbol("z[%d]" % i)
        for i, s in enumerate(u):
            rep[s] = sp.Symbol("u[%d]" % i)
        for i, s in enumerate(v0):
            rep[s] = sp.Symbol("v0[%d]" % i)
        return e.xreplace(rep)

    L = []
    up = nm.upper()
    L.append("/* ---- model: %s ---- */" % nm)
    L.append("#define CHMC_%s_X %d" % (up, X))
    L.append("#define CHMC_%s_V %d" % (up, V))
    L.append("#define CHMC_%s_Z %d" % (up, Z))
    L.append("#define CHMC_%s_V0 %d" % (up, V0))
    L.append("#define CHMC_%s_NK %d" % (up, NK))
    L.append("")
    # precompute
    L.append("CHMC_HD static inline void chmc_%s_precompute(const double* z, double dl, double* k) {" % nm)
    emit_assignments(L, [subs_arr(e) for _, e in h.consts], ["k[%d]" % i for i in range(NK)], "p")
    L.append("}")
    L.append("")
    # step
    L.append("CHMC_HD static inline void chmc_%s_step(const double* k, const double* x, const double* v, double* xn) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in f_h], ["xn[%d]" % i for i in range(X)], nx=X)
    L.append("}")
    L.append("")
    # step + jac
    L.append("CHMC_HD static inline void chmc_%s_step_jac(const double* k, const double* x, const double* v, double* xn, double* A, double* B, double* Zf) {" % nm)
    outs = [subs_arr(e) for e in f_h + A_h + B_h + Z_h]
    names = (["xn[%d]" % i for i in range(X)] + ["A[%d]" % i for i in range(X * X)] +
             ["B[%d]" % i for i in range(X * V)] + ["Zf[%d]" % i for i in range(X * Z)])
    emit_assignments(L, outs, names, nx=X)
    L.append("}")
    L.append("")
    # jac only (no xn)
    L.append("CHMC_HD static inline void chmc_%s_jac(const double* k, const double* x, const double* v, double* A, double* B, double* Zf) {" % nm)
    outs = [subs_arr(e) for e in A_h + B_h + Z_h]
    names = (["A[%d]" % i for i in range(X * X)] + ["B[%d]" % i for i in range(X * V)] +
             ["Zf[%d]" % i for i in range(X * Z)])
    emit_assignments(L, outs, names, nx=X)
    L.append("}")
    L.append("")
    # A and B only
    L.append("CHMC_HD static inline void chmc_%s_jac_ab(const double* k, const double* x, const double* v, double* A, double* B) {" % nm)
    outs = [subs_arr(e) for e in A_h + B_h]
    names = ["A[%d]" % i for i in range(X * X)] + ["B[%d]" % i for i in range(X * V)]
    emit_assignments(L, outs, names, nx=X)
    L.append("}")
    L.append("")
    # hess
    L.append("CHMC_HD static inline void chmc_%s_step_hess(const double* k, const double* x, const double* v, const double* S, double* out) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in H_h], ["out[%d]" % i for i in range(NXI)], nx=X)
    L.append("}")
    L.append("")
    # gz
    gz = sp.Matrix(m["gz"])
    L.append("CHMC_HD static inline void chmc_%s_gz(const double* u, double* z) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in gz], ["z[%d]" % i for i in range(Z)])
    L.append("}")
    G = gz.jacobian(sp.Matrix(u[:Z]))
    L.append("CHMC_HD static inline void chmc_%s_gz_jac(const double* u, double* G) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in G], ["G[%d]" % i for i in range(Z * Z)])
    L.append("}")
    ud = [sp.Symbol("ud[%d]" % i) for i in range(Z)]
    zb = [sp.Symbol("zb[%d]" % i) for i in range(Z)]
    phi = sum(zb[a] * G[a, b] * ud[b] for a in range(Z) for b in range(Z))
    L.append("/* out_k = d/du_k [ zb^T gz'(u) ud ] */")
    L.append("CHMC_HD static inline void chmc_%s_gz_hess(const double* u, const double* ud, const double* zb, double* out) {" % nm)
    emit_assignments(L, [subs_arr(sp.diff(phi, uk)) for uk in u[:Z]], ["out[%d]" % i for i in range(Z)])
    L.append("}")
    # gx0
    gx0 = sp.Matrix(m["gx0"])
    L.append("CHMC_HD static inline void chmc_%s_gx0(const double* z, const double* v0, double* x0) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in gx0], ["x0[%d]" % i for i in range(X)])
    L.append("}")
    Jz = gx0.jacobian(sp.Matrix(z))
    Jv = gx0.jacobian(sp.Matrix(v0))
    for e in list(Jz) + list(Jv):
        assert e.is_Number, "generate_x_0 must be affine in (z, v_0)"
    L.append("CHMC_HD static inline void chmc_%s_gx0_jac(double* dz, double* dv0) {" % nm)
    for i, e in enumerate(Jz):
        L.append("  dz[%d] = %s;" % (i, cc(sp.Float(e) if e != 0 else sp.Float(0))))
    for i, e in enumerate(Jv):
        L.append("  dv0[%d] = %s;" % (i, cc(sp.Float(e) if e != 0 else sp.Float(0))))
    L.append("}")
    # obs
    obs = m["obs"]
    L.append("CHMC_HD static inline double chmc_%s_obs(const double* x) {" % nm)
    L.append("  return %s;" % cc(subs_arr(obs)))
    L.append("}")
    g = [sp.diff(obs, xk) for xk in x]
    L.append("CHMC_HD static inline void chmc_%s_obs_grad(const double* x, double* g) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in g], ["g[%d]" % i for i in range(X)])
    L.append("}")
    xd = [sp.Symbol("xd[%d]" % i) for i in range(X)]
    hv = [sum(sp.diff(obs, x[k], x[l]) * xd[l] for l in range(X)) for k in range(X)]
    L.append("CHMC_HD static inline void chmc_%s_obs_hess_vec(const double* x, const double* xd, double* out) {" % nm)
    emit_assignments(L, [subs_arr(e) for e in hv], ["out[%d]" % i for i in range(X)])
    L.append("}")
    L.append("")
    return "\n".join(L)


HEADER = """/* GENERATED by tools/gen_models.py -- do not edit.
 *
 * One-step maps of the time-discretised diffusions and their first / second
 * derivatives, derived symbolically from the drift and diffusion coefficients
 * (reference: sde/example_models/fhn.py:17-51, sir.py:19-93 via
 * sde/integrators.py:8-14,43-63,95-149 and sde/transforms.py:9-63).
 */
#ifndef CHMC_MODELS_GEN_H
#define CHMC_MODELS_GEN_H
#include <math.h>
#ifndef CHMC_HD
#define CHMC_HD
#endif

"""


def main():
    parts = [HEADER]
    for mk in (model_fhn, model_fhn_nb, model_sir):
        m = mk()
        print("model", m["name"], file=sys.stderr)
        for i, e in enumerate(m["f"]):
            print("  f[%d] =" % i, e, file=sys.stderr)
        parts.append(gen_model(m))
    parts.append("#endif\n")
    text = "\n".join(parts)
    for out in (os.path.join(ROOT, "manifold_mcmc_for_diffusions_amd", "csrc", "models_gen.h"),
                os.path.join(ROOT, "oracle", "c", "models_gen.h")):
        with open(out, "w") as fh:
            fh.write(text)
        print("wrote", out, file=sys.stderr)


if __name__ == "__main__":
    main()

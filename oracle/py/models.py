"""ORACLE (test infrastructure only -- never imported by the product path).

Closed-form one-step maps of the two example diffusions, written with torch fp64
ops so that `torch.func` can differentiate them the way JAX differentiates the
SymNum-generated `forward_func` in the reference.  PARITY UNPINNED: the reference
ships no tests / golden vectors for this path.

Follows (behaviour, not text):
  sde/example_models/fhn.py:10-65   (FitzHugh-Nagumo, strong-order-1.5 step)
  sde/example_models/sir.py:9-93    (SIR, log transform + Euler-Maruyama + clipping)
  sde/integrators.py:8-14, 43-63, 95-149 ; sde/transforms.py:9-63
The closed forms are those of SURVEY.md Appendix B; tests/test_models.py checks
them against the independent sympy derivation in tools/gen_models.py.
"""
import math
import torch

DT = torch.float64


class FhnModel:
    """sde/example_models/fhn.py"""

    name = "fhn"
    model_id = 0
    dim_x = 2
    dim_w = 1
    dim_z = 4
    dim_v_0 = 2
    dim_v = 2
    dim_y = 1

    @staticmethod
    def forward_func(z, x, v, dl):
        # fhn.py:17-34 through integrators.py:43-63 (additive-noise strong order 1.5)
        sig, eps, gam, bet = z[0], z[1], z[2], z[3]
        x0, x1 = x[0], x[1]
        a0 = (x0 - x0 ** 3 - x1) / eps
        a1 = gam * x0 - x1 + bet
        dzeta = dl ** 1.5 * (v[0] + v[1] / math.sqrt(3.0)) / 2
        xn0 = x0 + dl * a0 + (dl ** 2 / 2) * ((1 - 3 * x0 ** 2) * a0 - a1) / eps - (sig / eps) * dzeta
        xn1 = x1 + dl * a1 + sig * math.sqrt(dl) * v[0] + (dl ** 2 / 2) * (gam * a0 - a1) - sig * dzeta
        return torch.stack([xn0, xn1])

    @staticmethod
    def obs_func(x_seq):  # fhn.py:37-38
        return x_seq[..., 0:1]

    @staticmethod
    def generate_z(u):  # fhn.py:41-43
        return torch.stack([torch.exp(u[0]), torch.exp(u[1]), torch.exp(u[2]), u[3]])

    @staticmethod
    def generate_sigma_y(u):  # fhn.py:46-47
        return torch.exp(u[4])

    @staticmethod
    def generate_x_0(z, v_0):  # fhn.py:50-51
        return v_0 - torch.stack([torch.zeros((), dtype=DT), z[3]])


class FhnNbModel(FhnModel):
    """The FitzHugh-Nagumo model as the reference's notebook sets it up (FitzHugh-Nagumo_example.ipynb cells 7-18):
    drift, diffusion coefficient and strong-order-1.5 step of fhn.py, with the notebook's own priors."""

    name = "fhn_nb"
    model_id = 2

    @staticmethod
    def generate_z(u):  # notebook cell 18
        return torch.stack([torch.exp(0.5 * u[0] - 1), torch.exp(0.5 * u[1] - 2), 0.5 * u[2] + 1, 0.5 * u[3] + 1])

    @staticmethod
    def generate_x_0(z, v_0):  # notebook cell 18
        return v_0 - 0.5


class SirModel:
    """sde/example_models/sir.py"""

    name = "sir"
    model_id = 1
    dim_x = 3
    dim_y = 1
    dim_w = 3
    dim_z = 4
    dim_v_0 = 1
    dim_v = 3
    N = 763.0

    @staticmethod
    def _forward_func(z, y, v, dl):
        # sir.py:19-51: Ito transform to (log S, log I, c) then Euler-Maruyama
        N = SirModel.N
        bet, gam, zet, eps = z[0], z[1], z[2], z[3]
        y0, y1, y2 = y[0], y[1], y[2]
        al = torch.exp(y2)
        a0 = -(al / N) * (torch.exp(y1) + 0.5 * torch.exp(y1 - y0))
        a1 = (al / N) * (torch.exp(y0) - 0.5 * torch.exp(y0 - y1)) - bet - 0.5 * bet * torch.exp(-y1)
        a2 = gam * (zet - y2)
        b00 = torch.exp((-y0 + y1 + y2) / 2) / math.sqrt(N)
        b10 = -torch.exp((y0 - y1 + y2) / 2) / math.sqrt(N)
        b11 = torch.sqrt(bet) * torch.exp(-y1 / 2)
        sq = math.sqrt(dl)
        return torch.stack([
            y0 + dl * a0 + sq * b00 * v[0],
            y1 + dl * a1 + sq * (b10 * v[0] + b11 * v[1]),
            y2 + dl * a2 + sq * eps * v[2],
        ])

    @staticmethod
    def forward_func(z, x, v, dl):
        # sir.py:54-70: clip first two components below at -500 before the step and keep
        # the old (clipped) value of a component that was at / below the cutoff
        xc = torch.cat([torch.clamp(x[:2], min=-500.0), x[2:]])
        x_ = SirModel._forward_func(z, xc, v, dl)
        return torch.stack([
            torch.where(xc[0] > -500, x_[0], xc[0]),
            torch.where(xc[1] > -500, x_[1], xc[1]),
            x_[2],
        ])

    @staticmethod
    def obs_func(x_seq):  # sir.py:73-74
        return torch.exp(x_seq[..., 1:2])

    @staticmethod
    def generate_z(u):  # sir.py:77-85
        return torch.stack([
            torch.exp(u[0]),
            torch.exp(u[1]),
            u[2],
            torch.exp(math.sqrt(0.75) * u[3] + 0.5 * u[1] - 3),
        ])

    @staticmethod
    def generate_x_0(z, v_0):  # sir.py:88-89
        return torch.stack([
            torch.tensor(math.log(762.0), dtype=DT),
            torch.tensor(0.0, dtype=DT),
            v_0[0],
        ])

    @staticmethod
    def generate_sigma_y(u):  # sir.py:92-93
        return torch.exp(u[4])


fhn = FhnModel
fhn_nb = FhnNbModel
sir = SirModel
MODELS = {"fhn": fhn, "sir": sir, "fhn_nb": fhn_nb}

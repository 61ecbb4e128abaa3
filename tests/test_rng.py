"""Device normal generator (Philox4x32-10 + Box-Muller, csrc/chmc_core.h KNormalFill) against a NumPy restatement,
and the statistical sanity of the stream.  CPU (emulation build) + GPU."""
import numpy as np
import pytest
from test_emu_logic import emu_lib  # noqa: F401
from helpers import make_case, make_ctx

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
MASK = np.uint64(0xFFFFFFFF)


def philox(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    return c0, c1, c2, c3


def reference_normals(Q, chain, seed, draw):
    npair = (Q + 1) // 2
    j = np.arange(npair, dtype=np.uint64)
    r0, r1, r2, r3 = philox(j, np.full(npair, draw & 0xFFFFFFFF), np.full(npair, chain), np.full(npair, draw >> 32),
                            seed & 0xFFFFFFFF, seed >> 32)
    u1 = ((r0 << np.uint64(21)) ^ (r1 >> np.uint64(11))).astype(np.float64) / 9007199254740992.0 + 0.5 / 9007199254740992.0
    u2 = ((r2 << np.uint64(21)) ^ (r3 >> np.uint64(11))).astype(np.float64) / 9007199254740992.0
    rad, ang = np.sqrt(-2.0 * np.log(u1)), 2.0 * np.pi * u2
    out = np.empty(2 * npair)
    out[0::2], out[1::2] = rad * np.cos(ang), rad * np.sin(ang)
    return out[:Q]


def check(ctx, case):
    seed, draw, off = 20200710, 3, 5
    ctx.set_state(case["q"], None, case["x_obs"], 0)
    ctx.sample_momentum(seed, draw, off)
    _, p, _, _ = ctx.get_state()
    for c in range(ctx.B):
        n = reference_normals(ctx.Q, c + off, seed, draw)
        expect = n - case["osys"].jacob_products(case["q"][c], case["x_obs"][c], 0, n, np.zeros(ctx.dim_c))[3]
        np.testing.assert_allclose(p[c], expect, rtol=1e-9, atol=1e-10)  # projected reference normals
    assert np.abs(ctx.lmult_by_jacob_constr(p)).max() < 1e-9


def test_momentum_generator_cpu(emu_lib):  # noqa: F811
    case = make_case("fhn", 6, 4, 2, True, B=3, seed=31)
    ctx = make_ctx(case)
    check(ctx, case)
    ctx.close()


def test_reference_stream_statistics():
    n = np.concatenate([reference_normals(20001, c, 7, d) for c in range(4) for d in range(3)])
    assert abs(n.mean()) < 0.01 and abs(n.var() - 1.0) < 0.02
    assert abs(((n ** 4).mean()) - 3.0) < 0.1
    assert abs(np.corrcoef(n[:-1], n[1:])[0, 1]) < 0.01
    assert not np.allclose(reference_normals(100, 0, 7, 1), reference_normals(100, 1, 7, 1))
    assert not np.allclose(reference_normals(100, 0, 7, 1), reference_normals(100, 0, 7, 2))


@pytest.mark.gpu
def test_momentum_generator_gpu():
    case = make_case("fhn", 12, 10, 5, True, B=3, seed=32)
    ctx = make_ctx(case)
    assert ctx.L.chmc_backend() == b"hip:gfx950"
    check(ctx, case)
    ctx.close()

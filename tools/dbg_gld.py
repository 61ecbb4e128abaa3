import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from helpers import make_case, make_ctx
for (m, T, S, R) in (("sir", 9, 6, None), ("sir", 8, 6, None), ("sir", 3, 6, None), ("sir", 6, 8, 2)):
    case = make_case(m, T, S, R, True, B=2, seed=11)
    ctx = make_ctx(case)
    ctx.set_state(case["q"], None, case["x_obs"], 0)
    g = ctx.grad_log_det_sqrt_gram()
    for c in range(2):
        _, _, ld, go = case["osys"].gram_ops(case["q"][c], case["x_obs"][c], 0)
        e = np.abs(g[c] - go)
        print(m, T, S, R, "RM", ctx.RM, "chain", c, "err u %.2e v0 %.2e v %.2e n %.2e | max grad %.2e" % (e[:4].max(), e[4:5].max(), e[5:5 + T * S * 3].max(), e[5 + T * S * 3:].max(), np.abs(go).max()), "argmax", int(np.argmax(e)))
    ctx.close()

for G in 1 0; do CHMC_XOBS_PAR=$G python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload
wl = FhnWorkload(256, num_steps_per_obs=400, device_init=True)
ctx = wl.ctx
for it in range(3):
    ctx.sample_momentum(wl.seed, 1000 + it)
    for _ in range(8):
        ctx.leapfrog_step(np.full(256, 0.1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.switch_partition()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    q, p, xo, part = ctx.get_state()
    print('XOBS_PAR', os.environ.get('CHMC_XOBS_PAR'), 'switch %.2f ms' % ((t1 - t0) * 1e3), 'xobs checksum %.15e' % np.abs(xo).sum(), 'finite', np.isfinite(xo).all())
PY
done

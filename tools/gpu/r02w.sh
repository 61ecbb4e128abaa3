mkdir -p gpurun_out/r02w
python - > gpurun_out/r02w/diag.log 2>&1 <<'PY'
import sys, os, numpy as np
sys.path.insert(0, 'examples')
import fhn_notebook_posterior as nb
rows, res, n_moving = nb.run(64, 450, 150, 0, out_dir='gpurun_out/r02w/run', verbose=True, transition='dynamic')
tr = {k: np.load(f) for k, f in res['trace_files'].items()}
s = tr['σ'][:, 150:]
mv = (np.diff(s, axis=1) != 0).mean(1)
oc = res.get('chain_outcomes')
print('n_step per chain available:', res['n_step'].shape)
for c in np.argsort(-np.abs(s.mean(1) - 0.282))[:12]:
    print(c, 'move frac %.2f' % mv[c], 'sigma mean %.3f sd %.3f' % (s[c].mean(), s[c].std()), 'eps %.4f' % tr['ϵ'][c, 150:].mean(),
          'first/last', s[c, 0].round(3), s[c, -1].round(3), oc[c] if oc is not None else '')
PY
tail -40 gpurun_out/r02w/diag.log

"""Per-parameter error of one full update (rollout + epochs x minibatches of clip + Adam) against the oracle, RWARE tiny-4ag shapes:
python scripts/debug/rware_update_err.py E n_head n_block"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_rware_gpu as t
E, nh, nb = (int(x) for x in sys.argv[1:4])
ol, dl = t._mk((8, 1, 3, 4, 1, 4, 11), 8, 16, E=E, nh=nh, nb=nb)
ol.rollout(); dl.rollout()
ol.update(); dl.update()
rows = []
for net, ref in ((dl.guider, ol.gp), (dl.actor, ol.ap)):
    for n, v in net.named.items():
        a, b = v.detach().cpu().double().reshape(-1), ref[n].reshape(v.shape).double().reshape(-1)
        d = (a - b).abs()
        rows.append((d.max().item(), n, int((d > 3e-5).sum()), d.numel(), b.abs().max().item()))
rows.sort(reverse=True)
print(f"lr {dl.g_opt.learning_rate() if hasattr(dl, 'g_opt') else '?'}")
for e, n, cnt, tot, sc in rows[:12]:
    print(f"{n:40s} max err {e:.3e}  elements over 3e-5: {cnt}/{tot}  (scale {sc:.2e})")

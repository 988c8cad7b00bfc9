"""Ad-hoc GPU-vs-oracle report (development aid; the asserted versions live in tests/)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scarlet_amd import BlendBatch, ComponentSpec, synthetic
from oracle import pgm

def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-300)

s = synthetic.make_blend(1234)
K = len(s["morphs"])
comps = [ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"]) for k in range(K)]
batch = BlendBatch(s["data"][None], s["weights"][None], [comps], kernel=s["diff_kernel"], max_iter=16)
sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
               [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k], sed_min_step=s["noise_rms"]) for k in range(K)])
model, rendered, logL = batch.forward()
om = sc.get_model(); orr = sc.render(om)
print("model", rel(model[0], om), "rendered", rel(rendered[0], orr), "logL", logL[0], sc.log_likelihood(orr))
gs, gm = batch.gradient()
_, grads = sc.loss_and_gradients(); sc.loss.clear()
print("g_sed", max(rel(gs[k], grads[k][0]) for k in range(K)), "g_morph", max(rel(gm[k], grads[k][1]) for k in range(K)))
E = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for it in range(4):
    batch.step(it, 1, e_rel=0.0) if E == 0 else batch.step(it, 1, e_rel=E, min_iter=1000)
    sc.step(it, E)
    sed, morphs = batch.parameters()
    mom = batch.moments()
    print("it", it, "loss", batch.loss_history()[0][-1], sc.loss[-1],
          "sed", max(rel(sed[k], sc.components[k].sed) for k in range(K)),
          "morph", max(np.abs(morphs[k] - sc.components[k].morph).max() for k in range(K)),
          "m_sed", max(rel(mom["m_sed"][k], sc.components[k].m_sed) for k in range(K)),
          "vh_morph", max(rel(mom["vhat_morph"][k], sc.components[k].vhat_morph) for k in range(K)))
    if it == 0:
        for k in range(K):
            print("  k", k, sed[k], sc.components[k].sed, np.abs(morphs[k] - sc.components[k].morph).max())

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import golden, hsc_scene
from test_gpu_facade import build_blend, components_of
hsc = golden("hsc_cosmos_35")
blend, obs = build_blend(hsc, True)
n, logL = blend.fit(30, e_rel=1e-5)
sc = hsc_scene(hsc)
sc.fit(30, e_rel=1e-5, resizing=True)
chi = np.array(blend.loss) - sc.log_norm
chi_ref = np.array(sc.loss) - sc.log_norm
for i, (a, b) in enumerate(zip(chi, chi_ref)):
    print(i, a, b, (a - b) / b)
for comp, c in zip(components_of(blend), sc.components):
    im = comp.children[1].parameters[0]
    print(im.shape, c.morph.shape, im.step, c.morph_step, np.abs(np.asarray(im) - c.morph).max(),
          np.abs(np.asarray(comp.children[0].parameters[0]) - c.sed).max())

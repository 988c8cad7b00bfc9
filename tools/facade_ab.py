"""Development aid: fit_blends with the resident batch against the per-round rebuilt batches
and against individual Blend.fit calls, blend by blend (iteration counts, final loss, boxes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import scarlet_amd as scarlet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def boxes(b):
    return [tuple(src.children[1].bbox.shape) for src in b.sources]


a = bench.build_facade_blends(0, n, 0)
ra = scarlet.fit_blends(a, K, e_rel=1e-4)
os.environ["SCARLET_AMD_FIT_BLENDS"] = "rebuild"
b = bench.build_facade_blends(0, n, 0)
rb = scarlet.fit_blends(b, K, e_rel=1e-4)
diff = [i for i in range(n) if ra[i] != rb[i] or boxes(a[i]) != boxes(b[i])]
print("resident vs rebuilt: %d of %d blends differ" % (len(diff), n), diff[:20])
for i in diff[:4]:
    one = bench.build_facade_blends(i, i + 1, 0)[0]
    r1 = one.fit(K, e_rel=1e-4)
    print(i, "resident", ra[i], "rebuilt", rb[i], "Blend.fit", r1)
    print("   boxes resident", boxes(a[i]), "\n   boxes rebuilt ", boxes(b[i]), "\n   boxes fit     ", boxes(one))
    la, lb, l1 = np.array(a[i].loss), np.array(b[i].loss), np.array(one.loss)
    m = min(len(la), len(lb), len(l1))
    print("   first loss index where resident != fit:", next((j for j in range(m) if la[j] != l1[j]), None),
          " rebuilt != fit:", next((j for j in range(m) if lb[j] != l1[j]), None))

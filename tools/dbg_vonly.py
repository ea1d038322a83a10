import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_product
from oracle import binding as O
P = load_product(); h = P.Handle(0)
rng = np.random.default_rng(11)
H, W = 256, 512
line = rng.integers(0, 256, size=(H, 3)).astype(np.uint8)
img = np.empty((H, W, 4), np.uint8); img[..., 3] = 255; img[..., :3] = line[:, None, :]
for filt in (2, 4):
    got = h.shrink_image(img, 32, 32, 1, filt, 2.0); exp = O.shrink_image(img, 32, 32, 1, filt, 2.0)
    t = 0
    w_, h_ = int(exp[1][t]), int(exp[2][t])
    g = got[3][t, :w_*h_*4].reshape(h_, w_, 4).astype(int); e = exp[3][t, :w_*h_*4].reshape(h_, w_, 4).astype(int)
    d = np.argwhere(g != e)
    print("filter", filt, "tile0 dims", w_, h_, "ndiff", len(d), "rows", sorted(set(d[:, 0].tolist())), "cols", sorted(set(d[:, 1].tolist()))[:8], "ch", sorted(set(d[:, 2].tolist())))
    if len(d): print(" sample got/exp", g[d[0][0], d[0][1]], e[d[0][0], d[0][1]])
    st, sz, k, prec = P.axis_table(32, h_, filt)
    print(" table starts", st.tolist(), "sizes", sz.tolist(), "prec", prec)

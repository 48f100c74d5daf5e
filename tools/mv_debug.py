import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from bench import MovingScene, shift, make_ctx
from tests.helpers import make_case, configure
from mpcholonavigation_amd.optimizer import Smpc
for B,T in ((2000,56),(70000,64)):
    cfg, scn, noise = make_case(B, T)
    g = Smpc(cfg); configure(g, scn, noise=noise)
    mv = MovingScene(scn, cfg.model_dt)
    u = scn.u0
    line=[]
    for k in range(40):
        tk = mv.tick()
        ug, og = g.optimize(tk, u)
        line.append(f"{og.furthest_reached_path_point}{'*' if og.passes>1 else ''}@{mv.x:.3f}/{tk.path_x[0]:.2f}")
        mv.advance(ug); u = shift(ug)
    print(B,T," ".join(line))

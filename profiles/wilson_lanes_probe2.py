import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
sys.argv = ["x"]
import wilson_lanes_probe as P
import pythtb_amd as tb, helpers as hp
for nb in (3, 4):
    mw = hp.random_model(tb.tb_model, 2 * nb, 2, 1, 7 + nb)
    for mesh, d in (([1025, 257], 0), ([257, 1025], 1)):
        ww = tb.wf_array(mw, mesh); ww.solve_on_grid([0.0, 0.0]); occ = list(range(nb))
        for form in (0, 1):
            for swz in (1, 0):
                print(nb, mesh, d, "form", form, "swz", swz, P.run(ww, occ, d, TBK_WILSON_FORM=form, TBK_WILSON_SWZ=swz), flush=True)

"""Remaining documented examples on the HIP backend: how far from the printed numbers? (GPU box)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refdocs_flow as F
d = F.load()
be = F.HipBackend("exact")
r = F.example_cv_heart_then_deviance(be, d)
rel = np.abs(r["deviance"] / d["deviance_wine"] - 1)
print("wine deviance rel err: max %.3e median %.3e" % (rel.max(), np.median(rel)))
for nudge in (0.0, 1e-13):
    r = F.example_predict_chain(be, d, student_lambda0_nudge=nudge)
    print("nudge", nudge, "iris", (r["iris_class"] == d["cv_iris_class"]).sum(), "heart", (r["heart_class"] == d["predict_heart_class"]).sum(),
          "student rows equal", (r["student_nonzero"] == d["predict_student_nonzero"]).all(axis=1).sum(), "npasses", r["student_npasses"])
    print(" print_cv", r["print_cv"][:, 1:4].tolist(), "want", d["print_cv_mtcars"][:, 1:4].tolist())
    print(" mtcars df eq", (r["mtcars_df"] == d["print_mtcars_df"]).all(), "lambda eq", (np.round(r["mtcars_lambda"], 2) == d["print_mtcars_lambda"]).all(),
          "dev eq", (F.signif_round(r["mtcars_dev"], 1) == d["print_mtcars_dev"]).sum())
print("score wine", F.example_score_wine(be, d), "want", float(d["score_wine_deviance"]))

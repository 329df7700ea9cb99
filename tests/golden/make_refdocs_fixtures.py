"""Golden vectors PRINTED BY THE REFERENCE ITSELF -> tests/golden/refdocs.npz (dev container only).

The reference tree ships its pkgdown site (docs/, built with pkgdown 1.1.0 => R 3.5.x, 2018): the
example sections of docs/reference/*.html hold the console output of a real Rcpp/Eigen build of
the package.  This script copies the NUMBERS out of those outputs (data, not page text) together
with the inputs the examples used:

  * data/heart.rda, wine.rda, student.rda (the reference's bundled data sets; abalone.npz and
    iris.npz already exist, see make_fixtures.py),
  * R's built-in `mtcars` columns drat, hp, disp (R's datasets package, Henderson & Velleman
    1981; typed in below and checked here against the lambda path the reference printed for it).

pkgdown 1.1.0 calls set.seed(1014) once and then runs the topics' examples in alphabetical order
in one R session, so the generator state carries from one example into the next; which examples
re-seed and which inherit is recorded in EXAMPLES below and exercised by tests/refdocs_flow.py.

    python tests/golden/make_refdocs_fixtures.py
"""
import html
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import load_rda  # noqa: E402

REF = "/root/reference"

# mtcars[, c("disp", "hp", "drat")], rows in R's order (Mazda RX4 ... Volvo 142E)
MTCARS = np.array([
    [160.0, 110, 3.90], [160.0, 110, 3.90], [108.0, 93, 3.85], [258.0, 110, 3.08], [360.0, 175, 3.15],
    [225.0, 105, 2.76], [360.0, 245, 3.21], [146.7, 62, 3.69], [140.8, 95, 3.92], [167.6, 123, 3.92],
    [167.6, 123, 3.92], [275.8, 180, 3.07], [275.8, 180, 3.07], [275.8, 180, 3.07], [472.0, 205, 2.93],
    [460.0, 215, 3.00], [440.0, 230, 3.23], [78.7, 66, 4.08], [75.7, 52, 4.93], [71.1, 65, 4.22],
    [120.1, 97, 3.70], [318.0, 150, 2.76], [304.0, 150, 3.15], [350.0, 245, 3.73], [400.0, 175, 3.08],
    [79.0, 66, 4.08], [120.3, 91, 4.43], [95.1, 113, 3.77], [351.0, 264, 4.22], [145.0, 175, 3.62],
    [301.0, 335, 3.54], [121.0, 109, 4.11]])


def _text(path):
    t = open(os.path.join(REF, "docs", path)).read()
    return html.unescape(re.sub(r"<[^>]+>", "", t))


def _parts(name):
    top = load_rda(os.path.join(REF, "data", name + ".rda"))[name]
    return dict(zip(top["attr"]["names"]["value"], top["value"]))


def _matrix(o):
    d = [int(v) for v in o["attr"]["dim"]["value"]]
    return np.asarray(o["value"], dtype=np.float64).reshape(d, order="F")


def main():
    out = {}
    # ---------------- inputs ----------------
    heart = _parts("heart")
    a = heart["x"]["attr"]
    out["heart_i"] = np.asarray(a["i"]["value"], dtype=np.int32)
    out["heart_p"] = np.asarray(a["p"]["value"], dtype=np.int32)
    out["heart_x"] = np.asarray(a["x"]["value"], dtype=np.float64)
    out["heart_dim"] = np.asarray(a["Dim"]["value"], dtype=np.int64)
    out["heart_y"] = np.asarray(heart["y"]["value"], dtype=np.int64)              # factor codes, 1-based
    out["heart_levels"] = np.array(heart["y"]["attr"]["levels"]["value"])
    wine = _parts("wine")
    out["wine_x"] = _matrix(wine["x"])
    out["wine_y"] = np.asarray(wine["y"]["value"], dtype=np.int64)
    out["wine_levels"] = np.array(wine["y"]["attr"]["levels"]["value"])
    student = _parts("student")
    out["student_x"] = _matrix(student["x"])
    out["student_y"] = _matrix(student["y"])
    out["mtcars_disp_hp_drat"] = MTCARS

    # ---------------- expected outputs (numbers the reference printed) ----------------
    # coef.sgdnet.html: fit <- sgdnet(matrix(rnorm(100), 50, 2), rnorm(50)); coef(fit)   [seed 1014 state]
    rows = {"(Intercept)": [], "V1": [], "V2": []}
    for line in _text("reference/coef.sgdnet.html").splitlines():
        m = re.match(r"#> (\(Intercept\)|V1|V2)\s+(.*)", line)
        if m:
            rows[m.group(1)] += [0.0 if v == "." else float(v) for v in m.group(2).split()]
    out["coef_gaussian"] = np.array([rows["(Intercept)"], rows["V1"], rows["V2"]])
    assert out["coef_gaussian"].shape == (3, 100)

    # cv_sgdnet.html: set.seed(1); heart; binomial, nfolds = 7, alpha = c(0, 1); link predictions
    t = _text("reference/cv_sgdnet.html")
    out["cv_heart_link"] = np.array([float(v) for v in re.findall(r"#>\s+\[\d+,\]\s+(-?\d+\.\d+)", t)])
    assert out["cv_heart_link"].size == 54

    # deviance.sgdnet.html: deviance(sgdnet(wine$x, wine$y, family = "multinomial"))  [inherits the state]
    t = _text("reference/deviance.sgdnet.html")
    blk = t[t.index("deviance(fit)"):]
    out["deviance_wine"] = np.array([float(v) for v in re.findall(r"(?<![\[\d])(\d+\.\d+)", blk)][:100])
    assert out["deviance_wine"].size == 100 and out["deviance_wine"][0] == 386.629686

    # predict.cv_sgdnet.html: set.seed(1); iris; multinomial, nfolds = 5; classes at lambda_min
    t = _text("reference/predict.cv_sgdnet.html")
    out["cv_iris_class"] = np.array(re.findall(r'#>\s+\[\d+,\]\s+"(\w+)"', t))
    assert out["cv_iris_class"].size == 50

    # predict.sgdnet.html [inherits]: heart classes at s = 1/n; student mgaussian non-zeros per lambda
    t = _text("reference/predict.sgdnet.html")
    out["predict_heart_class"] = np.array(re.findall(r'#>\s+\[\d+,\]\s+"(\w+)"', t))
    assert out["predict_heart_class"].size == 54
    nz = np.zeros((100, 21), dtype=bool)
    blocks = re.findall(r"#> \$s(\d+)\n((?:#>\s+\[1\].*\n|#> NULL\n)+)", t)
    assert len(blocks) == 100
    for k, body in blocks:
        for v in re.findall(r"(?<!\[)\b(\d+)\b(?!\])", body.replace("#>", "")):
            nz[int(k), int(v) - 1] = True
    out["predict_student_nonzero"] = nz

    # print.cv_sgdnet.html [inherits]: cv_sgdnet(mtcars$drat, mtcars$hp)
    t = _text("reference/print.cv_sgdnet.html")
    m1 = re.search(r"#> lambda_min Mean-Squared Error\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)", t)
    m2 = re.search(r"#> lambda_1se Mean-Squared Error\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)", t)
    up = re.findall(r"#> lambda_(?:min|1se)\s+(\d+\.\d+)\s*$", t, re.M)
    out["print_cv_mtcars"] = np.array([[float(v) for v in m1.groups()] + [float(up[0])],
                                       [float(v) for v in m2.groups()] + [float(up[1])]])   # alpha lambda mean sd lo up

    # print.sgdnet.html: sgdnet(with(mtcars, cbind(drat, hp)), mtcars$disp); print(fit, digits = 1)
    t = _text("reference/print.sgdnet.html")
    tab = re.findall(r"#> s\d+\s+(\d+)\s+(\S+)\s+(\S+)", t)
    assert len(tab) == 100
    out["print_mtcars_df"] = np.array([int(r[0]) for r in tab])
    out["print_mtcars_dev"] = np.array([float(r[1]) for r in tab])              # 1 significant digit
    out["print_mtcars_lambda"] = np.array([float(r[2]) for r in tab])           # 2 decimals

    # score.html: set.seed(1); wine; multinomial, nfolds = 5, alpha = c(0.5, 1); deviance at lambda_1se
    t = _text("reference/score.html")
    out["score_wine_deviance"] = np.array(float(re.search(r"#>\s+1\s*\n#>\s+(\d+\.\d+)", t).group(1)))
    assert float(out["score_wine_deviance"]) == 0.2717494

    np.savez_compressed(os.path.join(HERE, "refdocs.npz"), **out)
    print("wrote refdocs.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()

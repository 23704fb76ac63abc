"""Generate tests/golden/fu_goldens.npz: tables of the reference's channel model f_u (diffqc.cc:95-135, with my_expit :75-80
and bspline :82-93) computed BY THE REFERENCE'S OWN ARITHMETIC.

Run in the build container only (needs /root/reference and g++; neither exists / is needed on the GPU box):

    python tests/golden/make_fu_goldens.py            # regenerate and write the .npz
    python tests/golden/make_fu_goldens.py --check    # regenerate, compare with the committed .npz, write nothing

diffqc.cc as a whole needs Eigen and pybind11 (absent from this image), but its pulse functions need only <cmath> and <vector>.
This script slices, AT RUN TIME and into a temporary directory, the text of exactly those definitions out of
/root/reference/diffqc.cc -- the `Scalar` typedef, the three globals f_u reads (g_channels, g_duration, g_func_type) and the
functions my_expit, bspline, f_u, each located by its signature, not by line number -- puts the standard headers the file itself
includes in front (same order, minus pybind11 / Eigen: no stand-ins for anything), appends a driver `main` of its own that
reads the cases from stdin and prints f_u, and compiles that with g++.  Nothing of the reference is stored: the temporary
directory is deleted, the .npz holds inputs and outputs (numeric arrays) only.

Cases (the branches no sim_plain golden reaches): Legendre and bump bases; carrier w != 0 with both
quadratures vv[0] (cos) and vv[1] (sin); several channels per Hamiltonian sharing / not sharing a parameter row; a channel
whose amplitude is exactly zero and one just below / above the |N| < 1e-6 cutoff (diffqc.cc:128); |N| large enough for the
my_expit clamp at 32; t at the ends of the window, where the open-interval bump vanishes.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/diffqc.cc"
OUT = os.path.join(HERE, "fu_goldens.npz")

DRIVER = r"""
// driver (ours): cases on stdin, f_u on stdout with 17 significant digits
int main() {
  int n_cases;
  if (scanf("%d", &n_cases) != 1) return 2;
  for (int c = 0; c < n_cases; ++c) {
    int func_type, K, n_param, n_basis, n_t;
    double duration;
    if (scanf("%d %lf %d %d %d %d", &func_type, &duration, &K, &n_param, &n_basis, &n_t) != 6) return 2;
    g_func_type = func_type;
    g_duration = duration;
    g_channels.clear();
    for (int h = 0; h < K; ++h) {
      int nc;
      if (scanf("%d", &nc) != 1) return 2;
      std::vector<std::vector<Scalar>> chans;
      for (int i = 0; i < nc; ++i) {
        std::vector<Scalar> ch(4);
        for (int q = 0; q < 4; ++q)
          if (scanf("%lf", &ch[q]) != 1) return 2;
        chans.push_back(ch);
      }
      g_channels.push_back(chans);
    }
    std::vector<std::vector<std::vector<Scalar>>> vv(2, std::vector<std::vector<Scalar>>(n_param, std::vector<Scalar>(n_basis)));
    for (int a = 0; a < 2; ++a)
      for (int p = 0; p < n_param; ++p)
        for (int j = 0; j < n_basis; ++j)
          if (scanf("%lf", &vv[a][p][j]) != 1) return 2;
    for (int i = 0; i < n_t; ++i) {
      double t;
      if (scanf("%lf", &t) != 1) return 2;
      for (int h = 0; h < K; ++h) printf("%.17g\n", (double)f_u(h, t, vv));
    }
  }
  return 0;
}
"""


def _function_text(src, signature):
    """The definition that starts at `signature` up to its matching closing brace."""
    a = src.index(signature)
    i = src.index("{", a)
    depth = 0
    while True:
        ch = src[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                return src[a:i + 1]
        i += 1


def _line_with(src, needle):
    for ln in src.splitlines():
        if ln.strip() == needle:
            return ln
    raise SystemExit("reference line not found: " + needle)


def build_reference_fu(tmp):
    with open(REF) as f:
        src = f.read()
    parts = ["#include <stdio.h>", "#include <vector>", "#include <math.h>", "#include <cmath>",   # the file's own standard headers
             _line_with(src, "typedef double Scalar;"),
             _line_with(src, "std::vector<std::vector<std::vector<Scalar>>> g_channels;"),
             _line_with(src, "Scalar g_duration;"),
             _line_with(src, "int g_func_type; // 0: legendre. 1: b_spline"),
             _function_text(src, "Scalar my_expit(Scalar x)"),
             _function_text(src, "Scalar bspline(int b, int n_basis, Scalar t)"),
             _function_text(src, "Scalar f_u("),
             DRIVER]
    cc = os.path.join(tmp, "fu_ref.cc")
    exe = os.path.join(tmp, "fu_ref")
    with open(cc, "w") as f:
        f.write("\n".join(parts) + "\n")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, cc])
    return exe


def cases():
    rng = np.random.default_rng(20260403)
    out = []

    def case(name, func_type, duration, channels, vv, ts):
        out.append(dict(name=name, func_type=func_type, duration=duration, channels=channels, vv=np.asarray(vv, dtype=np.float64),
                        ts=np.asarray(ts, dtype=np.float64)))

    nb = 6
    ts = np.linspace(0.0, 2.5, 41)
    # one channel per Hamiltonian, carrier frequency w != 0, both quadratures
    for ft in (0, 1):
        vv = rng.normal(0.0, 1.0, (2, 3, nb))
        ch = [[[0.0, 1.3, 2.0 * np.pi * 0.7, 0.0]], [[0.0, 0.6, -3.1, 1.0]], [[0.0, 2.2, 0.0, 2.0]]]
        case("carrier_two_quadratures_ft{}".format(ft), ft, 2.5, ch, vv, ts)
    # several channels per Hamiltonian, shared parameter rows, a zero-amplitude channel
    for ft in (0, 1):
        vv = rng.normal(0.0, 0.8, (2, 4, nb))
        vv[:, 3, :] = 0.0                                   # parameter row 3: A = B = 0 exactly -> the |N| < 1e-6 branch
        ch = [[[0.0, 1.0, 5.0, 0.0], [1.0, 0.5, -2.0, 1.0]],
              [[0.0, 0.7, 1.5, 1.0], [0.0, 0.9, 0.0, 3.0], [2.0, 1.1, 4.0, 2.0]],
              [[0.0, 3.0, 0.3, 3.0]]]
        case("multi_channel_zero_row_ft{}".format(ft), ft, 2.5, ch, vv, ts)
    # amplitudes around the cutoff 1e-6 and around the my_expit clamp at 32 (constant basis function: Legendre P0 = 1)
    vv = np.zeros((2, 6, 3))
    for p, amp in enumerate((0.9e-6, 1.1e-6, 0.5, 31.0, 33.0, 40.0)):
        vv[0, p, 0] = amp * 0.6
        vv[1, p, 0] = amp * 0.8
    ch = [[[0.0, 1.0, 1.0, float(p)]] for p in range(6)]
    case("cutoff_and_clamp_ft0", 0, 1.0, ch, vv, np.linspace(0.0, 1.0, 9))
    # bump basis at the ends of the window and outside it (open interval), n_basis = 4 and 9
    for nbb in (4, 9):
        vv = rng.normal(0.0, 1.0, (2, 2, nbb))
        ch = [[[0.0, 1.0, 0.5, 0.0]], [[0.0, 1.0, 2.5, 1.0], [0.0, 0.4, -1.0, 0.0]]]
        case("bump_window_ends_nb{}".format(nbb), 1, 2.0, ch, vv, np.concatenate([[-0.25, 0.0, 1e-12], np.linspace(0.1, 1.9, 10), [2.0, 2.25]]))
    return out


def run_reference(exe, cs):
    lines = [str(len(cs))]
    for c in cs:
        vv = c["vv"]
        lines.append("{} {!r} {} {} {} {}".format(c["func_type"], float(c["duration"]), len(c["channels"]), vv.shape[1], vv.shape[2], len(c["ts"])))
        for chans in c["channels"]:
            lines.append(str(len(chans)))
            for ch in chans:
                lines.append(" ".join(repr(float(x)) for x in ch))
        lines.append(" ".join(repr(float(x)) for x in vv.ravel()))
        lines.append(" ".join(repr(float(x)) for x in c["ts"]))
    res = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    vals = np.array([float(x) for x in res.stdout.split()], dtype=np.float64)
    out, at = [], 0
    for c in cs:
        n = len(c["ts"]) * len(c["channels"])
        out.append(vals[at:at + n].reshape(len(c["ts"]), len(c["channels"])))
        at += n
    assert at == len(vals)
    return out


def generate():
    tmp = tempfile.mkdtemp(prefix="dq_fu_goldens_")
    try:
        exe = build_reference_fu(tmp)
        cs = cases()
        tables = run_reference(exe, cs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    arrays = {"names": np.array([c["name"] for c in cs])}
    for c, tab in zip(cs, tables):
        k = c["name"]
        arrays[k + "/func_type"] = np.array(c["func_type"])
        arrays[k + "/duration"] = np.array(c["duration"])
        arrays[k + "/vv"] = c["vv"]
        arrays[k + "/ts"] = c["ts"]
        nc = max(len(x) for x in c["channels"])
        ch = np.full((len(c["channels"]), nc, 4), np.nan)     # ragged channel lists padded with NaN rows
        for h, chans in enumerate(c["channels"]):
            ch[h, :len(chans)] = np.asarray(chans, dtype=np.float64)
        arrays[k + "/channels"] = ch
        arrays[k + "/f_u"] = tab
    return arrays


def main():
    arrays = generate()
    if "--check" in sys.argv:
        old = np.load(OUT)
        bad = [k for k in arrays if k not in old or not np.array_equal(old[k], arrays[k], equal_nan=(arrays[k].dtype.kind == "f"))]
        bad += [k for k in old.files if k not in arrays]
        print("{} arrays regenerated, {} differ from {}".format(len(arrays), len(bad), OUT))
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **arrays)
    print("wrote {} ({} arrays)".format(OUT, len(arrays)))


if __name__ == "__main__":
    main()

"""dev helper: whole-matrix comparison of two tools/gemm_bench GB_DUMP runs (e.g. old vs new bf16 kernel)"""
import sys, glob, numpy as np
a_pref, b_pref = sys.argv[1], sys.argv[2]
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
for fa in sorted(glob.glob(a_pref + "_*.bin")):
    fb = b_pref + fa[len(a_pref):]
    a, b = np.fromfile(fa, np.float32), np.fromfile(fb, np.float32)
    d = np.abs(a - b)
    print(fa.split("_")[-1], "n=%d max|a-b|=%.3g  count(>%g)=%d" % (a.size, d.max(), tol, int((d > tol).sum())))

"""Row a14 (bcf_call_glfgen + errmod_cal, bam2bcf.c:65-123 + HTSlib errmod.c): a SECOND WITNESS for the float outputs.

This is NOT reference evidence -- HTSlib's errmod.c / kfunc.c are absent from the reference tree and the reference's tests reach
errmod_cal only through 78 tview consensus characters (tests/test_oracle_goldens.py).  What this file adds is an independent
restatement of the same published model, written in Python from the algorithm's description (not from oracle/o_glf.c's code
structure), in two forms:

  1. IEEE form: every operation in the precision the C code uses (double; 80-bit long double for the binomial tail through
     numpy.longdouble; float for the final sums), pileup entries derived from the SAM text by this file's own ten-line pileup.
     Its qsum[4] / p[25] must equal the oracle's `glf` output BIT FOR BIT on generated columns (depth 60 with both strands and
     four quality levels; a 300x block where the 255-base cut applies).  Two implementations that agree to the last bit on
     ~10^5 floats share no transcription error in indexing, ordering or rounding sequence.
  2. High-precision form: the tables beta[q][n][k] (binomial tail ratio), lhet[n][k] and fk[n] evaluated with mpmath at 60
     digits from the closed forms, with the same Lanczos lgamma series the double code uses.  The double tables must lie within
     a few ulps of them: the long-double / double evaluation is numerically sound (no cancellation that would make the tables
     platform dependent), so that (1)'s agreement is agreement on the MODEL and not on a shared rounding accident.

The engine (k_glf_cols, kernels_glf.hip) is compared with the oracle on the GPU by tests/test_gpu_glf.py; this test needs no GPU."""
import math
import os
import subprocess

import numpy as np
import pytest

from synth import write_synth_sam

M_LN10 = 2.30258509299404568402
M_LN2 = 0.693147180559945309417
LANCZOS = [(0.1659470187408462e-06, 7), (0.9934937113930748e-05, 6), (-0.1385710331296526, 5), (12.50734324009056, 4),
           (-176.6150291498386, 3), (771.3234287757674, 2), (-1259.139216722289, 1)]


def kf_lgamma(z):
    x = 0.0
    for c, d in LANCZOS:
        x += c / (z + d)
    x += 676.5203681218835 / z
    x += 0.9999999999995183
    return math.log(x) - 5.58106146679532777 - z + (z - 0.5) * math.log(z + 6.5)


class ErrMod:
    """errmod_init of the published model: fk, beta, lhet (eta = 0.03)."""

    def __init__(self, depcorr, qs=range(1, 64), ns=range(1, 256)):
        eta = 0.03
        self.fk = [1.0] + [math.pow(1.0 - depcorr, n) * (1.0 - eta) + eta for n in range(1, 256)]
        lg = [0.0] + [kf_lgamma(float(i)) for i in range(1, 258)]            # lg[i] = lgamma(i)
        self.lC = {}
        for n in range(1, 256):
            for k in range(1, n + 1):
                self.lC[(n, k)] = lg[n + 1] - lg[k + 1] - lg[n - k + 1]
        self.beta = {}
        ld = np.longdouble
        with np.errstate(divide="ignore"):
            for q in qs:
                e = math.pow(10.0, -q / 10.0)
                le, le1 = math.log(e), math.log(1.0 - e)
                for n in ns:
                    s1 = ld(0.0)
                    for k in range(n, -1, -1):
                        arg = self.lC.get((n, k), 0.0) + k * le + (n - k) * le1          # double, like the C expression
                        s = s1 + np.exp(ld(arg))
                        self.beta[(q, n, k)] = float(ld(-10.0 / M_LN10) * np.log(s1 / s))
                        s1 = s

    def lhet(self, n, k):
        return self.lC.get((n, k), 0.0) - M_LN2 * n


def errmod_cal(em, bases, m=5):
    """25 floats (as numpy float32) for one column's packed bases (q << 5 | strand << 4 | base); the cut to 255 entries happens here"""
    f32, f64 = np.float32, np.float64
    q = [f32(0.0)] * (m * m)
    n = len(bases)
    if n == 0:
        return q, 0
    cut = 0
    if n > 255:
        bases, n, cut = bases[:255], 255, 1
    bases = sorted(bases)
    fsum, bsum, c, w = [0.0] * 16, [0.0] * 16, [0] * 16, [0] * 32
    for b in reversed(bases):
        qual = min(63, max(4, b >> 5))
        bs, base = b & 0x1f, b & 0xf
        fsum[base] += em.fk[w[bs]]
        bsum[base] += em.fk[w[bs]] * em.beta[(qual, n, c[base])]
        c[base] += 1
        w[bs] += 1
    for j in range(m):
        t1, t2 = f32(0.0), 0
        for k in range(m):
            if k != j:
                t1 = f32(f64(t1) + bsum[k]); t2 += c[k]
        if t2:
            q[j * m + j] = t1
        for k in range(j + 1, m):
            cjk = c[j] + c[k]
            t1, t2 = f32(0.0), 0
            for i in range(m):
                if i != j and i != k:
                    t1 = f32(f64(t1) + bsum[i]); t2 += c[i]
            v = -4.343 * em.lhet(cjk, c[k]) + (float(t1) if t2 else 0.0)
            if not t2:
                v = -4.343 * em.lhet(cjk, c[k])
            q[j * m + k] = q[k * m + j] = f32(v)
        for k in range(m):
            if q[j * m + k] < 0.0:
                q[j * m + k] = f32(0.0)
    return q, cut


NT16 = {"A": 0, "C": 1, "G": 2, "T": 3}


def columns_of(sam_path, min_baseQ=13, capQ=60):
    """this file's own pileup for reads whose CIGAR is one M operation: per column the packed entries in file order + qsum"""
    reads = []
    n_ref = 0
    for line in open(sam_path):
        if line.startswith("@"):
            if line.startswith("@SQ"):
                n_ref = int(line.split("LN:")[1].split()[0])
            continue
        f = line.rstrip("\n").split("\t")
        flag, pos, mapq, cigar, seq, qual = int(f[1]), int(f[3]) - 1, int(f[4]), f[5], f[9], f[10]
        assert cigar == "%dM" % len(seq), cigar
        if flag & 4:
            continue
        reads.append((pos, flag, mapq, seq, qual))
    cols = [[] for _ in range(n_ref)]
    qsums = [[np.float32(0)] * 4 for _ in range(n_ref)]
    nplp = [0] * n_ref
    for pos, flag, mapq, seq, qual in reads:
        mq = min(mapq if mapq < 255 else 20, capQ)
        for i, (ch, qc) in enumerate(zip(seq, qual)):
            col = pos + i
            nplp[col] += 1
            q = ord(qc) - 33
            if q < min_baseQ:
                continue
            q = max(4, min(63, min(min(q, 99), mq)))
            b = NT16.get(ch, 4)
            cols[col].append(q << 5 | (1 if flag & 16 else 0) << 4 | b)
            if b < 4:
                qsums[col][b] = np.float32(qsums[col][b] + np.float32(q))
    return cols, qsums, nplp


def _bits(x):
    return "%08x" % int(np.array([x], dtype=np.float32).view(np.uint32)[0])


@pytest.mark.parametrize("shape", [dict(n_ref=1500, depth=60, read_len=100, seed=71), dict(n_ref=500, depth=300, read_len=100, seed=72)], ids=["60x", "300x_cut"])
def test_ieee_restatement_equals_the_oracle_bit_for_bit(tmp_path, oracle_bin, shape):
    sam, fa = write_synth_sam(str(tmp_path), paired=False, indel_rate=0.0, sub_rate=0.02, **shape)
    out = subprocess.run([oracle_bin, "glf", "-f", fa, sam], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    cols, qsums, nplp = columns_of(sam)
    ns = sorted({min(len(c), 255) for c in cols if c})
    em = ErrMod(1.0 - 0.83, qs=sorted({min(63, max(4, b >> 5)) for c in cols for b in c}), ns=ns)
    checked = cut_cols = 0
    for line in out:
        f = line.split("\t")
        col = int(f[1]) - 1
        assert int(f[2]) == nplp[col] and int(f[3]) == len(cols[col])
        p, cut = errmod_cal(em, cols[col])
        assert int(f[4]) == cut
        cut_cols += cut
        assert f[5] == ",".join(_bits(x) for x in qsums[col]), (col, "qsum")
        assert f[6] == ",".join(_bits(x) for x in p), (col, "p")
        checked += 29
    assert checked > 29 * 0.9 * shape["n_ref"]
    assert (cut_cols > 100) == (shape["depth"] > 255)


def test_tables_are_within_ulps_of_a_60_digit_evaluation():
    """beta, lhet, fk of the IEEE form against mpmath at 60 digits (the same Lanczos series, the binomial tail summed exactly)."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    em = ErrMod(1.0 - 0.83, qs=[4, 13, 25, 37, 63], ns=[1, 2, 7, 30, 60, 120, 255])

    def lgam(z):
        x = sum(mp.mpf(c) / (z + d) for c, d in LANCZOS) + mp.mpf(676.5203681218835) / z + mp.mpf(0.9999999999995183)
        return mp.log(x) - mp.mpf(5.58106146679532777) - z + (z - mp.mpf(0.5)) * mp.log(z + mp.mpf(6.5))

    def lC(n, k):
        return lgam(mp.mpf(n + 1)) - lgam(mp.mpf(k + 1)) - lgam(mp.mpf(n - k + 1)) if k >= 1 else mp.mpf(0)

    def err_units(got, exact, scale):
        """absolute error in units of 2^-53 x `scale` (the size of the operands the double code rounds on the way)"""
        return float(abs(mp.mpf(got) - exact) / (mp.mpf(2) ** -53 * max(1.0, scale)))

    worst_beta = worst_lhet = worst_fk = 0
    worst_beta_rel = 0
    for q in (4, 13, 25, 37, 63):
        e = mp.mpf(math.pow(10.0, -q / 10.0))                      # the double the C code holds
        for n in (1, 2, 7, 30, 60, 120, 255):
            s1 = mp.mpf(0)
            big = max(float(n * abs(mp.log(e))), float(lgam(mp.mpf(n + 1))))       # size of the exponent's argument: its rounding IS the error
            for k in range(n, -1, -1):
                s = s1 + mp.exp(lC(n, k) + k * mp.log(e) + (n - k) * mp.log(1 - e))
                if s1 != 0:
                    exact = -10 / mp.mpf(M_LN10) * mp.log(s1 / s)
                    got = em.beta[(q, n, k)]
                    worst_beta = max(worst_beta, err_units(got, exact, big * 10 / M_LN10))
                    if exact > 1e-3:
                        worst_beta_rel = max(worst_beta_rel, float(abs(mp.mpf(got) - exact) / exact))
                else:
                    assert em.beta[(q, n, k)] == math.inf
                s1 = s
    for n in (1, 2, 7, 30, 60, 120, 255):
        for k in range(1, n + 1):
            worst_lhet = max(worst_lhet, err_units(em.lhet(n, k), lC(n, k) - mp.mpf(M_LN2) * n, float(lgam(mp.mpf(n + 1)))))
    for n in range(1, 256):
        worst_fk = max(worst_fk, err_units(em.fk[n], mp.mpf(1.0 - (1.0 - 0.83)) ** n * mp.mpf(1.0 - 0.03) + mp.mpf(0.03), 1.0))
    # measured: fk 1, lhet 11.5, beta 5.9 units, beta 3.8e-13 relative: the double tables carry the rounding of a handful of operations on
    # operands of that size (lC is a difference of three lgamma values up to ~1200; the exponent's argument a double sum of the same size),
    # nothing more -- far inside the float precision (6e-8) the results are stored in
    assert worst_fk <= 2, worst_fk
    assert worst_lhet <= 32, worst_lhet
    assert worst_beta <= 32, worst_beta
    assert worst_beta_rel < 1e-11, worst_beta_rel

// glf_tables.cpp -- coefficient tables of the genotype-likelihood error model (HTSlib errmod.c cal_coef, kfunc.c kf_lgamma;
// absent from the reference tree, restated from the published model -- see kernels_glf.hip and DESIGN.md, row a14).
// Host code, computed once per theta and uploaded; compiled with -ffp-contract=off so that the arithmetic is the plain
// IEEE sequence a default x86-64 build of HTSlib performs.
#include "glf_tables.h"
#include <cmath>

namespace sta {

static double lanczos_lgamma(double z)
{
    double x = 0;
    x += 0.1659470187408462e-06 / (z + 7);
    x += 0.9934937113930748e-05 / (z + 6);
    x -= 0.1385710331296526 / (z + 5);
    x += 12.50734324009056 / (z + 4);
    x -= 176.6150291498386 / (z + 3);
    x += 771.3234287757674 / (z + 2);
    x -= 1259.139216722289 / (z + 1);
    x += 676.5203681218835 / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}

void glf_tables(double depcorr, std::vector<double> &t)
{
    const double eta = 0.03;
    t.assign(GLF_TAB_DOUBLES, 0.0);
    double *fk = t.data() + GLF_FK_OFF, *beta = t.data() + GLF_BETA_OFF, *lhet = t.data() + GLF_LHET_OFF;
    fk[0] = 1.0;
    for (int n = 1; n < 256; ++n) fk[n] = pow(1. - depcorr, n) * (1.0 - eta) + eta;
    std::vector<double> lC(256 * 256, 0.0);
    for (int n = 1; n != 256; ++n) {
        double lgn = lanczos_lgamma(n + 1);
        for (int k = 1; k <= n; ++k) lC[(size_t)(n << 8 | k)] = lgn - lanczos_lgamma(k + 1) - lanczos_lgamma(n - k + 1);
    }
    for (int q = 1; q != 64; ++q) {
        double e = pow(10.0, -q / 10.0);
        double le = log(e), le1 = log(1.0 - e);
        for (int n = 1; n <= 255; ++n) {
            double *b = beta + (q << 16 | n << 8);
            long double sum = 0.0, sum1 = 0.0;
            for (int k = n; k >= 0; --k, sum1 = sum) {
                sum = sum1 + expl(lC[(size_t)(n << 8 | k)] + k * le + (n - k) * le1);
                b[k] = -10. / M_LN10 * logl(sum1 / sum);
            }
        }
    }
    for (int n = 0; n < 256; ++n)
        for (int k = 0; k < 256; ++k) lhet[n << 8 | k] = lC[(size_t)(n << 8 | k)] - M_LN2 * n;
}

}  // namespace sta

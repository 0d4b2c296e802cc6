// cons_host.cpp -- lookup tables of the consensus callers, built on the host with libm exactly where the reference builds
// them (consensus_init, bam_consensus.c:740-883; q2p[] / mqual_pow_1m[] from the formulas bam_consensus_tab.h:27-37 documents)
// and uploaded once per run.
#include "cons_host.h"
#include <cmath>
#include <cstring>

namespace sta {

static void init_probs(double p_het, double p_indel, double het_scale, double poly_mul, const int32_t qcal[3][101], bool like_116, cons::Probs &cp)
{
    cp.poly_mul = poly_mul;
    double prior[25];
    for (int i = 0; i < 25; ++i) prior[i] = p_het / 6;
    for (int i = 0; i < 25; i += 6) prior[i] = 1;
    for (int i = 4; i < 24; i += 5) prior[i] = p_indel / 6;
    for (int i = 20; i < 24; ++i) prior[i] = p_indel / 6;
    const int tri[15] = { 0, 1, 2, 3, 4, 6, 7, 8, 9, 12, 13, 14, 18, 19, 24 };
    for (int j = 0; j < 15; ++j) cp.lprior15[j] = log(prior[tri[j]]);
    const int32_t *smap = qcal[0], *umap = qcal[1], *omap = qcal[2];
    for (int i = 1; i < 101; ++i) {
        double prob = 1 - pow(10, -smap[i] / 10.0);
        cp.pMM[i] = log(prob);
        cp.pxx[i] = log((1 - prob) / 3);
        cp.pxM[i] = log((exp(cp.pMM[i]) + exp(cp.pxx[i])) / 2);
        cp.pxM[i] += log(het_scale);
        if (like_116) {
            cp.pmm[i] = cp.pMM[i];
            cp.poM[i] = cp.pum[i] = cp.pxM[i];
            cp.pox[i] = cp.poo[i] = cp.puu[i] = cp.pxx[i];
            continue;
        }
        prob = 1 - pow(10, -omap[i] / 10.0);
        cp.poo[i] = log((1 - prob) / 3);
        if (cp.poo[i] > cp.pMM[i] - .5) cp.poo[i] = cp.pMM[i] - .5;
        cp.pox[i] = log((exp(cp.poo[i]) + exp(cp.pxx[i])) / 2);
        cp.poM[i] = log((exp(cp.poo[i]) + exp(cp.pMM[i])) / 2);
        if (cp.poM[i] > cp.pxM[i] + .5) cp.poM[i] = cp.pxM[i] + .5;
        prob = 1 - pow(10, -umap[i] / 10.0);
        cp.pmm[i] = log(prob);
        cp.puu[i] = log((1 - prob) / 3);
        if (cp.puu[i] > cp.pMM[i] - .5) cp.puu[i] = cp.pMM[i] - .5;
        cp.pum[i] = log((exp(cp.puu[i]) + exp(cp.pmm[i])) / 2);
    }
    double *all[9] = { cp.pMM, cp.pxx, cp.pxM, cp.pmm, cp.poo, cp.pox, cp.poM, cp.puu, cp.pum };
    for (double *a : all) a[0] = a[1];
}

void cons_build_tables(const sta_cons_params &p, cons::Tables &t)
{
    memset(&t, 0, sizeof t);
    for (int i = -500; i <= 500; ++i) { t.e_tab[500 + i] = exp(i); t.e_tab2[500 + i] = exp(i / 10.); }
    for (int i = 0; i <= 100; ++i) t.q2p[i] = pow(10, -i / 10.0);
    for (int i = 0; i < 255; ++i) t.mqual_pow_1m[i] = pow(10, -(i * .9) / 10.0);
    t.mqual_pow_1m[255] = t.mqual_pow_1m[10];
    for (int i = 0; i < 256; ++i) t.ph2err[i] = pow(10, i / -10.0);
    if (p.mode == STA_CONS_SIMPLE) return;
    // main_consensus, bam_consensus.c:3424-3445.  MODE_BAYES_116 as a consensus_init() mode is never passed by the command,
    // so the samtools-1.16 branch of the table code is dead there; -m bayesian_116 only changes nm_init.
    if (p.mode == STA_CONS_PRECISE) init_probs(p.P_het, p.P_indel, 0.3 * p.het_scale, p.homopoly_redux, p.qcal, false, t.precise);
    if (p.mode == STA_CONS_MIXED) init_probs(pow(p.P_het, 0.7), pow(p.P_indel, 0.7), 0.3 * p.het_scale, p.homopoly_redux, p.qcal, false, t.precise);
    init_probs(p.P_het, p.P_indel, p.het_scale, p.mode == STA_CONS_RECALL ? p.homopoly_redux : 0.01, p.qcal, false, t.recall);
}

cons::Par cons_par(const sta_cons_params &p)
{
    cons::Par o; memset(&o, 0, sizeof o);
    o.mode = p.mode; o.use_qual = p.use_qual; o.min_qual = p.min_qual; o.adj_qual = p.adj_qual; o.use_mqual = p.use_mqual;
    o.nm_adjust = p.nm_adjust; o.nm_halo = p.nm_halo; o.sc_cost = p.sc_cost; o.low_mqual = p.low_mqual; o.high_mqual = p.high_mqual;
    o.min_depth = p.min_depth; o.cons_cutoff = p.cons_cutoff; o.ambig = p.ambig; o.default_qual = p.default_qual;
    o.excl_flags = p.excl_flags; o.incl_flags = p.incl_flags; o.min_mqual = p.min_mqual; o.homopoly_on = p.homopoly_fix != 0;
    o.scale_mqual = p.scale_mqual; o.call_fract = p.call_fract; o.het_fract = p.het_fract; o.homopoly_fix = p.homopoly_fix;
    return o;
}

}  // namespace sta

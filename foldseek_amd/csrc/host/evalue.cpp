// evalue.cpp -- the e-value network of structurealign (reference F/src/strucclustutils/EvalueNeuralNet.{h,cpp};
// dense layers as F/lib/kerasify/keras_model.cpp:187-212 applies them: float accumulate in (i, j) order, bias, ReLU).
#include "hostlib.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>

namespace fsh {

std::string libraryDir() {
    Dl_info info;
    if (dladdr((void *) &libraryDir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

bool Evaluer::load(const char *path, uint64_t dbResidues, std::string &err) {
    std::string p = path ? std::string(path) : libraryDir() + "/data/evalue_nn.bin";
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { err = "cannot open e-value network " + p; return false; }
    char magic[4];
    uint32_t nl = 0;
    bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "FSNN", 4) == 0 && fread(&nl, 4, 1, f) == 1 && nl > 0 && nl < 64;
    if (ok) {
        layers.resize(nl);
        for (auto &l : layers) {
            uint32_t h[4];
            ok = ok && fread(h, 4, 4, f) == 4;
            l.rows = h[0]; l.cols = h[1]; l.nbias = h[2]; l.act = h[3];
            ok = ok && l.rows <= 4096 && l.cols <= 4096 && l.nbias <= l.cols;
        }
        for (auto &l : layers) {
            if (!ok) break;
            l.w.resize((size_t) l.rows * l.cols);
            l.b.resize(l.nbias);
            ok = ok && fread(l.w.data(), 4, l.w.size(), f) == l.w.size();
            ok = ok && fread(l.b.data(), 4, l.b.size(), f) == l.b.size();
        }
    }
    fclose(f);
    if (!ok) { err = "malformed e-value network file " + p; layers.clear(); return false; }
    logDbResidueCount = log(static_cast<double>(dbResidues));
    return true;
}

void Evaluer::predictMuLambda(const uint8_t *seq, unsigned int L, int alphabetSize, double *lambda, double *mu) const {
    std::vector<float> in(alphabetSize + 1, 0.0f), tmp;
    for (unsigned int i = 0; i < L; i++) in[seq[i]]++;
    in[alphabetSize] = L;
    for (const Layer &l : layers) {
        tmp.assign(l.cols, 0.0f);
        for (uint32_t i = 0; i < l.rows; i++)
            for (uint32_t j = 0; j < l.cols; j++) tmp[j] = fmaf(in[i], l.w[(size_t) i * l.cols + j], tmp[j]);   // fused in the reference build
        for (uint32_t i = 0; i < l.nbias; i++) tmp[i] += l.b[i];
        if (l.act == 2)   // ReLU
            for (uint32_t j = 0; j < l.cols; j++)
                if (tmp[j] < 0.0) tmp[j] = 0.0;
        in = tmp;
    }
    const double mu1 = 0.17518475184751847, sigma1 = 0.03260331312698818;
    const double mu2 = -2.5569312493124934, sigmal2 = 0.4353169278257701;
    *lambda = std::fma((double) in[0], sigma1, mu1);   // fused in the reference build
    *mu = std::fma((double) in[1], sigmal2, mu2);
}

static double computePvalue(double score, double lambda_, double mu) {
    double h = lambda_ * (score - mu);
    if (h > 10) return -h;
    else if (h < -2.5) return -exp(-exp(-h));
    else return log((1.0 - exp(-exp(-h))));
}

double Evaluer::computeEvalueCorr(double score, double lambda_, double mu) const {
    double logPVal = computePvalue(score, lambda_, mu);
    double dbSizeTimesLogPVal = logPVal + logDbResidueCount;
    double evalue = exp(dbSizeTimesLogPVal);
    return pow(evalue, 0.32);
}

} // namespace fsh

struct fshost_evaluer { fsh::Evaluer e; };
static thread_local std::string g_evalErr;

extern "C" {
fshost_evaluer *fshost_evaluer_create(const char *nnPath, uint64_t dbResidues) {
    fshost_evaluer *h = new fshost_evaluer();
    if (!h->e.load(nnPath, dbResidues, g_evalErr)) { delete h; return nullptr; }
    return h;
}
void fshost_evaluer_free(fshost_evaluer *e) { delete e; }
void fshost_predict_mu_lambda(const fshost_evaluer *e, const uint8_t *q3di, unsigned int L, int alphabetSize, double *lambda, double *mu) {
    e->e.predictMuLambda(q3di, L, alphabetSize, lambda, mu);
}
double fshost_evalue_corr(const fshost_evaluer *e, double score, double lambda, double mu) { return e->e.computeEvalueCorr(score, lambda, mu); }
}

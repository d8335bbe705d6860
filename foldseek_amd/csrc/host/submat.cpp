// submat.cpp -- substitution matrices, composition bias and query profiles (host side of the hot path).
// Mirrors, for bit-identical integers, the reference's
//   SubstitutionMatrix::SubstitutionMatrix / readProbMatrix   (M/src/commons/SubstitutionMatrix.cpp:12-58,317-421)
//   BaseMatrix::computeBackground / generateSubMatrix          (M/src/commons/BaseMatrix.cpp:96-159)
//   SubstitutionMatrix::calcLocalAaBiasCorrection              (M/src/commons/SubstitutionMatrix.cpp:79-109)
//   SubstitutionMatrix::setupLetterMapping                     (M/src/commons/SubstitutionMatrix.cpp:255-296)
// The float/double expression shapes are kept exactly (float accumulators, double background) because the results
// are rounded to integers that feed the device kernels.
#include "hostlib.h"
#include "fs_params.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

namespace fsh {

static const double kAnyBack = 1E-5;   // BaseMatrix::ANY_BACK

bool Matrix::build(const std::vector<double> &score, const std::vector<double> &back, double lambda,
                   const std::string &letters, float bitFactorF, float scoreBiasF) {
    n = (int) letters.size();
    if (n < 2 || n > 32 || letters[n - 1] != 'X' || (int) back.size() < n || (int) score.size() < n * n) return false;
    this->letters = letters;
    const double bitFactor = (double) bitFactorF;
    const double scoringBias = (double) scoreBiasF;
    const int x = n - 1;
    std::vector<double> p(back.begin(), back.begin() + n);
    bool xIsPositive = false;
    for (int j = 0; j < n; j++)
        if (score[x * n + j] > 0 || score[j * n + x] > 0) { xIsPositive = true; break; }
    if (!xIsPositive)
        for (int i = 0; i < n - 1; i++) p[i] = p[i] * (1.0 - p[x]);
    std::vector<double> prob((size_t) n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) prob[i * n + j] = std::exp(lambda * score[i * n + j]) * p[i] * p[j];
    pBack = p;
    std::vector<double> marg(n);
    for (int i = 0; i < n; i++) {
        marg[i] = 0;
        for (int j = 0; j < n; j++) marg[i] += prob[i * n + j];
    }
    marg[n - 1] = kAnyBack;
    sub.assign((size_t) n * n, 0);
    tiny.assign((size_t) n * n, 0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double sm = std::log2(prob[i * n + j] / (marg[i] * marg[j]));
            double v = (bitFactor * sm + scoringBias);
            short s = (v < 0.0) ? v - 0.5 : v + 0.5;
            sub[i * n + j] = s;
            tiny[i * n + j] = (int8_t) s;
        }
    // letter mapping: header letters first, then the amino-acid aliases
    for (int c = 0; c < 256; c++) aa2num[c] = 255;
    for (int i = 0; i < n; i++) aa2num[(unsigned char) letters[i]] = (uint8_t) i;
    const std::string need = "ATGCDEFHIKLMNPQRSVWYX";
    bool allAA = true;
    for (char c : need) allAA = allAA && aa2num[(unsigned char) c] != 255;
    uint8_t map[256];
    for (int letter = 0; letter < 256; letter++) {
        const unsigned char up = (unsigned char) std::toupper(letter);
        uint8_t v;
        if (allAA) {
            switch (up) {
                case 'A': case 'T': case 'G': case 'C': case 'D': case 'E': case 'F': case 'H': case 'I': case 'K': case 'L':
                case 'M': case 'N': case 'P': case 'Q': case 'R': case 'S': case 'V': case 'W': case 'Y': case 'X':
                    v = aa2num[up]; break;
                case 'J': v = aa2num[(int) 'L']; break;
                case 'Z': v = aa2num[(int) 'E']; break;
                case 'B': v = aa2num[(int) 'D']; break;
                default: v = aa2num[(int) 'X']; break;
            }
        } else {
            v = (aa2num[up] == 255) ? (uint8_t) (n - 1) : aa2num[up];
        }
        map[letter] = v;
    }
    memcpy(aa2num, map, 256);
    return true;
}

bool Matrix::parse(const char *text, float bitFactor, float scoreBias) {
    std::istringstream in(text);
    std::string line, letters;
    std::vector<double> back, score;
    double lambda = 0;
    bool hasLambda = false, hasBack = false, started = false;
    std::vector<std::vector<double>> rows;
    std::string rowLetters;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '#') {
            if (line.find("# Background (precomputed optional):") == 0) {
                std::istringstream ls(line.substr(line.find(':') + 1));
                double f;
                while (ls >> f) back.push_back(f);
                hasBack = true;
            }
            if (line.find("# Lambda     (precomputed optional):") == 0) {
                lambda = strtod(line.c_str() + line.find(':') + 1, NULL);
                hasLambda = true;
            }
            continue;
        }
        std::istringstream ls(line);
        std::vector<std::string> w;
        std::string tok;
        while (ls >> tok) w.push_back(tok);
        if (w.size() <= 1) continue;
        if (!started) {
            for (auto &t : w) {
                if (!isalpha((unsigned char) t[0])) return false;
                letters.push_back((char) toupper((unsigned char) t[0]));
            }
            started = true;
            continue;
        }
        if (!isalpha((unsigned char) w[0][0]) || w.size() < letters.size() + 1) return false;
        rowLetters.push_back((char) toupper((unsigned char) w[0][0]));
        std::vector<double> r;
        for (size_t i = 0; i < letters.size(); i++) r.push_back(strtod(w[i + 1].c_str(), NULL));
        rows.push_back(r);
    }
    // lambda / background estimation (M/src/commons/LambdaCalculation.cpp) is not part of the hot path
    if (!hasLambda || !hasBack || letters.empty() || letters.find('X') == std::string::npos) return false;
    const int nn = (int) letters.size();
    score.assign((size_t) nn * nn, 0.0);
    for (size_t k = 0; k < rows.size(); k++) {
        size_t i = letters.find(rowLetters[k]);
        if (i == std::string::npos) return false;
        for (int j = 0; j < nn; j++) score[i * nn + j] = rows[k][j];
    }
    return build(score, back, lambda, letters, bitFactor, scoreBias);
}

bool Matrix::builtin(int which, float bitFactor, float scoreBias) {
    if (which == FSHOST_MAT_3DI) {
        std::vector<double> s(FS_MAT3DI_SCORE, FS_MAT3DI_SCORE + FS_MAT3DI_N * FS_MAT3DI_N);
        std::vector<double> b(FS_MAT3DI_BACK, FS_MAT3DI_BACK + FS_MAT3DI_N);
        return build(s, b, FS_MAT3DI_LAMBDA, FS_MAT3DI_LETTERS, bitFactor, scoreBias);
    } else if (which == FSHOST_MAT_BLOSUM62) {
        std::vector<double> s(FS_BLOSUM62_SCORE, FS_BLOSUM62_SCORE + FS_BLOSUM62_N * FS_BLOSUM62_N);
        std::vector<double> b(FS_BLOSUM62_BACK, FS_BLOSUM62_BACK + FS_BLOSUM62_N);
        return build(s, b, FS_BLOSUM62_LAMBDA, FS_BLOSUM62_LETTERS, bitFactor, scoreBias);
    }
    return false;
}

void compBias(const Matrix &m, const uint8_t *seq, int N, float scale, float *out) {
    const int windowSize = 40;
    const int n = m.n;
    for (int i = 0; i < N; i++) {
        const int minPos = std::max(0, (i - windowSize / 2));
        const int maxPos = std::min(N, (i + windowSize / 2));
        const int windowLength = maxPos - minPos;
        int sumSubScores = 0;
        const short *subMat = &m.sub[(size_t) seq[i] * n];
        for (int j = minPos; j < maxPos; j++) sumSubScores += subMat[seq[j]];
        sumSubScores -= subMat[seq[i]];
        float deltaS_i = (float) sumSubScores;
        deltaS_i /= -1.0 * static_cast<float>(windowLength);
        // the reference binary (GCC, -mfma, default fp-contract) fuses this multiply-add; say so explicitly
        for (int a = 0; a < n; a++) deltaS_i = (float) std::fma(m.pBack[a], (double) static_cast<float>(subMat[a]), (double) deltaS_i);
        out[i] = scale * deltaS_i;
    }
}

void roundBias(const float *cb, int L, int8_t *out) {
    for (int i = 0; i < L; i++) out[i] = (int8_t) ((cb[i] < 0.0) ? cb[i] - 0.5 : cb[i] + 0.5);
}

int prefilterProfile(const Matrix &m, const uint8_t *q, int L, bool compBiasOn, float scale, int8_t *pssm, int *scoreCap) {
    const int n = m.n;
    std::vector<float> cbf(L, 0.0f);
    std::vector<int8_t> cb(L, 0);
    if (compBiasOn) {
        compBias(m, q, L, scale, cbf.data());
        roundBias(cbf.data(), L, cb.data());
    }
    // bias of the uint8 CPU kernel: |min(mat)| + |min(0, min(cb))|
    int compositionBias = 0;
    for (int i = 0; i < L; i++) compositionBias = std::min(compositionBias, (int) cb[i]);
    int bias = 0;
    for (int i = 0; i < n * n; i++) bias = std::min(bias, (int) m.tiny[i]);
    bias = std::abs(bias) + std::abs(compositionBias);
    *scoreCap = 255 - bias;
    for (int a = 0; a < n; a++)
        for (int i = 0; i < L; i++) {
            if (q[i] >= n) return FSGPU_E_ARG;
            short b = compBiasOn ? static_cast<short>((cbf[i] < 0.0) ? (cbf[i] - 0.5) : (cbf[i] + 0.5)) : 0;
            pssm[(size_t) a * L + i] = (int8_t) (m.sub[(size_t) a * n + q[i]] + b);
        }
    return FSGPU_OK;
}

int alignProfiles(const Matrix &mAA, const Matrix &m3Di, const uint8_t *qAA, const uint8_t *q3Di, int L, bool compBiasOn,
                  float scale3Di, int16_t *pAA, int16_t *p3Di, int8_t *cbAAout, int8_t *cbSSout) {
    const int n = m3Di.n;
    if (mAA.n != n) return FSGPU_E_ARG;
    std::vector<int8_t> cbAA(L, 0), cbSS(L, 0);
    if (compBiasOn) {
        std::vector<float> tmp(L);
        compBias(mAA, qAA, L, 1.0, tmp.data());            // AA bias, scale 1.0
        roundBias(tmp.data(), L, cbAA.data());
        compBias(mAA, q3Di, L, scale3Di, tmp.data());      // 3Di bias against the AA matrix (sic)
        roundBias(tmp.data(), L, cbSS.data());
    }
    for (int i = 0; i < L; i++) if (qAA[i] >= n || q3Di[i] >= n) return FSGPU_E_ARG;
    if (p3Di)                                   // p3Di == NULL: the caller wants the biases only
        for (int a = 0; a < n; a++)
            for (int i = 0; i < L; i++) {
                if (pAA) pAA[(size_t) a * L + i] = (int16_t) (mAA.tiny[a * n + qAA[i]] + cbAA[i]);
                p3Di[(size_t) a * L + i] = (int16_t) (m3Di.tiny[a * n + q3Di[i]] + cbSS[i]);
            }
    if (cbAAout) memcpy(cbAAout, cbAA.data(), L);
    if (cbSSout) memcpy(cbSSout, cbSS.data(), L);
    return FSGPU_OK;
}

} // namespace fsh

struct fshost_matrix { fsh::Matrix m; };

extern "C" {

fshost_matrix *fshost_matrix_create(int which, float bitFactor, float scoreBias) {
    fshost_matrix *h = new fshost_matrix();
    if (!h->m.builtin(which, bitFactor, scoreBias)) { delete h; return nullptr; }
    return h;
}
fshost_matrix *fshost_matrix_from_text(const char *text, float bitFactor, float scoreBias) {
    if (!text) return nullptr;
    fshost_matrix *h = new fshost_matrix();
    if (!h->m.parse(text, bitFactor, scoreBias)) { delete h; return nullptr; }
    return h;
}
fshost_matrix *fshost_matrix_from_scores(const int16_t *scores, int n, const double *pBack) {
    if (!scores || !pBack || n < 1 || n > 32) return nullptr;
    fshost_matrix *h = new fshost_matrix();
    fsh::Matrix &m = h->m;
    m.n = n;
    m.sub.assign(scores, scores + (size_t) n * n);
    m.tiny.resize((size_t) n * n);
    for (size_t i = 0; i < (size_t) n * n; i++) m.tiny[i] = (int8_t) scores[i];
    m.pBack.assign(pBack, pBack + n);
    m.letters.assign((size_t) n, '?');
    for (int i = 0; i < 256; i++) m.aa2num[i] = (uint8_t) (n - 1);           // no alphabet: callers of this variant pass numeric codes
    return h;
}
void fshost_matrix_free(fshost_matrix *m) { delete m; }
int fshost_matrix_size(const fshost_matrix *m) { return m ? m->m.n : 0; }
const int16_t *fshost_matrix_scores(const fshost_matrix *m) { return m ? m->m.sub.data() : nullptr; }
const char *fshost_matrix_text(int which, size_t *len) {
    if (which != FSHOST_MAT_3DI) { if (len) *len = 0; return nullptr; }
    if (len) *len = sizeof(FS_MAT3DI_TEXT) - 1;
    return FS_MAT3DI_TEXT;
}
const double *fshost_matrix_background(const fshost_matrix *m) { return m ? m->m.pBack.data() : nullptr; }
void fshost_matrix_encode(const fshost_matrix *m, const char *ascii, int len, uint8_t *codes) {
    for (int i = 0; i < len; i++) codes[i] = m->m.aa2num[(unsigned char) ascii[i]];
}
char fshost_matrix_letter(const fshost_matrix *m, int code) { return (code >= 0 && code < m->m.n) ? m->m.letters[code] : 'X'; }
void fshost_comp_bias(const fshost_matrix *m, const uint8_t *seq, int L, float scale, float *out) { fsh::compBias(m->m, seq, L, scale, out); }
void fshost_round_bias(const float *cb, int L, int8_t *out) { fsh::roundBias(cb, L, out); }
int fshost_prefilter_profile(const fshost_matrix *m3di, const uint8_t *q3di, int L, int compBias, float scale, int8_t *pssm, int *scoreCap) {
    if (!m3di || !q3di || L <= 0 || !pssm || !scoreCap) return FSGPU_E_ARG;
    return fsh::prefilterProfile(m3di->m, q3di, L, compBias != 0, scale, pssm, scoreCap);
}
int fshost_align_profiles(const fshost_matrix *mAA, const fshost_matrix *m3Di, const uint8_t *qAA, const uint8_t *q3Di, int L,
                          int compBias, float scale3Di, int16_t *pAA, int16_t *p3Di, int8_t *cbAA, int8_t *cbSS) {
    if (!mAA || !m3Di || !qAA || !q3Di || L <= 0 || !p3Di) return FSGPU_E_ARG;
    return fsh::alignProfiles(mAA->m, m3Di->m, qAA, q3Di, L, compBias != 0, scale3Di, pAA, p3Di, cbAA, cbSS);
}

// banded_sw + computerBacktrace alone (host only): rectangle [qStart, qEnd] x [dbStart, dbEnd] of a pair given by codes; for callers
// that already know the start cell, and for tests without a GPU
int fshost_banded_backtrace(const fshost_matrix *mAA, const fshost_matrix *m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA,
                            const int8_t *cbSS, const uint8_t *tAA, const uint8_t *t3Di, int qStart, int qEnd, int dbStart, int dbEnd, int score,
                            int gapOpen, int gapExtend, unsigned int *identicalAA, char *backtrace, size_t btCap) {
    if (!mAA || !m3Di || !qAA || !q3Di || !cbAA || !cbSS || !tAA || !t3Di || !backtrace || qStart < 0 || dbStart < 0 || qEnd < qStart || dbEnd < dbStart) return FSGPU_E_ARG;
    std::string path;
    if (!fsh::bandedBacktrace(mAA->m, m3Di->m, qAA + qStart, q3Di + qStart, cbAA + qStart, cbSS + qStart,
                         qEnd - qStart + 1, tAA + dbStart, t3Di + dbStart, dbEnd - dbStart + 1, score, gapOpen, gapExtend, path))
        return 0;
    unsigned int ids = 0;
    int qp = qStart, tp = dbStart;
    for (char c : path) {
        if (c == 'M') { ids += tAA[tp] == qAA[qp]; qp++; tp++; }
        else if (c == 'I') qp++;
        else tp++;
    }
    if (identicalAA) *identicalAA = ids;
    if (path.size() + 1 > btCap) return FSGPU_E_ARG;
    memcpy(backtrace, path.c_str(), path.size() + 1);
    return 1;
}

int fshost_block_backtrace(const fshost_matrix *mAA, const fshost_matrix *m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA,
                           const int8_t *cbSS, int Lq, const uint8_t *tAA, const uint8_t *t3Di, int Lt, int qEnd, int dbEnd, int score,
                           int gapOpen, int gapExtend, int *qStart, int *dbStart, unsigned int *identicalAA, char *backtrace, size_t btCap) {
    fsh::BlockAlnOut o;
    fsh::blockBacktrace(mAA->m, m3Di->m, qAA, q3Di, cbAA, cbSS, Lq, tAA, t3Di, Lt, qEnd, dbEnd, score, gapOpen, gapExtend, o);
    if (qStart) *qStart = o.qStart;
    if (dbStart) *dbStart = o.dbStart;
    if (identicalAA) *identicalAA = o.identicalAA;
    if (backtrace && btCap) {
        size_t n = std::min(btCap - 1, o.backtrace.size());
        memcpy(backtrace, o.backtrace.data(), n);
        backtrace[n] = 0;
    }
    return o.ok ? 1 : 0;
}

} // extern "C"

// ---- k-mer prefilter query side -------------------------------------------------------------------------------
extern "C" int fshost_kmer_threshold(float sensitivity, int kmerSize) {
    // Prefiltering::getKmerThreshold, non-profile branch; Foldseek overrides k = 7 (FoldseekBase.cpp:585)
    float best;
    if (kmerSize == 5) best = 160.75f - (sensitivity * 12.75f);
    else if (kmerSize == 6) best = 163.2f - (sensitivity * 8.917f);
    else if (kmerSize == 7) return (int) (197.0 - (11.22 * sensitivity));
    else return -1;
    return (int) best;
}

extern "C" int fshost_kmer_query_prepare(const fshost_matrix *mKmer, const fshost_matrix *mUngapped, const uint8_t *q3di, int L,
                                         int compBias, float scale, int kmerThrBase, int kmerSize, int spaced,
                                         int16_t *kmerThr, int8_t *profile) {
    static const int s6[10] = {1, 1, 0, 1, 0, 1, 0, 0, 1, 1};
    static const int s7[12] = {1, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1};
    if (!mKmer || !mUngapped || L < 0 || (kmerSize != 6 && kmerSize != 7)) return -1;
    int pos[8], psize = spaced ? (kmerSize == 6 ? 10 : 12) : kmerSize, np = 0;
    for (int i = 0; i < psize; i++) if (!spaced || (kmerSize == 6 ? s6[i] : s7[i])) pos[np++] = i;
    std::vector<uint8_t> q(L);
    for (int i = 0; i < L; i++) { uint8_t c = q3di[i]; c = c >= 32 ? c - 32 : c; q[i] = c > 20 ? 20 : c; }
    std::vector<float> bias(L + 1, 0.0f);
    if (compBias) fshost_comp_bias(mKmer, q.data(), L, scale, bias.data());
    const int n = fshost_matrix_size(mUngapped);
    const int16_t *us = fshost_matrix_scores(mUngapped);
    for (int p = 0; p < L; p++) {
        float c = bias[p];
        c = (c < 0.0) ? c / 4 - 0.5 : c / 4 + 0.5;                 // UngappedAlignment.cpp:399-403
        const char corr = (char) c;
        for (int a = 0; a < 21; a++) profile[p * 21 + a] = (int8_t) (us[q[p] * n + a] + corr);
    }
    const int nPos = L - psize + 1;
    for (int i = 0; i < nPos; i++) {
        float bc = 0;
        for (int z = 0; z < kmerSize; z++) bc += bias[i + pos[z]];
        const short b = (short) ((bc < 0.0) ? bc - 0.5 : bc + 0.5);   // QueryMatcher.cpp:267-268
        const int t = kmerThrBase - b;
        kmerThr[i] = (int16_t) (t > 0 ? t : 0);
    }
    return nPos > 0 ? nPos : 0;
}

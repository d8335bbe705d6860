// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (part of oracle/_ref).
//
// Thin extern "C" driver around the REFERENCE's own classes, compiled from the sources where they lie
// under /root/reference by oracle/Makefile (nothing is copied into this repository):
//   SmithWaterman::ssw_init / ungapped_alignment      M/src/alignment/StripedSmithWaterman.cpp:1364,1817
//   StructureSmithWaterman::ssw_init / alignScoreEndPos / alignStartPosBacktraceBlock
//                                                     F/src/commons/StructureSmithWaterman.cpp:1556,263,369
//   SubstitutionMatrix, Sequence, EvalueNeuralNet     M/src/commons, F/src/strucclustutils
// The driver feeds numeric sequences and returns raw results so that tests can pin oracle/fs_oracle.c
// (and through it the HIP path) to what the reference computes, and bench.py can time the reference's
// AVX2 loops on the host cores (cpu_baseline.kind == "reference").
//
// The per-target control flow below mirrors runFilterOnCpu (M/src/prefiltering/ungappedprefilter.cpp:346-482)
// and alignStructure (F/src/strucclustutils/structurealign.cpp:37-112); those two live in module files
// that drag in the whole DB/CLI layer, so only their arithmetic-free glue is re-expressed here.
#include "StripedSmithWaterman.h"
#include "StructureSmithWaterman.h"
#include "SubstitutionMatrix.h"
#include "Sequence.h"
#include "Parameters.h"
#include "EvalueNeuralNet.h"
#include "ref_resources.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

// ---- symbols the linked reference objects need but the hot path never executes ------------------------
#include "ProfileStates.h"
ProfileStates::ProfileStates(int, double *) { abort(); }
ProfileStates::~ProfileStates() {}
namespace Sls { double AlignmentEvaluer::area(double, double, double) const { abort(); } }

namespace {
struct Mats {
    std::string s3di, sblosum;
    Mats() {
        s3di = std::string("mat3di.out:") + std::string((const char *) ref_mat3di_out, ref_mat3di_out_len);
        sblosum = std::string("blosum62.out:") + std::string((const char *) ref_blosum62_out, ref_blosum62_out_len);
    }
};
Mats &mats() { static Mats m; return m; }
const char *matText(int which) { return which == 0 ? mats().s3di.c_str() : mats().sblosum.c_str(); }

std::string toAscii(const BaseMatrix &m, const uint8_t *codes, int L) {
    std::string s(L, 'X');
    for (int i = 0; i < L; i++) s[i] = m.num2aa[codes[i]];
    return s;
}
} // namespace

extern "C" {

int ref_submat(int which, float bitFactor, float scoreBias, int16_t *sub, double *pback) {
    SubstitutionMatrix m(matText(which), bitFactor, scoreBias);
    int n = m.alphabetSize;
    for (int i = 0; i < n; i++) {
        pback[i] = m.pBack[i];
        for (int j = 0; j < n; j++) sub[i * n + j] = m.subMatrix[i][j];
    }
    return n;
}

void ref_comp_bias(int which, float bitFactor, float scoreBias, const uint8_t *seq, int L, float scale, float *out) {
    SubstitutionMatrix m(matText(which), bitFactor, scoreBias);
    SubstitutionMatrix::calcLocalAaBiasCorrection(&m, seq, L, out, scale);
}

// runFilterOnCpu inner loop: scores of one query against n targets (numeric codes, masked >= 32 allowed).
// threads > 1 uses the same "omp for schedule(static)" over targets as the reference.
double ref_ungapped(const uint8_t *q, int Lq, int compBias, float compBiasScale,
                    const uint8_t *tcat, const int64_t *toff, const int32_t *tlen, int64_t n,
                    int threads, int32_t *scores) {
    SubstitutionMatrix subMat(matText(0), 2.0, 0.0);  // ungappedprefilter.cpp:541
    int8_t tiny[32 * 32];
    for (int i = 0; i < subMat.alphabetSize; i++)
        for (int j = 0; j < subMat.alphabetSize; j++) tiny[i * subMat.alphabetSize + j] = subMat.subMatrix[i][j];
    int maxLen = Lq;
    for (int64_t i = 0; i < n; i++) maxLen = std::max(maxLen, (int) tlen[i]);
    maxLen += 2;
    std::string qa = toAscii(subMat, q, Lq);
    double secs = 0;
#pragma omp parallel num_threads(threads)
    {
        Sequence qSeq(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMat, 0, false, compBias);
        SmithWaterman aligner(maxLen, subMat.alphabetSize, compBias, compBiasScale, NULL);
        qSeq.mapSequence(0, 0, qa.c_str(), Lq);
        aligner.ssw_init(&qSeq, tiny, &subMat);
        std::vector<unsigned char> t(maxLen + 1);
        const unsigned char xChar = subMat.aa2num[static_cast<int>('X')];
#pragma omp barrier
        auto t0 = std::chrono::steady_clock::now();
#pragma omp for schedule(static)
        for (int64_t id = 0; id < n; id++) {
            const uint8_t *src = tcat + toff[id];
            const int L = tlen[id];
            // padded-DB codes: >= 32 means soft-masked -> X (ungappedprefilter.cpp:402-405)
            for (int i = 0; i < L; i++) t[i] = (src[i] >= 32) ? xChar : src[i];
            scores[id] = aligner.ungapped_alignment(t.data(), L);
        }
        auto t1 = std::chrono::steady_clock::now();
#pragma omp master
        secs = std::chrono::duration<double>(t1 - t0).count();
    }
    return secs;
}

struct RefSw {
    int32_t score, qEnd, dbEnd, word;
    float qCov, tCov;
};

struct RefAln {
    int32_t fwdScore, revScore, score;   // score = fwd - rev
    int32_t qStart, qEnd, dbStart, dbEnd;
    int32_t status;                      // 0 ok, 1 cov gate, 2 evalue gate (fwd), 3 evalue gate (diff)
    int32_t alnLen, identicalAA;
    float qCov, tCov, seqId;
    double evalue;
};

// One query against n targets: forward + reversed-query alignScoreEndPos exactly as
// structurealign.cpp:343-347 prepares them.  alignmentType 0 -> AA factor 0.0, 2 -> 1.4 (structurealign.cpp:264).
// If aln != NULL, also applies the alignStructure gates (covThr = 0) and, with doBacktrace, the block-aligner
// backtrace; cigars are appended to cigarOut separated by '\n'.
double ref_structure_align(const uint8_t *qAA, const uint8_t *q3Di, int Lq, int alignmentType,
                           int compBias, float compBiasScale, int gapOpen, int gapExtend,
                           const uint8_t *tAAcat, const uint8_t *t3Dicat, const int64_t *toff, const int32_t *tlen,
                           int64_t n, int64_t dbResidues, double evalThr, int doBacktrace, int threads,
                           RefSw *fwdOut, RefSw *revOut, RefAln *aln, char *cigarOut, int64_t cigarCap) {
    SubstitutionMatrix subMat3Di(matText(0), 2.1, 0.0);
    float aaFactor = (alignmentType == 2) ? 1.4 : 0.0;
    SubstitutionMatrix subMatAA(matText(1), aaFactor, 0.0);
    int8_t tinyAA[32 * 32], tiny3Di[32 * 32];
    const int A = subMat3Di.alphabetSize;
    for (int i = 0; i < A; i++)
        for (int j = 0; j < A; j++) {
            tiny3Di[i * A + j] = subMat3Di.subMatrix[i][j];
            tinyAA[i * A + j] = subMatAA.subMatrix[i][j];
        }
    int maxLen = Lq;
    for (int64_t i = 0; i < n; i++) maxLen = std::max(maxLen, (int) tlen[i]);
    maxLen += 2;
    std::string qaa = toAscii(subMatAA, qAA, Lq), q3 = toAscii(subMat3Di, q3Di, Lq);
    double secs = 0;
    std::vector<std::string> cigars(aln != NULL ? n : 0);
#pragma omp parallel num_threads(threads)
    {
        EvalueNeuralNet evaluer(dbResidues, &subMat3Di);
        StructureSmithWaterman fwd(maxLen, A, compBias, compBiasScale, &subMatAA, &subMat3Di);
        StructureSmithWaterman rev(maxLen, A, compBias, compBiasScale, &subMatAA, &subMat3Di);
        Sequence qSeqAA(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMatAA, 0, false, compBias);
        Sequence qSeq3Di(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMat3Di, 0, false, compBias);
        Sequence tSeqAA(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMatAA, 0, false, compBias);
        Sequence tSeq3Di(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMat3Di, 0, false, compBias);
        qSeq3Di.mapSequence(0, 0, q3.c_str(), Lq);
        qSeqAA.mapSequence(0, 0, qaa.c_str(), Lq);
        std::pair<double, double> muLambda = evaluer.predictMuLambda(qSeq3Di.numSequence, qSeq3Di.L);
        fwd.ssw_init(&qSeqAA, &qSeq3Di, tinyAA, tiny3Di, &subMatAA);
        qSeq3Di.reverse();
        qSeqAA.reverse();
        rev.ssw_init(&qSeqAA, &qSeq3Di, tinyAA, tiny3Di, &subMatAA);
        std::string backtrace;
#pragma omp barrier
        auto t0 = std::chrono::steady_clock::now();
#pragma omp for schedule(dynamic, 1)
        for (int64_t id = 0; id < n; id++) {
            const int L = tlen[id];
            std::string taa = toAscii(subMatAA, tAAcat + toff[id], L), t3 = toAscii(subMat3Di, t3Dicat + toff[id], L);
            tSeq3Di.mapSequence(id, id, t3.c_str(), L);
            tSeqAA.mapSequence(id, id, taa.c_str(), L);
            StructureSmithWaterman::s_align a = fwd.alignScoreEndPos<StructureSmithWaterman::PROFILE>(
                tSeqAA.numSequence, tSeq3Di.numSequence, L, gapOpen, gapExtend, Lq / 2);
            if (fwdOut) { RefSw r = {(int32_t) a.score1, a.qEndPos1, a.dbEndPos1, a.word, a.qCov, a.tCov}; fwdOut[id] = r; }
            if (aln == NULL) {
                StructureSmithWaterman::s_align b = rev.alignScoreEndPos<StructureSmithWaterman::PROFILE>(
                    tSeqAA.numSequence, tSeq3Di.numSequence, L, gapOpen, gapExtend, Lq / 2);
                if (revOut) { RefSw r = {(int32_t) b.score1, b.qEndPos1, b.dbEndPos1, b.word, b.qCov, b.tCov}; revOut[id] = r; }
                continue;
            }
            // ---- alignStructure (structurealign.cpp:37-112), covThr = 0.0 / covMode 0 ----
            RefAln &o = aln[id];
            memset(&o, 0, sizeof(o));
            o.fwdScore = a.score1; o.qEnd = a.qEndPos1; o.dbEnd = a.dbEndPos1; o.qStart = -1; o.dbStart = -1;
            if (!Util::hasCoverage(0.0f, 0, a.qCov, a.tCov)) { o.status = 1; continue; }
            a.evalue = evaluer.computeEvalueCorr(a.score1, muLambda.first, muLambda.second);
            o.evalue = a.evalue;
            if (a.evalue > evalThr) { o.status = 2; continue; }
            StructureSmithWaterman::s_align b = rev.alignScoreEndPos<StructureSmithWaterman::PROFILE>(
                tSeqAA.numSequence, tSeq3Di.numSequence, L, gapOpen, gapExtend, Lq / 2);
            if (revOut) { RefSw r = {(int32_t) b.score1, b.qEndPos1, b.dbEndPos1, b.word, b.qCov, b.tCov}; revOut[id] = r; }
            o.revScore = b.score1;
            int32_t score = static_cast<int32_t>(a.score1) - static_cast<int32_t>(b.score1);
            o.score = score;
            a.evalue = evaluer.computeEvalueCorr(score, muLambda.first, muLambda.second);
            o.evalue = a.evalue;
            if (a.evalue > evalThr) { o.status = 3; continue; }
            backtrace.clear();
            float seqId = 0.0;
            if (doBacktrace) {
                StructureSmithWaterman::s_align tmp = fwd.alignStartPosBacktraceBlock(
                    tSeqAA.numSequence, tSeq3Di.numSequence, L, gapOpen, gapExtend, backtrace, a);
                a = tmp;   // structurealign.cpp:83-88: the failure test reads the wrong variable, so tmp is always taken
            }
            // Matcher::computeAlnLength (M/src/alignment/Matcher.cpp:158)
            unsigned int alnLength = std::max(abs(a.qEndPos1 - a.qStartPos1), abs(a.dbEndPos1 - a.dbStartPos1)) + 1;
            if (backtrace.size() > 0) {
                alnLength = backtrace.size();
                // Util::computeSeqId, SEQ_ID_ALN_LEN (M/src/commons/Util.cpp:603)
                seqId = static_cast<float>(a.identicalAACnt) / static_cast<float>(alnLength);
            }
            o.qStart = a.qStartPos1; o.dbStart = a.dbStartPos1; o.qEnd = a.qEndPos1; o.dbEnd = a.dbEndPos1;
            o.qCov = a.qCov; o.tCov = a.tCov; o.alnLen = alnLength; o.seqId = seqId; o.identicalAA = a.identicalAACnt;
            cigars[id] = backtrace;
        }
        auto t1 = std::chrono::steady_clock::now();
#pragma omp master
        secs = std::chrono::duration<double>(t1 - t0).count();
    }
    if (aln != NULL && cigarOut != NULL) {
        int64_t p = 0;
        for (int64_t i = 0; i < n; i++) {
            if (p + (int64_t) cigars[i].size() + 2 > cigarCap) break;
            memcpy(cigarOut + p, cigars[i].data(), cigars[i].size());
            p += cigars[i].size();
            cigarOut[p++] = '\n';
        }
        cigarOut[p] = '\0';
    }
    return secs;
}

// alignStartPosBacktrace<PROFILE> of the reference (F/src/commons/StructureSmithWaterman.cpp:540-739: reverse striped pass, banded_sw,
// computerBacktrace) for one query against n targets, the way structurealign would call it for a sequence query if its fall-back
// fired (structurealign.cpp:91-100: alignmentMode 3, covMode 0 / covThr 0, maskLen = Lq / 2).  out[id] = {qStart, dbStart, identicalAA,
// status (0 ok, 1 forward score 0)}; cigars appended to cigarOut separated by '\n'.
void ref_structure_startpos(const uint8_t *qAA, const uint8_t *q3Di, int Lq, int alignmentType, int compBias, float compBiasScale,
                            int gapOpen, int gapExtend, const uint8_t *tAAcat, const uint8_t *t3Dicat, const int64_t *toff,
                            const int32_t *tlen, int64_t n, int32_t *out4, char *cigarOut, int64_t cigarCap) {
    SubstitutionMatrix subMat3Di(matText(0), 2.1, 0.0);
    float aaFactor = (alignmentType == 2) ? 1.4 : 0.0;
    SubstitutionMatrix subMatAA(matText(1), aaFactor, 0.0);
    int8_t tinyAA[32 * 32], tiny3Di[32 * 32];
    const int A = subMat3Di.alphabetSize;
    for (int i = 0; i < A; i++)
        for (int j = 0; j < A; j++) { tiny3Di[i * A + j] = subMat3Di.subMatrix[i][j]; tinyAA[i * A + j] = subMatAA.subMatrix[i][j]; }
    int maxLen = Lq;
    for (int64_t i = 0; i < n; i++) maxLen = std::max(maxLen, (int) tlen[i]);
    maxLen += 2;
    std::string qaa = toAscii(subMatAA, qAA, Lq), q3 = toAscii(subMat3Di, q3Di, Lq);
    StructureSmithWaterman fwd(maxLen, A, compBias, compBiasScale, &subMatAA, &subMat3Di);
    Sequence qSeqAA(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMatAA, 0, false, compBias);
    Sequence qSeq3Di(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMat3Di, 0, false, compBias);
    Sequence tSeqAA(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMatAA, 0, false, compBias);
    Sequence tSeq3Di(maxLen, Parameters::DBTYPE_AMINO_ACIDS, &subMat3Di, 0, false, compBias);
    qSeq3Di.mapSequence(0, 0, q3.c_str(), Lq);
    qSeqAA.mapSequence(0, 0, qaa.c_str(), Lq);
    fwd.ssw_init(&qSeqAA, &qSeq3Di, tinyAA, tiny3Di, &subMatAA);
    std::string backtrace;
    int64_t p = 0;
    for (int64_t id = 0; id < n; id++) {
        const int L = tlen[id];
        std::string taa = toAscii(subMatAA, tAAcat + toff[id], L), t3 = toAscii(subMat3Di, t3Dicat + toff[id], L);
        tSeq3Di.mapSequence(id, id, t3.c_str(), L);
        tSeqAA.mapSequence(id, id, taa.c_str(), L);
        StructureSmithWaterman::s_align a = fwd.alignScoreEndPos<StructureSmithWaterman::PROFILE>(tSeqAA.numSequence, tSeq3Di.numSequence, L, gapOpen, gapExtend, Lq / 2);
        int32_t *o = out4 + id * 4;
        o[0] = o[1] = -1; o[2] = 0; o[3] = 1;
        backtrace.clear();
        if (a.score1 > 0) {
            StructureSmithWaterman::s_align b = fwd.alignStartPosBacktrace<StructureSmithWaterman::PROFILE>(
                tSeqAA.numSequence, tSeq3Di.numSequence, L, gapOpen, gapExtend, 3, backtrace, a, 0, 0.0f, Lq / 2);
            o[0] = b.qStartPos1; o[1] = b.dbStartPos1; o[2] = (int32_t) b.identicalAACnt; o[3] = 0;
        }
        if (cigarOut && p + (int64_t) backtrace.size() + 2 <= cigarCap) {
            memcpy(cigarOut + p, backtrace.data(), backtrace.size());
            p += backtrace.size();
            cigarOut[p++] = '\n';
        }
    }
    if (cigarOut) cigarOut[p] = '\0';
}

void ref_mu_lambda(const uint8_t *q3Di, int L, int64_t dbResidues, double *lambda, double *mu) {
    SubstitutionMatrix subMat3Di(matText(0), 2.1, 0.0);
    EvalueNeuralNet evaluer(dbResidues, &subMat3Di);
    std::vector<unsigned char> s(q3Di, q3Di + L);
    std::pair<double, double> ml = evaluer.predictMuLambda(s.data(), L);
    *lambda = ml.first;
    *mu = ml.second;
}

double ref_evalue_corr(double score, double lambda, double mu, int64_t dbResidues) {
    SubstitutionMatrix subMat3Di(matText(0), 2.1, 0.0);
    EvalueNeuralNet evaluer(dbResidues, &subMat3Di);
    return evaluer.computeEvalueCorr(score, lambda, mu);
}

int ref_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"

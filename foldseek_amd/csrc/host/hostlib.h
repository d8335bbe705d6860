// hostlib.h -- internal C++ declarations of the host-side path (see include/fshost.h for the C ABI and citations).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "fshost.h"

namespace fsh {

struct Matrix {
    int n = 0;
    std::string letters;              // num2aa
    std::vector<short> sub;           // [n*n]
    std::vector<int8_t> tiny;         // same as int8
    std::vector<double> pBack;        // [n]
    uint8_t aa2num[256];
    bool build(const std::vector<double> &score, const std::vector<double> &back, double lambda, const std::string &letters,
               float bitFactor, float scoreBias);
    bool parse(const char *text, float bitFactor, float scoreBias);
    bool builtin(int which, float bitFactor, float scoreBias);
};

void compBias(const Matrix &m, const uint8_t *seq, int N, float scale, float *out);
void roundBias(const float *cb, int L, int8_t *out);
int prefilterProfile(const Matrix &m, const uint8_t *q, int L, bool compBiasOn, float scale, int8_t *pssm, int *scoreCap);
int alignProfiles(const Matrix &mAA, const Matrix &m3Di, const uint8_t *qAA, const uint8_t *q3Di, int L, bool compBiasOn,
                  float scale3Di, int16_t *pAA, int16_t *p3Di, int8_t *cbAA, int8_t *cbSS);

struct Evaluer {
    struct Layer { uint32_t rows, cols, nbias, act; std::vector<float> w, b; };
    std::vector<Layer> layers;
    double logDbResidueCount = 0;
    bool load(const char *path, uint64_t dbResidues, std::string &err);
    void predictMuLambda(const uint8_t *seq, unsigned int L, int alphabetSize, double *lambda, double *mu) const;
    double computeEvalueCorr(double score, double lambda, double mu) const;
};

std::string libraryDir();   // directory that holds libfsgpu.so

// block-aligner based start position + backtrace (alignStartPosBacktraceBlock); returns false when the X-drop
// aligner does not reach the SW score (the reference then leaves start positions at -1 and the backtrace empty).
struct BlockAlnOut { int qStart = -1, dbStart = -1; unsigned int identicalAA = 0; std::string backtrace; bool ok = false; };
void blockBacktrace(const Matrix &mAA, const Matrix &m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA,
                    const int8_t *cbSS, int Lq, const uint8_t *tAA, const uint8_t *t3Di, int Lt, int qEnd, int dbEnd,
                    int targetScore, int gapOpen, int gapExtend, BlockAlnOut &out);

// banded_sw with band doubling + trace-back (StructureSmithWaterman.cpp:1723-1957) over the rectangle that starts at the
// given pointers; path = M / I / D string from the start cell to the end cell.  false: impossible direction code.
bool bandedBacktrace(const Matrix &mAA, const Matrix &m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA, const int8_t *cbSS,
                     int qLen, const uint8_t *tAA, const uint8_t *t3Di, int dbLen, int score, int gapOpen, int gapExtend, std::string &path);

} // namespace fsh

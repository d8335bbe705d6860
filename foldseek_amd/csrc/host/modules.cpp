// modules.cpp -- the three sub-commands of the hot path with the reference's module contract
// ("int f(int argc, const char **argv)", DBs in -> DBs out, EXIT_SUCCESS / message + EXIT_FAILURE):
//   ungappedprefilter <queryDB_ss> <targetDB_ss[_pad]> <outPrefDB>          M/src/prefiltering/ungappedprefilter.cpp:484-595
//   structurealign    <queryDB> <targetDB[_pad]> <prefDB> <outAlnDB>         F/src/strucclustutils/structurealign.cpp:141-481
//   prefilter         <queryDB_ss> <targetDB_ss[_pad]> <outPrefDB>          M/src/prefiltering/Main.cpp -> Prefiltering.cpp:22-245,755-982
//   makepaddedseqdb   <seqDB> <outPaddedDB>                                  M/src/util/makepaddedseqdb.cpp:14-154
//   search            <queryDB> <targetDB[_pad]> <outAlnDB> [<outPrefDB>]   prefilter + structurealign of
//                     F/data/structuresearch.sh:41-53,116-143 in ONE process: the target DB and (for the k-mer mode) its
//                     index stay resident, no prefilter DB round trip; same result DBs
//   gpuserver         <targetDB_ss[_pad]>                                    M/src/util/gpuserver.cpp:24-101 (resident DB, shm protocol);
//                     client side: ungappedprefilter --gpu-server 1           M/src/prefiltering/ungappedprefilter.cpp:71-122,208-257
// They read and write the same on-disk databases as the reference modules, so the shell workflows
// (F/data/structuresearch.sh:41-53,116-143) can call them in place of the originals.  All DP work is done by the
// device library (fsgpu_*), this file is DB plumbing + option parsing.
#include "hostlib.h"
#include "mmseqs_db.h"
#include "gpu_shm.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <regex.h>
#include <thread>
#include <vector>

using namespace fsh;

namespace {

struct Options {
    std::vector<std::string> pos;
    std::map<std::string, std::string> kv;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    // MultiParam values ("aa:10,nucl:10", "seq:80,prof:80") carry the amino-acid / sequence component first
    static std::string first(const std::string &v) {
        const size_t c = v.find(':');
        if (c == std::string::npos) return v;
        const size_t e = v.find(',', c);
        return v.substr(c + 1, e == std::string::npos ? std::string::npos : e - c - 1);
    }
    std::string gets(const std::string &k, const std::string &d) const { auto it = kv.find(k); return it == kv.end() ? d : first(it->second); }
    int geti(const std::string &k, int d) const { auto it = kv.find(k); return it == kv.end() ? d : atoi(first(it->second).c_str()); }
    double getd(const std::string &k, double d) const { auto it = kv.find(k); return it == kv.end() ? d : atof(first(it->second).c_str()); }
};

// ---- option tables -----------------------------------------------------------------------------------------------
// The workflows hand every module its complete parameter string (Parameters::createParameterString, e.g.
// F/src/workflow/StructureSearch.cpp:95-121), so a drop-in has to know every flag of the module it replaces
// (M/src/commons/Parameters.cpp:415-447,508-524,913-921,1638-1640; F/src/commons/LocalParameters.cpp:161-167) and say
// what it does with it:
//   USE     implemented, the value is honoured
//   IGNORE  has no effect on this path in the reference either (profile pseudo counts, nucleotide-only options, I/O
//           tuning) -- accepted with any value
//   ONLY    changes results in the reference; this implementation covers the listed values only and REFUSES anything
//           else (EXIT_FAILURE, like the reference's "Error in argument"), never silently computes something different
// Unknown flags are refused with the reference's message (Parameters.cpp:2087).
enum FlagSupport { USE, IGNORE, ONLY };
struct FlagSpec {
    const char *name;
    bool isBool;            // typeid(bool): the value is optional, a bare flag toggles the default (Parameters.cpp:2052-2062)
    FlagSupport support;
    const char *only;       // ONLY: '|'-separated accepted values (first MultiParam component); bool defaults for isBool
};

// flags every module shares (Parameters.cpp COMMAND_COMMON) + the device selection flags of this implementation
const FlagSpec kCommonFlags[] = {
    {"--threads", false, USE, nullptr}, {"-v", false, IGNORE, nullptr}, {"--compressed", false, ONLY, "0"},
    {"--db-load-mode", false, IGNORE, nullptr}, {"--sub-mat", false, ONLY, "3di.out"},
    {"--gpus", false, USE, nullptr}, {"--gpu-device", false, USE, nullptr}, {"--gpu", false, IGNORE, nullptr},
    {nullptr, false, USE, nullptr}};

const FlagSpec kPrefilterFlags[] = {           // Parameters::prefilter (Parameters.cpp:350-395)
    {"--seed-sub-mat", false, ONLY, "3di.out"}, {"-s", false, USE, nullptr}, {"-k", false, ONLY, "0|6"},
    {"--target-search-mode", false, ONLY, "0"}, {"--k-score", false, USE, nullptr}, {"--alph-size", false, ONLY, "21"},
    {"--max-seq-len", false, USE, nullptr}, {"--max-seqs", false, USE, nullptr}, {"--split", false, ONLY, "0|1"},
    {"--split-mode", false, IGNORE, nullptr}, {"--split-memory-limit", false, IGNORE, nullptr},
    {"--disk-space-limit", false, IGNORE, nullptr}, {"-c", false, USE, nullptr}, {"--cov-mode", false, USE, nullptr},
    {"--comp-bias-corr", false, USE, nullptr}, {"--comp-bias-corr-scale", false, USE, nullptr},
    {"--diag-score", true, USE, "1"}, {"--exact-kmer-matching", false, ONLY, "0"}, {"--mask", false, ONLY, "0"},
    {"--mask-prob", false, IGNORE, nullptr}, {"--mask-lower-case", false, USE, nullptr}, {"--mask-n-repeat", false, USE, nullptr},
    {"--min-ungapped-score", false, USE, nullptr}, {"--add-self-matches", true, USE, "0"}, {"--spaced-kmer-mode", false, USE, nullptr},
    {"--spaced-kmer-pattern", false, ONLY, ""}, {"--local-tmp", false, IGNORE, nullptr}, {"--pca", false, IGNORE, nullptr},
    {"--pcb", false, IGNORE, nullptr}, {"--taxon-list", false, ONLY, ""}, {"-e", false, IGNORE, nullptr},
    {nullptr, false, USE, nullptr}};

const FlagSpec kUngappedFlags[] = {            // Parameters::ungappedprefilter (Parameters.cpp:508-524)
    {"-c", false, IGNORE, nullptr}, {"-e", false, IGNORE, nullptr}, {"--cov-mode", false, IGNORE, nullptr},   // read in prefilter-mode 2 only
    {"--comp-bias-corr", false, USE, nullptr}, {"--comp-bias-corr-scale", false, USE, nullptr},
    {"--min-ungapped-score", false, USE, nullptr}, {"--max-seqs", false, USE, nullptr}, {"--taxon-list", false, ONLY, ""},
    {"--gpu-server", false, USE, nullptr}, {"--gpu-server-wait-timeout", false, USE, nullptr},
    {"--prefilter-mode", false, ONLY, "0|1"},  // 2 = ungapped + gapped (Marv GAPLESS_SMITH_WATERMAN) is not on this path
    {"--shm-name", false, USE, nullptr}, {"--gpu-server-version", false, USE, nullptr},
    {nullptr, false, USE, nullptr}};

const FlagSpec kAlignFlags[] = {               // LocalParameters::structurealign = structurealign + Parameters::align
    {"--tmscore-threshold", false, ONLY, "0|0.0|0.000"}, {"--tmscore-threshold-mode", false, IGNORE, nullptr},
    {"--lddt-threshold", false, ONLY, "0|0.0|0.000"},
    {"--alignment-type", false, ONLY, "0|2"}, {"--exact-tmscore", false, IGNORE, nullptr},
    {"-a", true, USE, "0"}, {"--add-backtrace", true, USE, "0"}, {"--alignment-mode", false, ONLY, "0|3"},
    {"--alignment-output-mode", false, ONLY, "0"}, {"--wrapped-scoring", true, ONLY, "0"}, {"-e", false, USE, nullptr},
    {"--min-seq-id", false, USE, nullptr}, {"--min-aln-len", false, USE, nullptr}, {"--seq-id-mode", false, USE, nullptr},
    {"--alt-ali", false, USE, nullptr}, {"-c", false, USE, nullptr}, {"--cov-mode", false, USE, nullptr},
    {"--max-seq-len", false, USE, nullptr}, {"--comp-bias-corr", false, USE, nullptr}, {"--comp-bias-corr-scale", false, USE, nullptr},
    {"--max-rejected", false, USE, nullptr}, {"--max-accept", false, USE, nullptr}, {"--add-self-matches", true, USE, "0"},
    {"--pca", false, IGNORE, nullptr}, {"--pcb", false, IGNORE, nullptr}, {"--score-bias", false, ONLY, "0|0.0|0.000"},
    {"--realign", true, ONLY, "0"}, {"--realign-score-bias", false, IGNORE, nullptr}, {"--realign-max-seqs", false, IGNORE, nullptr},
    {"--corr-score-weight", false, IGNORE, nullptr}, {"--gap-open", false, USE, nullptr}, {"--gap-extend", false, USE, nullptr},
    {"--zdrop", false, IGNORE, nullptr},
    {nullptr, false, USE, nullptr}};
const FlagSpec kStructAlignOnlyFlags[] = {     // structurealign, not structurerescorediagonal (LocalParameters.cpp:154-167)
    {"--sort-by-structure-bits", false, USE, nullptr}, {"--align-batch", false, USE, nullptr}, {nullptr, false, USE, nullptr}};
const FlagSpec kRescoreOnlyFlags[] = {         // ours: what to do with pairs whose reference result is undefined (see fsgpu.h)
    {"--undefined-diagonals", false, ONLY, "fail|skip"}, {nullptr, false, USE, nullptr}};

const FlagSpec kSearchFlags[] = {              // the fused module: prefilter / ungappedprefilter + structurealign
    {"--prefilter-mode", false, ONLY, "0|1"}, {nullptr, false, USE, nullptr}};

const FlagSpec kPaddedFlags[] = {              // Parameters::makepaddedseqdb (Parameters.cpp:913-921)
    {"--score-bias", false, IGNORE, nullptr}, {"--mask", false, ONLY, "0"}, {"--mask-prob", false, IGNORE, nullptr},
    {"--mask-lower-case", false, IGNORE, nullptr}, {"--mask-n-repeat", false, IGNORE, nullptr}, {"--write-lookup", false, USE, nullptr},
    {nullptr, false, USE, nullptr}};

const FlagSpec kConvertFlags[] = {             // LocalParameters::convertalignments (Parameters.cpp:661-672, LocalParameters.cpp:117)
    {"--format-mode", false, ONLY, "0|2|4"},   // 1 SAM and 3 HTML end in "Not implemented yet" / need the C-alpha DB in the reference too; 5 superposed PDB needs it
    {"--format-output", false, USE, nullptr}, {"--translation-table", false, IGNORE, nullptr}, {"--gap-open", false, IGNORE, nullptr},
    {"--gap-extend", false, IGNORE, nullptr}, {"--db-output", true, USE, "0"}, {"--search-type", false, IGNORE, nullptr},
    {"--exact-tmscore", false, IGNORE, nullptr}, {nullptr, false, USE, nullptr}};

const FlagSpec kIndexdbFlags[] = {             // Parameters::indexdb (Parameters.cpp:850-872)
    {"--seed-sub-mat", false, ONLY, "3di.out"}, {"-k", false, ONLY, "0|6"}, {"--alph-size", false, ONLY, "21"},
    {"--comp-bias-corr", false, USE, nullptr}, {"--comp-bias-corr-scale", false, IGNORE, nullptr}, {"--max-seq-len", false, USE, nullptr},
    {"--max-seqs", false, IGNORE, nullptr}, {"--index-dbsuffix", false, USE, nullptr}, {"--mask", false, ONLY, "0"},
    {"--mask-prob", false, IGNORE, nullptr}, {"--mask-lower-case", false, USE, nullptr}, {"--mask-n-repeat", false, USE, nullptr},
    {"--spaced-kmer-mode", false, USE, nullptr}, {"--spaced-kmer-pattern", false, ONLY, ""}, {"-s", false, USE, nullptr},
    {"--k-score", false, USE, nullptr}, {"--check-compatible", false, ONLY, "0"}, {"--search-type", false, IGNORE, nullptr},
    {"--split", false, ONLY, "0|1"}, {"--split-memory-limit", false, IGNORE, nullptr}, {"--index-subset", false, USE, nullptr},
    {nullptr, false, USE, nullptr}};

const FlagSpec kServerFlags[] = {              // Parameters::gpuserver (Parameters.cpp:1638-1640)
    {"--max-seqs", false, USE, nullptr}, {"--prefilter-mode", false, ONLY, "0|1"}, {"--max-seq-len", false, USE, nullptr},
    {"--shm-name", false, USE, nullptr}, {"--gpu-server-version", false, USE, nullptr},
    {nullptr, false, USE, nullptr}};

// FSGPU_MODULE_TIMING=1: phase times of a module run on stderr (wall clock of the calling thread; the per-thread phases are summed
// over the host threads, so they can exceed the loop's wall time)
static double nowSec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool moduleTiming() { const char *e = getenv("FSGPU_MODULE_TIMING"); return e && *e && *e != '0'; }

// Value domains of the reference's parameters: the `regex` argument of their definitions (M/src/commons/Parameters.cpp:28-330,
// F/src/commons/LocalParameters.cpp:9-130 with the overrides at :56,:78,:461), applied the way Parameters::parseParameters does
// (Parameters.cpp:1820-1827,1905-2040: POSIX extended, REG_NEWLINE, regexec = a SEARCH, so an unanchored pattern accepts what it accepts
// there).  A value outside its domain ends in the reference's "Error in argument <flag>" before anything is opened.
//   V_REGEX       int / size_t / float / double parameters
//   V_BYTES       ByteParser parameters: "Error in argument regex <flag>"
//   V_NUCLAA_INT  MultiParam<NuclAA<int>>: "N" or "aa:N,nucl:M"; a component with trailing junk or out of range is "Error in value
//                 parsing <flag>", a purely non-numeric one reads as 0 (MultiParam.h:210-219) -- as there
enum ValueKind { V_REGEX, V_BYTES, V_NUCLAA_INT };
struct FlagDomain { const char *name; ValueKind kind; const char *regex; };
const FlagDomain kFlagDomains[] = {
    {"--alignment-mode", V_REGEX, "^[0-3]{1}$"}, {"--alignment-output-mode", V_REGEX, "^[0-1]{1}$"}, {"--alignment-type", V_REGEX, "^[0-3]{1}$"},
    {"--alph-size", V_NUCLAA_INT, nullptr}, {"--alt-ali", V_REGEX, "^[0-9]{1}[0-9]*$"}, {"--check-compatible", V_REGEX, "^[0-2]{1}$"},
    {"--comp-bias-corr", V_REGEX, "^[0-1]{1}$"}, {"--comp-bias-corr-scale", V_REGEX, "^0(\\.[0-9]+)?|^1(\\.0+)?$"},
    {"--compressed", V_REGEX, "^[0-1]{1}$"}, {"--corr-score-weight", V_REGEX, "^-?[0-9]*(\\.[0-9]+)?$"}, {"--cov-mode", V_REGEX, "^[0-5]{1}$"},
    {"--db-load-mode", V_REGEX, "[0-3]{1}"}, {"--disk-space-limit", V_BYTES, "^(0|[1-9]{1}[0-9]*(B|K|M|G|T)?)$"},
    {"--exact-kmer-matching", V_REGEX, "^[0-1]{1}$"}, {"--exact-tmscore", V_REGEX, "^[0-1]{1}$"}, {"--format-mode", V_REGEX, "^[0-5]{1}$"},
    {"--gap-extend", V_NUCLAA_INT, nullptr}, {"--gap-open", V_NUCLAA_INT, nullptr}, {"--gpu", V_REGEX, "^[0-1]{1}$"},
    {"--gpu-server", V_REGEX, "^[0-1]{1}$"}, {"--gpu-server-wait-timeout", V_REGEX, "^-?[0-9]+"}, {"--index-subset", V_REGEX, "^[0-9]{1}[0-9]*$"},
    {"--lddt-threshold", V_REGEX, "^0(\\.[0-9]+)?|1(\\.0+)?$"}, {"--mask", V_REGEX, "^[0-1]{1}"}, {"--mask-lower-case", V_REGEX, "^[0-1]{1}"},
    {"--mask-n-repeat", V_REGEX, "^[0-9]{1}[0-9]*$"}, {"--mask-prob", V_REGEX, "^0(\\.[0-9]+)?|^1(\\.0+)?$"},
    {"--max-accept", V_REGEX, "^[1-9]{1}[0-9]*$"}, {"--max-rejected", V_REGEX, "^[1-9]{1}[0-9]*$"}, {"--max-seq-len", V_REGEX, "^[0-9]{1}[0-9]*"},
    {"--max-seqs", V_REGEX, "^[1-9]{1}[0-9]*$"}, {"--min-aln-len", V_REGEX, "^[0-9]{1}[0-9]*$"},
    {"--min-seq-id", V_REGEX, "^0(\\.[0-9]+)?|1(\\.0+)?$"}, {"--min-ungapped-score", V_REGEX, "^[0-9]{1}[0-9]*$"},
    {"--prefilter-mode", V_REGEX, "^[0-3]{1}$"}, {"--realign-max-seqs", V_REGEX, "^[0-9]{1}[0-9]*$"},
    {"--realign-score-bias", V_REGEX, "^-?[0-9]*(\\.[0-9]+)?$"}, {"--score-bias", V_REGEX, "^-?[0-9]*(\\.[0-9]+)?$"},
    {"--search-type", V_REGEX, "^[0-4]{1}"}, {"--seq-id-mode", V_REGEX, "^[0-2]{1}$"}, {"--sort-by-structure-bits", V_REGEX, "^[0-1]{1}$"},
    {"--spaced-kmer-mode", V_REGEX, "^[0-1]{1}"}, {"--split", V_REGEX, "^[0-9]{1}[0-9]*$"},
    {"--split-memory-limit", V_BYTES, "^(0|[1-9]{1}[0-9]*(B|K|M|G|T)?)$"}, {"--split-mode", V_REGEX, "^[0-2]{1}$"},
    {"--target-search-mode", V_REGEX, "^[0-1]{1}$"}, {"--threads", V_REGEX, "^[1-9]{1}[0-9]*$"},
    {"--tmscore-threshold", V_REGEX, "^0(\\.[0-9]+)?|1(\\.0+)?$"}, {"--tmscore-threshold-mode", V_REGEX, "^[0-2]{1}$"},
    {"--translation-table", V_REGEX, "^[1-9]{1}[0-9]*$"}, {"--write-lookup", V_REGEX, "^[0-1]{1}"}, {"-c", V_REGEX, "^0(\\.[0-9]+)?|^1(\\.0+)?$"},
    {"-e", V_REGEX, "^([-+]?[0-9]*\\.?[0-9]+([eE][-+]?[0-9]+)?)|[0-9]*(\\.[0-9]+)?$"}, {"-k", V_REGEX, "^[0-9]{1}[0-9]*$"},
    {"-s", V_REGEX, "^[0-9]*(\\.[0-9]+)?$"}, {"-v", V_REGEX, "^[0-4]{1}$"},
    {nullptr, V_REGEX, nullptr}};

// "" = inside the domain, else the reference's message
std::string domainError(const std::string &flag, const std::string &value) {
    const FlagDomain *d = kFlagDomains;
    while (d->name && flag != d->name) d++;
    if (!d->name) return "";
    if (d->kind == V_NUCLAA_INT) {
        auto component = [](const std::string &v) {          // MultiParam::assign(const std::string &, int &)
            char *rest = nullptr;
            errno = 0;
            (void) strtol(v.c_str(), &rest, 10);
            return !((rest != v.c_str() && *rest != '\0') || errno == ERANGE);
        };
        bool ok;
        if (value.find(',') == std::string::npos) ok = component(value);
        else {
            // two components, "aa:" and "nucl:" in either order, each with exactly one ':' (MultiParam.cpp:15-33)
            const size_t comma = value.find(',');
            const std::string parts[2] = {value.substr(0, comma), value.substr(comma + 1)};
            bool haveAA = false, haveNucl = false;
            ok = parts[1].find(',') == std::string::npos && !parts[0].empty() && !parts[1].empty();
            for (int i = 0; ok && i < 2; i++) {
                const bool aa = parts[i].rfind("aa:", 0) == 0, nucl = parts[i].rfind("nucl:", 0) == 0;
                if (!aa && !nucl) continue;
                const std::string num = parts[i].substr(aa ? 3 : 5);
                if (num.empty() || num.find(':') != std::string::npos || !component(num)) ok = false;
                haveAA = haveAA || aa; haveNucl = haveNucl || nucl;
            }
            ok = ok && haveAA && haveNucl;
        }
        return ok ? "" : "Error in value parsing " + flag;
    }
    regex_t re;
    if (regcomp(&re, d->regex, REG_EXTENDED | REG_NEWLINE) != 0) return std::string("Error in regex ") + d->regex;
    const int nomatch = regexec(&re, value.c_str(), 0, nullptr, 0);
    regfree(&re);
    if (!nomatch) return "";
    return (d->kind == V_BYTES ? "Error in argument regex " : "Error in argument ") + flag;
}

const FlagSpec *findFlag(const std::string &a, std::initializer_list<const FlagSpec *> tables) {
    for (const FlagSpec *t : tables)
        for (; t->name; t++)
            if (a == t->name) return t;
    return nullptr;
}

bool valueAllowed(const std::string &v, const char *only) {
    std::string rest = only;
    for (;;) {
        const size_t bar = rest.find('|');
        const std::string one = rest.substr(0, bar);
        if (v == one) return true;
        // a matrix may be given as a path: compare the file name
        if (!one.empty() && v.size() > one.size() && v.compare(v.size() - one.size(), one.size(), one) == 0 && v[v.size() - one.size() - 1] == '/') return true;
        if (bar == std::string::npos) return false;
        rest = rest.substr(bar + 1);
    }
}

// Parses the argument vector against the module's tables.  On failure `err` holds the message (reference wording for
// unknown flags and malformed booleans) and the caller returns EXIT_FAILURE.
bool parseArgs(int argc, const char **argv, const char *module, std::initializer_list<const FlagSpec *> tables, Options &o, std::string &err) {
    for (int i = 0; i < argc; i++) {
        const std::string a = argv[i];
        const bool isFlag = a.size() > 1 && a[0] == '-' && !isdigit((unsigned char) a[1]) && a[1] != '.';
        if (!isFlag) { o.pos.push_back(a); continue; }
        const FlagSpec *f = findFlag(a, tables);
        if (!f) { err = "Unrecognized parameter \"" + a + "\""; return false; }
        if (o.kv.count(a)) { err = "Duplicate parameter " + a; return false; }        // Parameters.cpp:1895-1899
        std::string v;
        if (f->isBool) {
            if (i + 1 == argc || argv[i + 1][0] == '-') {
                // a bare bool flag toggles the module default (all defaults here are false except --diag-score)
                v = (a == "--diag-score") ? "0" : "1";
            } else {
                const std::string b = argv[++i];
                if (b == "true" || b == "TRUE" || b == "1") v = "1";
                else if (b == "false" || b == "FALSE" || b == "0") v = "0";
                else { err = "Invalid boolean string " + b; return false; }
            }
        } else {
            if (i + 1 >= argc) { err = "Missing argument " + a; return false; }
            v = argv[++i];
            err = domainError(a, v);
            if (!err.empty()) return false;
        }
        if (f->support == ONLY && !valueAllowed(Options::first(v), f->only)) {
            err = std::string(module) + ": " + a + " " + v + " is not implemented on the device path (supported: " +
                  (f->only[0] ? f->only : "unset") + ")";
            return false;
        }
        o.kv[a] = v;
    }
    return true;
}

int fail(const std::string &msg) {
    fprintf(stderr, "%s\n", msg.c_str());
    return EXIT_FAILURE;
}

// Target database in the padded GPU layout, either taken as is from disk or built in memory from an ASCII DB.
struct PaddedTarget {
    std::vector<uint8_t> own3di, ownAA;
    const uint8_t *d3 = nullptr, *dA = nullptr;
    std::vector<uint64_t> offsets;
    std::vector<int32_t> lengths;
    std::vector<uint32_t> keys;
    uint64_t bytes = 0;
};

// contiguous ranges of [0, n) on up to 8 host threads (DB encoding at start-up: 350 M residues per alphabet at 1M targets)
template <typename F>
void parallelRanges(size_t n, F fn) {
    const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t) 8, (size_t) std::max(1, fshost_usable_cores()), n / 4096 + 1}));
    if (T == 1) { fn((size_t) 0, n); return; }
    std::vector<std::thread> ths;
    for (size_t t = 1; t < T; t++) ths.emplace_back(fn, n * t / T, n * (t + 1) / T);
    fn((size_t) 0, n / T);
    for (auto &th : ths) th.join();
}

bool loadPadded(const DbReader &r3, const DbReader *rA, const Matrix &m3, const Matrix *mA, PaddedTarget &t, std::string &err) {
    const size_t n = r3.size();
    t.offsets.resize(n + 1); t.lengths.resize(n); t.keys.resize(n);
    const bool padded = (r3.extended() & DBTYPE_EXTENDED_GPU) != 0;
    const bool paddedAA = rA && (rA->extended() & DBTYPE_EXTENDED_GPU) != 0;
    if (rA && rA->size() != n) { err = "AA and 3Di target databases differ in size"; return false; }
    if (padded) {
        for (size_t i = 0; i < n; i++) { t.offsets[i] = r3.offset(i); t.lengths[i] = (int32_t) r3.seqLen(i); t.keys[i] = r3.key(i); }
        t.offsets[n] = r3.dataSize();
        t.d3 = (const uint8_t *) r3.dataBase();
        t.bytes = r3.dataSize();
    } else {
        // ASCII database: encode in index order (lower case = soft-masked -> code + 32), pad each entry to a multiple of 4
        uint64_t off = 0;
        for (size_t i = 0; i < n; i++) {
            t.offsets[i] = off; t.lengths[i] = (int32_t) r3.seqLen(i); t.keys[i] = r3.key(i);
            off += ((uint64_t) t.lengths[i] + 3) / 4 * 4;
        }
        t.offsets[n] = off; t.bytes = off;
        t.own3di.assign(off, 20);
        parallelRanges(n, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                const char *s = r3.data(i);
                uint8_t *dst = t.own3di.data() + t.offsets[i];
                for (int k = 0; k < t.lengths[i]; k++) {
                    const uint8_t c = m3.aa2num[(unsigned char) s[k]];
                    dst[k] = (uint8_t) (islower((unsigned char) s[k]) ? c + 32 : c);
                }
            }
        });
        t.d3 = t.own3di.data();
    }
    if (!rA) return true;
    if (paddedAA) {
        // both halves padded by base:makepaddedseqdb: same order, same offsets
        if (!padded) { err = "padded AA database with an unpadded 3Di database"; return false; }
        for (size_t i = 0; i < n; i++)
            if (rA->offset(i) != r3.offset(i) || rA->seqLen(i) != r3.seqLen(i)) { err = "padded AA and 3Di databases are not aligned"; return false; }
        t.dA = (const uint8_t *) rA->dataBase();
        return true;
    }
    // ASCII AA database.  For a padded target this is what Foldseek's makepaddedseqdb workflow leaves behind
    // (F/data/makepaddeddb.sh:18-40): only <db>_ss is re-encoded, <db> is the source AA data file linked under the new
    // name with an index whose keys were renamed to the padded ids (renamedbkeys).  The device wants the AA codes at the
    // 3Di offsets, so they are encoded here, entry by entry, matched by KEY.
    t.ownAA.assign(t.bytes, 20);
    std::atomic<int64_t> badEntry(-1);
    std::atomic<int> badKind(0);
    parallelRanges(n, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1 && badEntry.load(std::memory_order_relaxed) < 0; i++) {
            const int64_t ia = rA->idOf(t.keys[i]);
            if (ia < 0) { badKind = 1; badEntry = (int64_t) i; return; }
            if ((int32_t) rA->seqLen((size_t) ia) != t.lengths[i]) { badKind = 2; badEntry = (int64_t) i; return; }
            const char *a = rA->data((size_t) ia);
            uint8_t *dst = t.ownAA.data() + t.offsets[i];
            for (int k = 0; k < t.lengths[i]; k++) dst[k] = mA->aa2num[(unsigned char) a[k]];
        }
    });
    if (badEntry >= 0) {
        const std::string key = std::to_string(t.keys[(size_t) badEntry.load()]);
        err = badKind == 1 ? "AA target database has no entry with key " + key : "AA and 3Di entries of key " + key + " differ in length";
        return false;
    }
    t.dA = t.ownAA.data();
    return true;
}

void fillParams(const Options &o, fshost_params &p) {
    fshost_params_default(&p);
    p.maxResListLen = o.geti("--max-seqs", p.maxResListLen);
    p.minDiagScoreThr = o.geti("--min-ungapped-score", p.minDiagScoreThr);
    p.compBiasCorrection = o.geti("--comp-bias-corr", p.compBiasCorrection);
    p.alignmentType = o.geti("--alignment-type", p.alignmentType);
    p.gapOpen = o.geti("--gap-open", p.gapOpen);
    p.gapExtend = o.geti("--gap-extend", p.gapExtend);
    p.evalThr = o.getd("-e", p.evalThr);
    p.covThr = (float) o.getd("-c", p.covThr);
    p.covMode = o.geti("--cov-mode", p.covMode);
    p.addBacktrace = (o.geti("-a", 0) || o.geti("--add-backtrace", 0)) ? 1 : 0;
    p.maxAccept = o.geti("--max-accept", p.maxAccept);
    p.maxRejected = o.geti("--max-rejected", p.maxRejected);
    p.seqIdThr = (float) o.getd("--min-seq-id", p.seqIdThr);
    p.alnLenThr = o.geti("--min-aln-len", p.alnLenThr);
    p.seqIdMode = o.geti("--seq-id-mode", p.seqIdMode);
    p.altAlignment = std::max(0, o.geti("--alt-ali", p.altAlignment));
}

// Sequence::mapSequence cuts entries at --max-seq-len (M/src/commons/Sequence.cpp:289-300); truncation is not implemented
// here, so a database with longer entries than the limit is refused instead of being searched differently.
bool checkMaxSeqLen(const Options &o, const DbReader &r, const std::string &what, std::string &err) {
    const long maxLen = std::min<long>(o.geti("--max-seq-len", FSGPU_MAX_SEQ_LEN), FSGPU_MAX_SEQ_LEN);
    for (size_t i = 0; i < r.size(); i++)
        if ((long) r.seqLen(i) > maxLen) {
            err = what + " entry " + std::to_string(r.key(i)) + " has " + std::to_string(r.seqLen(i)) + " residues: longer than --max-seq-len " +
                  std::to_string(maxLen) + " (truncation is not implemented on the device path)";
            return false;
        }
    return true;
}

// --sort-by-structure-bits 1 (default of the workflow) rescales scores by TM-score x LDDT and needs both C-alpha DBs
// (structurealign.cpp:182-197): without them the reference warns and turns it off -- so do we; WITH them the rescoring
// would run, which this path does not implement: refuse.
bool resolveStructureBits(const Options &o, const std::string &qdb, const std::string &tdb, std::string &err) {
    if (o.geti("--sort-by-structure-bits", 1) == 0) return true;
    auto exists = [](const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; };
    if (exists(qdb + "_ca.dbtype") && exists(tdb + "_ca.dbtype")) {
        err = "structurealign: --sort-by-structure-bits 1 with C-alpha databases (TM-score / LDDT rescoring) is not implemented on the device path; pass --sort-by-structure-bits 0";
        return false;
    }
    fprintf(stderr, "Cannot find %s C-alpha or %s C-alpha database\nDisabling --sort-by-structure-bits\n", qdb.c_str(), tdb.c_str());
    return true;
}

// ---- devices ------------------------------------------------------------------------------------------------------
// --gpus N (or "all"): the target DB is loaded once on --gpu-device, replicated to N - 1 more devices with ONE broadcast
// (fsgpu_db_broadcast: RCCL over xGMI, peer copies as fallback), and host threads -- `--threads` per GPU -- pull queries
// from one shared counter: dynamic query sharding, no communication after the broadcast.
struct DeviceSet {
    std::vector<fsgpu_ctx *> root;            // one context per GPU, each owning that GPU's resident DB (+ k-mer index)
    int perGpuThreads = 1;
    int threads() const { return (int) root.size() * perGpuThreads; }
    bool open(const Options &o, const PaddedTarget &pt, bool withAA, int defaultThreads, std::string &err, bool alignFeeders = false) {
        int count = 0;
        const int first = o.geti("--gpu-device", 0);
        auto it = o.kv.find("--gpus");
        int want = 1;
        fsgpu_ctx *c0 = nullptr;
        if (fsgpu_create(first, &c0) != FSGPU_OK) { err = std::string("GPU: ") + fsgpu_last_error(nullptr); return false; }
        root.push_back(c0);
        if (it != o.kv.end()) {
            count = fsgpu_device_count();
            want = it->second == "all" ? count - first : atoi(it->second.c_str());
            if (want < 1 || first + want > count) { err = "--gpus: " + it->second + " devices requested from device " + std::to_string(first) + ", " + std::to_string(count) + " visible"; return false; }
        }
        if (fsgpu_db_load(c0, pt.d3, withAA ? pt.dA : nullptr, pt.offsets.data(), pt.lengths.data(), pt.lengths.size(), pt.bytes) != FSGPU_OK) {
            err = std::string("GPU: ") + fsgpu_last_error(c0); return false;
        }
        for (int d = 1; d < want; d++) {
            fsgpu_ctx *c = nullptr;
            if (fsgpu_create(first + d, &c) != FSGPU_OK) { err = std::string("GPU: ") + fsgpu_last_error(nullptr); return false; }
            root.push_back(c);
        }
        if (want > 1) {
            int usedRccl = 0;
            if (fsgpu_db_broadcast(c0, root.data() + 1, want - 1, &usedRccl) != FSGPU_OK) { err = std::string("GPU: ") + fsgpu_last_error(c0); return false; }
            fprintf(stderr, "target DB replicated to %d GPUs (%s)\n", want, usedRccl ? "RCCL broadcast" : "peer copies");
            // FSGPU_REQUIRE_RCCL=1: a multi-GPU run whose replication silently fell back to peer copies is an error (scaling runs set it)
            { const char *e = getenv("FSGPU_REQUIRE_RCCL"); if (!usedRccl && e && *e && *e != '0' && !getenv("FSGPU_NO_RCCL")) { err = "GPU: the RCCL broadcast of the target DB was not used (FSGPU_REQUIRE_RCCL=1)"; return false; } }
        }
        if (want == 1) {
            // FSGPU_REQUIRE_RCCL=1 on one GPU: nothing to replicate, but the library the replication would use must work on this device
            const char *e = getenv("FSGPU_REQUIRE_RCCL");
            if (e && *e && *e != '0' && !getenv("FSGPU_NO_RCCL")) {
                if (fsgpu_rccl_selfcheck(c0) != FSGPU_OK) { err = std::string("GPU: ") + fsgpu_last_error(c0) + " (FSGPU_REQUIRE_RCCL=1)"; return false; }
                fprintf(stderr, "RCCL self-check on device %d: one-rank communicator + ncclBroadcast ok (usedRccl 1)\n", first);
            }
        }
        // Every feeder thread beyond the first owns a context clone with its own scratch -- for the k-mer prefilter up to 2 x 2.4e8 index hits x 24 B
        // = 11.5 GB each: 16 of them fit a 288 GB device next to the resident DB and its index, 32 need not (the Prefiltering adapter inside the
        // reference, adapters/prefiltering_fsgpu.inc, uses the same cap).  The aligning modules below take at most 8 feeders anyway.
        perGpuThreads = std::max(1, std::min(o.geti("--threads", defaultThreads), alignFeeders ? 32 : 16));
        if (alignFeeders) {
            // Modules that align (search, structurealign): --threads is the number of cores the job may use, as for the reference's OpenMP
            // loop.  The per-hit backtraces -- most of the host work once a hit list holds real homologs -- run in the process-wide worker
            // pool (search.cpp::HostPool, --threads workers), the feeder threads issue the device batches and sleep while they wait: half
            // as many feeders as cores, at most 8 per GPU.  All-vs-all of 200 000 structures, 16-core quota (tools/allvsall_ab2.sh):
            // 16 feeders + 6 workers 3.23 s, 12 + 8 3.07 s, 8 + 8 2.81 s, 8 + 16 2.58 s, 4 + 14 2.77 s.  FSGPU_FEEDERS / FSGPU_HOST_WORKERS override.
            const int cores = perGpuThreads;
            const char *ef = getenv("FSGPU_FEEDERS");
            // (round 5: at least two feeders from two cores on -- with the backtraces of a batch on the device a feeder mostly sleeps in its waits, and a
            // lone feeder leaves the device idle while it prepares, gates and formats: one rank's share of a 16-core node is two cores)
            perGpuThreads = ef ? std::max(1, std::min(atoi(ef), 32)) : std::max(cores >= 2 ? 2 : 1, std::min(8, (cores + 1) / 2));
            if (!getenv("FSGPU_HOST_WORKERS")) fshost_set_host_workers(cores >= 4 ? cores : std::max(0, cores - 1));
        }
        const char *rq = getenv("FSGPU_REQUIRE_RCCL");
        if (moduleTiming() || (rq && *rq && *rq != '0'))
            fprintf(stderr, "host budget: %d usable cores (cgroup quota), %d GPU(s), %d feeder thread(s) per GPU, %d backtrace workers\n",
                    fshost_usable_cores(), (int) root.size(), perGpuThreads, fshost_host_workers());
        return true;
    }
    // worker tix runs on GPU tix % nGpus; the first worker of a GPU uses its root context, the others a clone
    fsgpu_ctx *forThread(int tix, bool &owned) {
        fsgpu_ctx *r = root[tix % root.size()];
        owned = tix >= (int) root.size();
        if (!owned) return r;
        fsgpu_ctx *c = nullptr;
        return fsgpu_clone(r, &c) == FSGPU_OK ? c : nullptr;
    }
    void close() { for (fsgpu_ctx *c : root) fsgpu_destroy(c); root.clear(); }
};

// ---- gpuserver protocol ------------------------------------------------------------------------------------------
volatile sig_atomic_t gKeepRunning = 1;
void onSignal(int) { gKeepRunning = 0; }
void installSignalHandlers() {
    struct sigaction act;
    memset(&act, 0, sizeof(act));
    act.sa_handler = onSignal;
    sigaction(SIGINT, &act, nullptr);
    sigaction(SIGTERM, &act, nullptr);
}

std::string shmNameFor(const Options &o, const std::string &db) {
    auto it = o.kv.find("--shm-name");
    if (it != o.kv.end()) return it->second;
    // the reference hashes its own git version string into the name (GpuUtil.cpp:18-34): to pair with a reference
    // binary pass that string with --gpu-server-version (or FSGPU_SERVER_VERSION); both of our sides default to the same
    const char *ver = getenv("FSGPU_SERVER_VERSION");
    auto iv = o.kv.find("--gpu-server-version");
    std::string version = iv != o.kv.end() ? iv->second : (ver ? ver : "fsgpu-amd");
    const char *dev = getenv("HIP_VISIBLE_DEVICES");
    if (!dev) dev = getenv("CUDA_VISIBLE_DEVICES");
    return gpuShmName(db, dev, version.c_str());
}

// client: the query loop of runFilterOnGpu in server mode (ungappedprefilter.cpp:171-327), one query in flight
int ungappedPrefilterViaServer(const Options &o, const DbReader &q, const DbReader &t, bool sameDB, const fshost_params &par, const Matrix &m3) {
    const std::string name = shmNameFor(o, o.pos[1]);
    const int waitTimeout = o.geti("--gpu-server-wait-timeout", 600);
    const auto start = std::chrono::steady_clock::now();
    bool printed = false;
    while (!gpuShmExists(name)) {
        if (waitTimeout == 0) return fail("gpuserver for database " + o.pos[1] + " not found.\nPlease start gpuserver with the same HIP_VISIBLE_DEVICES");
        if (!printed) { fprintf(stderr, "Waiting for `gpuserver`\n"); printed = true; }
        const auto elapsed = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - start).count();
        if (waitTimeout > 0 && elapsed >= waitTimeout)
            return fail("gpuserver for database " + o.pos[1] + " not found after " + std::to_string(elapsed) + " seconds.\nPlease start gpuserver with the same HIP_VISIBLE_DEVICES");
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
    }
    std::string err;
    GpuShm *shm = gpuShmOpen(name, err);
    if (!shm) return fail(err);
    installSignalHandlers();
    DbWriter w;
    if (!w.open(o.pos[2], DBTYPE_PREFILTER_RES, err)) { gpuShmUnmap(shm); return fail(err); }
    std::vector<uint8_t> codes;
    std::vector<int8_t> pssm;
    // geometry as validated by gpuShmOpen, read ONCE: the areas are addressed through these, not through header fields re-read later
    const unsigned int shmMaxSeqLen = shm->maxSeqLen, shmMaxRes = shm->maxResListLen;
    int8_t *const shmQuery = shm->query(), *const shmProfile = shm->profile();
    const GpuShmResult *const shmResults = shm->results();
    std::vector<GpuShmResult> results(shmMaxRes);
    std::vector<fsgpu_hit> hits;
    std::string out;
    char line[128];
    int rc = EXIT_SUCCESS;
    for (size_t id = 0; id < q.size() && rc == EXIT_SUCCESS; id++) {
        const uint32_t L = q.seqLen(id);
        out.clear();
        if (L > 0) {
            if (L > shmMaxSeqLen) { rc = fail("query longer than the gpuserver's --max-seq-len"); break; }
            codes.resize(L);
            const char *sq = q.data(id);
            for (uint32_t i = 0; i < L; i++) codes[i] = m3.aa2num[(unsigned char) sq[i]];
            pssm.resize((size_t) m3.n * L);
            int cap = 0;
            if (prefilterProfile(m3, codes.data(), (int) L, par.compBiasCorrection != 0, par.prefCompBiasScale, pssm.data(), &cap) != FSGPU_OK) { rc = fail("bad query residue code"); break; }
            unsigned int nres = 0;
            for (bool claimed = false; !claimed;) {
                if (shm->serverExit.load(std::memory_order_acquire)) { rc = fail("GPU server has unexpectedly shut down"); break; }
                if (!gKeepRunning) { rc = EXIT_FAILURE; break; }
                int expected = GpuShm::IDLE;
                if (!shm->state.compare_exchange_strong(expected, GpuShm::RESERVED, std::memory_order_acq_rel)) { std::this_thread::yield(); continue; }
                claimed = true;
                memcpy(shmQuery, codes.data(), L);
                memcpy(shmProfile, pssm.data(), (size_t) m3.n * L);
                shm->queryLen = L;
                std::atomic_thread_fence(std::memory_order_release);
                shm->state.store(GpuShm::READY, std::memory_order_release);
                while (shm->state.load(std::memory_order_acquire) != GpuShm::DONE) {
                    if (shm->serverExit.load(std::memory_order_acquire)) { rc = fail("GPU server has unexpectedly shut down"); break; }
                    std::this_thread::yield();
                }
                if (rc != EXIT_SUCCESS) break;
                std::atomic_thread_fence(std::memory_order_acquire);
                nres = std::min(shm->resultLen, shmMaxRes);
                memcpy(results.data(), shmResults, nres * sizeof(GpuShmResult));
                shm->state.store(GpuShm::IDLE, std::memory_order_release);
            }
            if (rc != EXIT_SUCCESS) break;
            hits.clear();
            const uint32_t qKey = q.key(id);
            for (unsigned int i = 0; i < nres; i++) {
                if (results[i].id >= t.size()) { rc = fail("gpuserver returned a target id outside the database (different DB?)"); break; }
                const uint32_t tKey = t.key(results[i].id);
                const bool isIdentity = qKey == tKey && sameDB;                          // includeIdentity: not a flag of this module
                if (isIdentity || results[i].score > par.minDiagScoreThr) hits.push_back({tKey, results[i].score});
            }
            std::sort(hits.begin(), hits.end(), [](const fsgpu_hit &a, const fsgpu_hit &b) {   // hit_t::compareHitsByScoreAndId
                if (a.score != b.score) return a.score > b.score;
                return a.id < b.id;
            });
            const size_t n = std::min<size_t>(hits.size(), (size_t) par.maxResListLen);
            for (size_t k = 0; k < n; k++) out.append(line, fshost_format_prefilter_hit(line, hits[k].id, hits[k].score, 0));
        }
        w.write(q.key(id), out.data(), out.size());
    }
    gpuShmUnmap(shm);
    if (rc != EXIT_SUCCESS) return rc;
    if (!w.close(err)) return fail(err);
    return EXIT_SUCCESS;
}

// Util::parseFastaHeader (M/src/commons/Util.cpp:147-229): the accession of a header line -- the field between the
// database prefix ("sp|", "gi|x|y|" ...) and the next '|' or blank, else the first word
// `avail`: bytes of the header entry (an entry that lost its terminator must not be read past its end)
std::string fastaHeaderName(const char *headerPtr, size_t avail) {
    size_t len = 0;
    while (len < avail && headerPtr[len] != '\0' && !isspace((unsigned char) headerPtr[len])) len++;
    const std::string header(headerPtr, len);
    if (header.empty()) return "";
    size_t offset = header.compare(0, 10, "consensus_") == 0 ? 10 : 0;
    static const struct { const char *prefix; unsigned int length, bar; } dbs[] = {
        {"cl|", 3, 1}, {"sp|", 3, 1}, {"tr|", 3, 1}, {"gb|", 3, 1}, {"ref|", 4, 1}, {"pdb|", 4, 1}, {"bbs|", 4, 1}, {"lcl|", 4, 1},
        {"pir||", 5, 1}, {"prf||", 5, 1}, {"gnl|", 4, 2}, {"pat|", 4, 2}, {"gi|", 3, 3}};
    for (const auto &d : dbs) {
        if (header.compare(offset, d.length, d.prefix) != 0) continue;
        size_t start = offset + d.length;
        for (unsigned int j = 0; j + 1 < d.bar; j++) {
            const size_t end = header.find('|', start);
            if (end == std::string::npos) return "";
            start = end + 1;
        }
        size_t end = header.find('|', start);
        if (end == std::string::npos) end = header.find_first_of(" \n", start);
        return header.substr(start, end == std::string::npos ? std::string::npos : end - start);
    }
    const size_t end = header.find_first_of(" \n", offset);
    return header.substr(offset, end == std::string::npos ? std::string::npos : end - offset);
}

// entry numbers of a sequence DB, longest first (stable)
std::vector<size_t> lengthOrder(const DbReader &r) {
    std::vector<size_t> order(r.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return r.seqLen(x) > r.seqLen(y); });
    return order;
}

} // namespace

extern "C" {

int fsmod_makepaddedseqdb(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "makepaddedseqdb", {kPaddedFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 2) return fail("usage: makepaddedseqdb <sequenceDB> <outPaddedDB> [--write-lookup 0|1]");
    DbReader r, rh;
    std::string err;
    if (!r.open(o.pos[0], err)) return fail(err);
    // header DB <in>_h (par.hdr1): the reference opens it unconditionally (makepaddedseqdb.cpp:22-23); databases written by
    // test harnesses may lack it, then names in the lookup fall back to the key
    std::string herr;
    const bool haveHeaders = rh.open(o.pos[0] + "_h", herr);
    Matrix m;
    if (!m.builtin(FSHOST_MAT_3DI, 2.0f, 0.0f)) return fail("matrix construction failed");
    const size_t n = r.size();
    // DBReader SORT_BY_LENGTH = descending length, ties ascending id (M/src/commons/DBReader.h:436-448), iterated backwards
    std::vector<size_t> ord(n);
    for (size_t i = 0; i < n; i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return r.seqLen(a) > r.seqLen(b); });
    std::reverse(ord.begin(), ord.end());
    const bool writeLookup = o.geti("--write-lookup", 1) != 0;
    FILE *fd = fopen(o.pos[1].c_str(), "wb");
    FILE *fi = fopen((o.pos[1] + ".index").c_str(), "wb");
    FILE *fl = writeLookup ? fopen((o.pos[1] + ".lookup").c_str(), "wb") : nullptr;
    FILE *fh = haveHeaders ? fopen((o.pos[1] + "_h").c_str(), "wb") : nullptr;
    FILE *fhi = haveHeaders ? fopen((o.pos[1] + "_h.index").c_str(), "wb") : nullptr;
    if (!fd || !fi || (writeLookup && !fl) || (haveHeaders && (!fh || !fhi))) return fail("cannot create " + o.pos[1]);
    bool ioOk = true;
    uint64_t off = 0, hoff = 0;
    std::string out;
    for (size_t k = 0; k < n && ioOk; k++) {
        const size_t id = ord[k];
        const char *s = r.data(id);
        const uint32_t L = r.seqLen(id);
        out.clear();
        for (uint32_t i = 0; i < L; i++) {
            const uint8_t c = m.aa2num[(unsigned char) s[i]];
            out.push_back((char) (islower((unsigned char) s[i]) ? c + 32 : c));
        }
        out.append((L % 4 == 0) ? 0 : 4 - L % 4, (char) 20);
        ioOk = ioOk && fwrite(out.data(), 1, out.size(), fd) == out.size();
        ioOk = ioOk && fprintf(fi, "%zu\t%llu\t%u\n", k, (unsigned long long) off, L + 2) > 0;
        off += out.size();
        std::string name = std::to_string(r.key(id));
        if (haveHeaders) {
            const int64_t hid = rh.idOf(r.key(id));
            if (hid < 0) return fail("Invalid header key " + std::to_string(r.key(id)) + ".");
            const char *h = rh.data((size_t) hid);
            const uint32_t hl = rh.entryLen((size_t) hid);
            ioOk = ioOk && fwrite(h, 1, hl, fh) == hl;                       // entry incl. its terminator, verbatim
            ioOk = ioOk && fprintf(fhi, "%zu\t%llu\t%u\n", k, (unsigned long long) hoff, hl) > 0;
            hoff += hl;
            name = fastaHeaderName(h, hl);
        }
        // lookup: new key, entry name parsed from the header, ORIGINAL key in the file-number column (makepaddedseqdb.cpp:121-138)
        if (fl) ioOk = ioOk && fprintf(fl, "%zu\t%s\t%u\n", k, name.c_str(), r.key(id)) > 0;
    }
    ioOk = (fclose(fd) == 0) && ioOk;
    ioOk = (fclose(fi) == 0) && ioOk;
    if (fl) ioOk = (fclose(fl) == 0) && ioOk;
    if (fh) ioOk = (fclose(fh) == 0) && ioOk;
    if (fhi) ioOk = (fclose(fhi) == 0) && ioOk;
    auto writeType = [&](const std::string &path, int32_t t) {
        FILE *ft = fopen(path.c_str(), "wb");
        if (!ft) return false;
        const bool ok = fwrite(&t, 4, 1, ft) == 1;
        return (fclose(ft) == 0) && ok;
    };
    ioOk = writeType(o.pos[1] + ".dbtype", (int32_t) ((uint32_t) r.dbtype() | ((uint32_t) (r.extended() | DBTYPE_EXTENDED_GPU) << 16))) && ioOk;
    if (haveHeaders) ioOk = writeType(o.pos[1] + "_h.dbtype", 12 /* DBTYPE_GENERIC_DB */) && ioOk;
    if (!ioOk) return fail("write error on " + o.pos[1]);
    return EXIT_SUCCESS;
}

int fsmod_ungappedprefilter(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "ungappedprefilter", {kUngappedFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 3) return fail("usage: ungappedprefilter <queryDB_ss> <targetDB_ss> <outPrefDB> [--max-seqs N] [--min-ungapped-score S] [--comp-bias-corr 0|1] [--threads T]");
    std::string err;
    DbReader q, t;
    if (!q.open(o.pos[0], err) || !t.open(o.pos[1], err)) return fail(err);
    const bool sameDB = o.pos[0] == o.pos[1];
    fshost_params par;
    fillParams(o, par);
    par.prefCompBiasScale = (float) o.getd("--comp-bias-corr-scale", 0.15);
    Matrix m3;
    m3.builtin(FSHOST_MAT_3DI, 2.0f, 0.0f);
    if (o.geti("--gpu-server", 0) != 0) return ungappedPrefilterViaServer(o, q, t, sameDB, par, m3);
    PaddedTarget pt;
    if (!loadPadded(t, nullptr, m3, nullptr, pt, err)) return fail(err);
    DeviceSet ds;
    if (!ds.open(o, pt, false, 3, err)) { ds.close(); return fail(err); }
    DbWriter w;
    if (!w.open(o.pos[2], DBTYPE_PREFILTER_RES, err)) { ds.close(); return fail(err); }
    const int nthreads = ds.threads();
    std::vector<std::string> results(q.size());
    // queries are taken in order of decreasing length (the result DB is keyed, the order of the work is free): the queries of a batch
    // then fall into one or two register classes and share their scan launches (fsgpu_gapless_scan_multi)
    const std::vector<size_t> order = lengthOrder(q);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::string firstErr;
    auto work = [&](int tix) {
        bool owned = false;
        fsgpu_ctx *ctx = ds.forThread(tix, owned);
        if (!ctx) { bad++; return; }
        fshost_search *s = fshost_search_create(ctx, &par, pt.keys.data(), nullptr, pt.d3, nullptr, pt.offsets.data(), pt.lengths.data());
        // batches of queries: one multi-query scan per batch (queries of equal ceil(L / 16) share a launch)
        const size_t batch = 32;
        std::vector<fsgpu_hit> hits(batch * (size_t) par.maxResListLen);
        std::vector<std::vector<uint8_t>> codes(batch);
        std::vector<const uint8_t *> pq(batch);
        std::vector<int> Ls(batch), nh(batch);
        std::vector<int64_t> ident(batch);
        std::vector<size_t> qid(batch);
        char line[128];
        for (;;) {
            const size_t b0 = next.fetch_add(batch);
            if (b0 >= q.size() || bad) break;
            size_t m = 0;
            for (size_t oi = b0; oi < std::min(q.size(), b0 + batch); oi++) {
                const size_t id = order[oi];
                const uint32_t L = q.seqLen(id);
                if (L == 0) continue;
                codes[m].resize(L);
                const char *sq = q.data(id);
                for (uint32_t i = 0; i < L; i++) codes[m][i] = m3.aa2num[(unsigned char) sq[i]];
                pq[m] = codes[m].data(); Ls[m] = (int) L; qid[m] = id;
                ident[m] = sameDB ? t.idOf(q.key(id)) : -1;
                m++;
            }
            if (m == 0) continue;
            if (fshost_search_prefilter_batch(s, (int) m, pq.data(), Ls.data(), ident.data(), hits.data(), nh.data()) != FSGPU_OK) {
                if (!bad++) firstErr = fshost_search_error(s);
                break;
            }
            for (size_t k = 0; k < m; k++) {
                std::string &out = results[qid[k]];
                const fsgpu_hit *h = hits.data() + k * (size_t) par.maxResListLen;
                for (int i = 0; i < nh[k]; i++) out.append(line, fshost_format_prefilter_hit(line, pt.keys[h[i].id], h[i].score, 0));
            }
        }
        fshost_search_free(s);
        if (owned) fsgpu_destroy(ctx);
    };
    std::vector<std::thread> ths;
    for (int i = 1; i < nthreads; i++) ths.emplace_back(work, i);
    work(0);
    for (auto &th : ths) th.join();
    ds.close();
    if (bad) return fail("ungappedprefilter failed: " + firstErr);
    for (size_t id = 0; id < q.size(); id++) w.write(q.key(id), results[id].data(), results[id].size());   // empty entries too
    if (!w.close(err)) return fail(err);
    return EXIT_SUCCESS;
}

// Util::canBeCovered (M/src/commons/Util.cpp:542-559) for the three modes runSplit applies it to (Prefiltering.cpp:880-887)
static bool canBeCovered(float covThr, int covMode, float q, float t) {
    switch (covMode) {
        case 0: return (q / t >= covThr) && (t / q >= covThr);
        case 2: return (t / q) >= covThr;
        case 5: return (std::min(t, q) / std::max(t, q)) >= covThr;
        default: return true;
    }
}

int fsmod_prefilter(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "prefilter", {kPrefilterFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 3) return fail("usage: prefilter <queryDB_ss> <targetDB_ss> <outPrefDB> [-s S] [-k 6] [--k-score T] [--max-seqs N] [--min-ungapped-score S] "
                                       "[--comp-bias-corr 0|1] [--comp-bias-corr-scale F] [--mask-lower-case 0|1] [--mask-n-repeat N] [--spaced-kmer-mode 0|1] "
                                       "[--add-self-matches 0|1] [-c F --cov-mode M] [--threads T]");
    std::string err;
    DbReader q, t;
    if (!q.open(o.pos[0], err) || !t.open(o.pos[1], err)) return fail(err);
    const bool sameDB = o.pos[0] == o.pos[1];
    const bool includeIdentical = o.geti("--add-self-matches", 0) != 0;
    if (!checkMaxSeqLen(o, q, "query", err) || !checkMaxSeqLen(o, t, "target", err)) return fail(err);
    // defaults as Foldseek sets them for its prefilter call (F/src/workflow/StructureSearch.cpp:101, F/src/commons/LocalParameters.cpp:382-412)
    const int kmerSize = o.geti("-k", 0) == 0 ? 6 : o.geti("-k", 6);     // k = 0: auto -> 6 below 3.35e9 residues (IndexTable.h:456-458)
    if (kmerSize != 6) return fail("prefilter: only -k 6 is implemented on the device path");
    if (t.residues() >= 3350000000ull && o.geti("-k", 0) == 0) return fail("prefilter: database needs k = 7, which is not implemented on the device path");
    if (o.geti("--exact-kmer-matching", 0) != 0 || o.geti("--mask", 0) != 0)
        return fail("prefilter: --exact-kmer-matching 1 and --mask 1 are not implemented on the device path");
    const float sens = (float) o.getd("-s", 9.5);
    // --k-score INT_MAX (the default the workflow passes) = derive the threshold from -s (Prefiltering.cpp:1036-1096)
    const int kmerThr = o.geti("--k-score", INT_MAX) != INT_MAX ? o.geti("--k-score", 0) : fshost_kmer_threshold(sens, kmerSize);
    const int spaced = o.geti("--spaced-kmer-mode", 1);
    const int maxRes = (int) std::min<uint64_t>((uint64_t) o.geti("--max-seqs", 1000), std::max<uint64_t>(t.size(), 1));
    const int compBias = o.geti("--comp-bias-corr", 1);
    const float cbScale = (float) o.getd("--comp-bias-corr-scale", 0.15);
    const float covThr = (float) o.getd("-c", 0.0);
    const int covMode = o.geti("--cov-mode", 0);
    fshost_matrix *m8 = fshost_matrix_create(FSHOST_MAT_3DI, 8.0f, -0.2f), *m2 = fshost_matrix_create(FSHOST_MAT_3DI, 2.0f, -0.2f);
    Matrix m3;
    if (!m8 || !m2 || !m3.builtin(FSHOST_MAT_3DI, 2.0f, 0.0f)) return fail("matrix construction failed");
    PaddedTarget pt;
    if (!loadPadded(t, nullptr, m3, nullptr, pt, err)) return fail(err);
    DeviceSet ds;
    if (!ds.open(o, pt, false, 2, err)) { ds.close(); return fail(err); }
    fsgpu_kmer_index_params ip;
    ip.kmerSize = kmerSize; ip.spaced = spaced; ip.kmerThr = kmerThr;
    ip.maskLowerCase = o.geti("--mask-lower-case", 1); ip.maskNrepeats = o.geti("--mask-n-repeat", 6);
    for (fsgpu_ctx *c : ds.root)      // every GPU builds its own index from its copy of the DB (0.1 - 0.6 s, cheaper than moving it)
        if (fsgpu_kmer_index_build(c, &ip, fshost_matrix_scores(m8)) != FSGPU_OK) { err = std::string("GPU: ") + fsgpu_last_error(c); ds.close(); return fail(err); }
    fsgpu_kmer_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.maxResListLen = maxRes; sp.minDiagScoreThr = o.geti("--min-ungapped-score", 30);
    sp.kmerScoreOnly = o.geti("--diag-score", 1) == 0 ? 1 : 0;        // the first step of easy-cluster's cascade: k-mer match counts instead of diagonal scores
    DbWriter w;
    if (!w.open(o.pos[2], DBTYPE_PREFILTER_RES, err)) { ds.close(); return fail(err); }
    const int nthreads = ds.threads();
    const size_t batch = 512;            // capacity of a thread's staging; a call takes fsgpu_kmer_batch_hint() queries (32 until the first call has measured the hit rate)
    std::vector<std::string> results(q.size());
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0), unstable(0);
    std::string firstErr;
    auto work = [&](int tix) {
        bool owned = false;
        fsgpu_ctx *ctx = ds.forThread(tix, owned);
        if (!ctx) { bad++; return; }
        std::vector<std::vector<uint8_t>> codes(batch);
        std::vector<std::vector<int16_t>> thr(batch);
        std::vector<std::vector<int8_t>> prof(batch);
        std::vector<fsgpu_kmer_query> qs(batch);
        std::vector<fsgpu_kmer_hit> hits(batch * (size_t) maxRes);
        std::vector<int32_t> nout(batch), status(batch);
        char line[128];
        for (;;) {
            // device batches as large as the hit rate allows, but the tail of the query set is shared out evenly over the threads
            const size_t seen = next.load();
            const size_t fair = std::max<size_t>(32, (q.size() > seen ? q.size() - seen : 0) / ((size_t) nthreads * 2));
            const size_t take = std::min<size_t>(std::min<size_t>(batch, (size_t) fsgpu_kmer_batch_hint(ctx)), fair);
            const size_t b0 = next.fetch_add(take);
            if (b0 >= q.size() || bad) break;
            const size_t nb = std::min(take, q.size() - b0);
            for (size_t k = 0; k < nb; k++) {
                const size_t id = b0 + k;
                const uint32_t L = q.seqLen(id);
                codes[k].resize(L + 1); thr[k].resize(L + 1); prof[k].resize((size_t) L * 21 + 1);
                const char *sq = q.data(id);
                for (uint32_t i = 0; i < L; i++) codes[k][i] = m3.aa2num[(unsigned char) sq[i]];
                fshost_kmer_query_prepare(m8, m2, codes[k].data(), (int) L, compBias, cbScale, kmerThr, kmerSize, spaced, thr[k].data(), prof[k].data());
                qs[k].seq = codes[k].data(); qs[k].kmerThr = thr[k].data(); qs[k].profile = prof[k].data(); qs[k].L = (int32_t) L; qs[k].reserved = 0;
                qs[k].identity = (sameDB || includeIdentical) ? t.idOf(q.key(id)) : -1;
            }
            if (fsgpu_kmer_search(ctx, &sp, qs.data(), (int) nb, hits.data(), nout.data(), status.data(), nullptr) != FSGPU_OK) {
                if (!bad++) firstErr = fsgpu_last_error(ctx);
                break;
            }
            for (size_t k = 0; k < nb; k++) {
                if (status[k] < 0) { if (!bad++) firstErr = "query " + std::to_string(q.key(b0 + k)) + ": hit buffers of the reference would overflow (status " + std::to_string(status[k]) + ")"; break; }
                if (status[k] == FSGPU_KMER_UNSTABLE) unstable++;
                std::string &out = results[b0 + k];
                const float qLen = (float) q.seqLen(b0 + k);
                for (int h = 0; h < nout[k]; h++) {
                    const fsgpu_kmer_hit &hit = hits[k * (size_t) maxRes + h];
                    if (covThr > 0.0 && (covMode == 0 || covMode == 2 || covMode == 5) && !canBeCovered(covThr, covMode, qLen, (float) pt.lengths[hit.id])) continue;
                    out.append(line, fshost_format_prefilter_hit(line, pt.keys[hit.id], hit.score, (int) (int16_t) hit.diagonal));
                }
            }
        }
        if (owned) fsgpu_destroy(ctx);
    };
    std::vector<std::thread> ths;
    for (int i = 1; i < nthreads; i++) ths.emplace_back(work, i);
    work(0);
    for (auto &th : ths) th.join();
    fshost_matrix_free(m8); fshost_matrix_free(m2);
    ds.close();
    if (bad) return fail("prefilter failed: " + firstErr);
    if (unstable) fprintf(stderr, "prefilter: %d queries matched more than half of the diagonal buffer; equal scores at the --max-seqs cut are ordered deterministically there (the reference's order is unspecified)\n", (int) unstable);
    for (size_t id = 0; id < q.size(); id++) w.write(q.key(id), results[id].data(), results[id].size());   // empty entries too (Prefiltering.cpp:900)
    if (!w.close(err)) return fail(err);
    return EXIT_SUCCESS;
}

// prefilter (k-mer: --prefilter-mode 0, gapless: --prefilter-mode 1) + structurealign fused in one process
int fsmod_search(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "search", {kSearchFlags, kPrefilterFlags, kAlignFlags, kStructAlignOnlyFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 3 && o.pos.size() != 4)
        return fail("usage: search <queryDB> <targetDB> <outAlnDB> [<outPrefDB>] [--prefilter-mode 0|1] [-s S] [--max-seqs N] [-e E] [--alignment-type 0|2] [-a] [--threads T] ...");
    std::string err;
    const double tStart = nowSec();
    std::atomic<int64_t> usPrep(0), usPref(0), usAlign(0), usFormat(0), usAlnPrep(0), usAlnDev(0), usAlnGate(0), usAlnBack(0), nRev(0), nPairs(0), nBtDev(0), nBtAll(0);
    DbReader qA, q3, tA, t3;
    if (!qA.open(o.pos[0], err) || !q3.open(dbPathWithSuffix(o.pos[0], "_ss"), err) || !tA.open(o.pos[1], err) || !t3.open(dbPathWithSuffix(o.pos[1], "_ss"), err)) return fail(err);
    if (qA.size() != q3.size()) return fail("query AA and 3Di databases differ in size");
    const bool sameDB = o.pos[0] == o.pos[1];
    const bool includeIdentical = o.geti("--add-self-matches", 0) != 0;
    const int prefMode = o.geti("--prefilter-mode", 0);
    if (prefMode != 0 && prefMode != 1) return fail("search: --prefilter-mode 0 (k-mer) or 1 (ungapped)");
    fshost_params par;
    fillParams(o, par);
    par.prefCompBiasScale = (float) o.getd("--comp-bias-corr-scale", 0.15);    // StructureSearch.cpp:101
    par.alnCompBiasScale = 0.5f;                                               // StructureSearch.cpp:107
    const int kmerThr = o.geti("--k-score", INT_MAX) != INT_MAX ? o.geti("--k-score", 0) : fshost_kmer_threshold((float) o.getd("-s", 9.5), 6);
    if (!resolveStructureBits(o, o.pos[0], o.pos[1], err)) return fail(err);
    if (!checkMaxSeqLen(o, q3, "query", err) || !checkMaxSeqLen(o, t3, "target", err)) return fail(err);
    const int spaced = o.geti("--spaced-kmer-mode", 1);
    if (prefMode == 0 && (o.geti("-k", 0) != 0 && o.geti("-k", 6) != 6)) return fail("search: only -k 6 is implemented on the device path");
    Matrix m3, mA;
    m3.builtin(FSHOST_MAT_3DI, 2.1f, 0.0f);
    mA.builtin(FSHOST_MAT_BLOSUM62, par.alignmentType == 2 ? 1.4f : 0.0f, 0.0f);
    fshost_matrix *m8 = fshost_matrix_create(FSHOST_MAT_3DI, 8.0f, -0.2f), *m2 = fshost_matrix_create(FSHOST_MAT_3DI, 2.0f, -0.2f);
    if (!m8 || !m2) return fail("matrix construction failed");
    PaddedTarget pt;
    if (!loadPadded(t3, &tA, m3, &mA, pt, err)) return fail(err);
    const double tLoaded = nowSec();
    DeviceSet ds;
    if (!ds.open(o, pt, true, 3, err, true)) { ds.close(); return fail(err); }
    const int maxRes = (int) std::min<uint64_t>((uint64_t) par.maxResListLen, std::max<uint64_t>(t3.size(), 1));
    if (prefMode == 0) {
        fsgpu_kmer_index_params ip;
        ip.kmerSize = 6; ip.spaced = spaced; ip.kmerThr = kmerThr;
        ip.maskLowerCase = o.geti("--mask-lower-case", 1); ip.maskNrepeats = o.geti("--mask-n-repeat", 6);
        for (fsgpu_ctx *c : ds.root)
            if (fsgpu_kmer_index_build(c, &ip, fshost_matrix_scores(m8)) != FSGPU_OK) { err = std::string("GPU: ") + fsgpu_last_error(c); ds.close(); return fail(err); }
    }
    const double tDevice = nowSec();
    fsgpu_kmer_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.maxResListLen = maxRes; sp.minDiagScoreThr = par.minDiagScoreThr;
    sp.kmerScoreOnly = (prefMode == 0 && o.geti("--diag-score", 1) == 0) ? 1 : 0;
    DbWriter w, wp;
    if (!w.open(o.pos[2], DBTYPE_ALIGNMENT_RES, err)) { ds.close(); return fail(err); }
    const bool writePref = o.pos.size() == 4;
    if (writePref && !wp.open(o.pos[3], DBTYPE_PREFILTER_RES, err)) { ds.close(); return fail(err); }
    const int nthreads = ds.threads();
    // k-mer prefilter: capacity of a thread's staging (1024 = the device batch limit; fewer when --max-seqs is so large that 1024 result slabs would
    // not fit 256 MB per thread), a round takes fsgpu_kmer_batch_hint() queries
    const size_t kmerSlab = (size_t) maxRes * (size_t) (1 + std::max(0, par.altAlignment)) * sizeof(fshost_result);
    const size_t batch = prefMode == 0 ? std::max<size_t>(1, std::min<size_t>(1024, ((size_t) 256 << 20) / std::max<size_t>(kmerSlab, 1))) : 16;
    std::vector<std::string> results(q3.size()), prefs(writePref ? q3.size() : 0);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::string firstErr;
    // longest queries first (see fsmod_ungappedprefilter): queries of a batch share scan launches and SW register classes
    const std::vector<size_t> order = lengthOrder(q3);
    auto work = [&](int tix) {
        bool owned = false;
        fsgpu_ctx *ctx = ds.forThread(tix, owned);
        if (!ctx) { bad++; return; }
        fshost_search *s = fshost_search_create(ctx, &par, pt.keys.data(), nullptr, pt.d3, pt.dA, pt.offsets.data(), pt.lengths.data());
        std::vector<std::vector<uint8_t>> cA(batch), c3(batch);
        std::vector<std::vector<int16_t>> thr(batch);
        std::vector<std::vector<int8_t>> prof(batch);
        std::vector<fsgpu_kmer_query> kq(batch);
        std::vector<fsgpu_kmer_hit> khits(prefMode == 0 ? batch * (size_t) maxRes : 1);
        std::vector<uint32_t> kkept(prefMode == 0 ? batch * (size_t) maxRes : 1);
        std::vector<int32_t> nkept(batch), knres(batch);
        std::vector<fshost_result> kres;
        std::vector<fsgpu_hit> ghits((size_t) maxRes);
        std::vector<int32_t> nout(batch), status(batch);
        std::vector<std::vector<uint32_t>> ids(batch);
        std::vector<std::vector<fshost_result>> res(batch);
        std::vector<const uint8_t *> pA(batch), p3(batch);
        std::vector<const uint32_t *> pT(batch);
        std::vector<fshost_result *> pR(batch);
        std::vector<int> Ls(batch), ns(batch), nres(batch);
        std::vector<int64_t> ident(batch);
        std::vector<size_t> qid(batch);
        std::vector<char> line(1024 + 2 * 65536 * 2);
        char pl[128];
        for (;;) {
            size_t take = batch;
            if (prefMode == 0) {
                const size_t seen = next.load();
                const size_t left = q3.size() > seen ? q3.size() - seen : 0;
                const size_t fair = nthreads == 1 ? std::max<size_t>(32, left) : std::max<size_t>(32, left / ((size_t) nthreads * 2));   // several feeders: none takes more than half its share of what is left
                take = std::min<size_t>(std::min<size_t>(batch, (size_t) fsgpu_kmer_batch_hint(ctx)), fair);
            }
            const size_t b0 = next.fetch_add(take);
            if (b0 >= q3.size() || bad) break;
            const size_t nb = std::min(take, q3.size() - b0);
            for (size_t k = 0; k < nb; k++) {
                const size_t id = order[b0 + k];
                qid[k] = id;
                const uint32_t L = q3.seqLen(id);
                const int64_t aid = qA.idOf(q3.key(id));
                if (aid < 0 || qA.seqLen((size_t) aid) != L) { if (!bad++) firstErr = "query AA / 3Di entries do not match"; break; }
                cA[k].resize(L + 1); c3[k].resize(L + 1);
                const char *sA = qA.data((size_t) aid), *s3 = q3.data(id);
                for (uint32_t i = 0; i < L; i++) { cA[k][i] = mA.aa2num[(unsigned char) sA[i]]; c3[k][i] = m3.aa2num[(unsigned char) s3[i]]; }
                ident[k] = (sameDB || includeIdentical) ? t3.idOf(q3.key(id)) : -1;
                Ls[k] = (int) L; pA[k] = cA[k].data(); p3[k] = c3[k].data();
                ids[k].clear();
            }
            if (bad) break;
            // ---- prefilter ----
            double t0 = nowSec();
            if (prefMode == 0) {
                // the whole batch in one library call (fshost_search_kmer_batch: profiles, device prefilter, coverage pre-filter, alignment) -- the same entry
                // point bench.py's all-vs-all leg drives
                std::vector<int64_t> alnIdent(nb);
                for (size_t k = 0; k < nb; k++) alnIdent[k] = (sameDB || includeIdentical) ? (int64_t) qid[k] : -1;   // structurealign compares reader INDICES (structurealign.cpp:359)
                const size_t rcap = (size_t) maxRes * (size_t) (1 + std::max(0, par.altAlignment));
                kres.resize(batch * rcap);
                double sec[4] = {0, 0, 0, 0};
                if (fshost_search_kmer_batch(s, m8, m2, &sp, kmerThr, spaced, (int) nb, pA.data(), p3.data(), Ls.data(), ident.data(), alnIdent.data(), khits.data(), nout.data(), status.data(),
                                             kkept.data(), nkept.data(), kres.data(), knres.data(), sec) != FSGPU_OK) { if (!bad++) firstErr = fshost_search_error(s); break; }
                usPrep += (int64_t) (sec[0] * 1e6); usPref += (int64_t) ((sec[1] + sec[2]) * 1e6); usAlign += (int64_t) (sec[3] * 1e6);
                t0 = nowSec();
                {
                    double st[8];
                    fshost_search_stats(s, st);
                    usAlnPrep += (int64_t) (st[2] * 1e6); usAlnDev += (int64_t) (st[3] * 1e6); usAlnGate += (int64_t) (st[4] * 1e6); usAlnBack += (int64_t) (st[5] * 1e6);
                    nRev += (int64_t) st[7];
                    { int64_t bd = 0, ba = 0; fshost_search_backtrace_counts(s, &bd, &ba); nBtDev += bd; nBtAll += ba; }
                }
                for (size_t k = 0; k < nb && !bad; k++) {
                    if (status[k] < 0) { if (!bad++) firstErr = "query " + std::to_string(q3.key(qid[k])) + ": hit buffers of the reference would overflow"; break; }
                    nPairs += nkept[k];
                    if (writePref) {
                        const uint32_t *kept = kkept.data() + k * (size_t) maxRes;
                        int kp = 0;                                  // the kept ids are a subsequence of the hits' ids
                        for (int h = 0; h < nout[k] && kp < nkept[k]; h++) {
                            const fsgpu_kmer_hit &hit = khits[k * (size_t) maxRes + h];
                            if (hit.id != kept[kp]) continue;
                            kp++;
                            prefs[qid[k]].append(pl, fshost_format_prefilter_hit(pl, pt.keys[hit.id], hit.score, (int) (int16_t) hit.diagonal));
                        }
                    }
                    std::string &out = results[qid[k]];
                    for (int r = 0; r < knres[k]; r++) {
                        const fshost_result *rr = &kres[k * rcap + (size_t) r];
                        out.append(line.data(), fshost_format_result(line.data(), rr, fshost_search_backtrace(s, rr), par.addBacktrace));
                    }
                }
                if (bad) break;
                usFormat += (int64_t) ((nowSec() - t0) * 1e6);
                continue;
            } else {
                std::vector<const uint8_t *> gq; std::vector<int> gL, gn; std::vector<int64_t> gi; std::vector<size_t> gk;
                for (size_t k = 0; k < nb; k++) if (Ls[k] > 0) { gq.push_back(c3[k].data()); gL.push_back(Ls[k]); gi.push_back(ident[k]); gk.push_back(k); }
                gn.resize(gq.size());
                ghits.resize(std::max<size_t>(1, gq.size()) * (size_t) par.maxResListLen);
                if (!gq.empty() && fshost_search_prefilter_batch(s, (int) gq.size(), gq.data(), gL.data(), gi.data(), ghits.data(), gn.data()) != FSGPU_OK) {
                    if (!bad++) firstErr = fshost_search_error(s);
                    break;
                }
                for (size_t j = 0; j < gq.size(); j++) {
                    const size_t k = gk[j];
                    const fsgpu_hit *hh = ghits.data() + j * (size_t) par.maxResListLen;
                    for (int h = 0; h < gn[j]; h++) {
                        ids[k].push_back(hh[h].id);
                        if (writePref) prefs[qid[k]].append(pl, fshost_format_prefilter_hit(pl, pt.keys[hh[h].id], hh[h].score, 0));
                    }
                }
            }
            if (bad) break;
            { const double t1 = nowSec(); usPref += (int64_t) ((t1 - t0) * 1e6); t0 = t1; }
            // ---- align: one multi-query launch for the batch (queries without hits keep an empty entry) ----
            std::vector<size_t> live;
            for (size_t k = 0; k < nb; k++) if (!ids[k].empty() && Ls[k] > 0) live.push_back(k);
            if (live.empty()) continue;
            std::vector<const uint8_t *> lA, l3; std::vector<const uint32_t *> lT; std::vector<fshost_result *> lR;
            std::vector<int> lL, lN, lres(live.size()); std::vector<int64_t> lI;
            for (size_t k : live) {
                res[k].resize(ids[k].size() * (size_t) (1 + par.altAlignment) + 1);
                lA.push_back(pA[k]); l3.push_back(p3[k]); lT.push_back(ids[k].data()); lR.push_back(res[k].data());
                lL.push_back(Ls[k]); lN.push_back((int) ids[k].size());
                // structurealign compares the query's and the target's INDEX in their readers (structurealign.cpp:359)
                lI.push_back((sameDB || includeIdentical) ? (int64_t) qid[k] : -1);
            }
            if (fshost_search_align_batch(s, (int) live.size(), lA.data(), l3.data(), lL.data(), lI.data(), lT.data(), lN.data(), lR.data(), lres.data()) != FSGPU_OK) {
                if (!bad++) firstErr = fshost_search_error(s);
                break;
            }
            { const double t1 = nowSec(); usAlign += (int64_t) ((t1 - t0) * 1e6); t0 = t1; }
            {
                double st[8];
                fshost_search_stats(s, st);
                usAlnPrep += (int64_t) (st[2] * 1e6); usAlnDev += (int64_t) (st[3] * 1e6); usAlnGate += (int64_t) (st[4] * 1e6); usAlnBack += (int64_t) (st[5] * 1e6);
                nRev += (int64_t) st[7];
                { int64_t bd = 0, ba = 0; fshost_search_backtrace_counts(s, &bd, &ba); nBtDev += bd; nBtAll += ba; }
                for (int x : lN) nPairs += x;
            }
            for (size_t j = 0; j < live.size(); j++) {
                std::string &out = results[qid[live[j]]];
                for (int r = 0; r < lres[j]; r++)
                    out.append(line.data(), fshost_format_result(line.data(), &res[live[j]][r], fshost_search_backtrace(s, &res[live[j]][r]), par.addBacktrace));
            }
            usFormat += (int64_t) ((nowSec() - t0) * 1e6);
        }
        fshost_search_free(s);
        if (owned) fsgpu_destroy(ctx);
    };
    std::vector<std::thread> ths;
    for (int i = 1; i < nthreads; i++) ths.emplace_back(work, i);
    work(0);
    for (auto &th : ths) th.join();
    const double tLoop = nowSec();
    fshost_matrix_free(m8); fshost_matrix_free(m2);
    ds.close();
    if (bad) return fail("search failed: " + firstErr);
    for (size_t id = 0; id < q3.size(); id++) {
        w.write(q3.key(id), results[id].data(), results[id].size());
        if (writePref) wp.write(q3.key(id), prefs[id].data(), prefs[id].size());
    }
    if (!w.close(err) || (writePref && !wp.close(err))) return fail(err);
    if (moduleTiming())
        fprintf(stderr, "search timing: load %.2f s, device open + index %.2f s, query loop %.2f s (%d threads; summed over threads: prepare %.2f, prefilter %.2f, align %.2f [profiles %.2f, SW launches + waits %.2f, gates %.2f, "
                "backtrace %.2f (%lld of %lld on the device); %lld pairs, %lld reversed], format %.2f), close + write %.2f s\n", tLoaded - tStart, tDevice - tLoaded, tLoop - tDevice, nthreads, usPrep / 1e6, usPref / 1e6,
                usAlign / 1e6, usAlnPrep / 1e6, usAlnDev / 1e6, usAlnGate / 1e6, usAlnBack / 1e6, (long long) nBtDev.load(), (long long) nBtAll.load(), (long long) nPairs.load(), (long long) nRev.load(), usFormat / 1e6, nowSec() - tLoop);
    return EXIT_SUCCESS;
}

int fsmod_structurealign(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "structurealign", {kAlignFlags, kStructAlignOnlyFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 4) return fail("usage: structurealign <queryDB> <targetDB> <prefDB> <outAlnDB> [-e E] [--alignment-type 0|2] [-a] [--threads T] ...");
    std::string err;
    DbReader qA, q3, tA, t3, pref;
    if (!qA.open(o.pos[0], err) || !q3.open(dbPathWithSuffix(o.pos[0], "_ss"), err) || !tA.open(o.pos[1], err) || !t3.open(dbPathWithSuffix(o.pos[1], "_ss"), err) ||
        !pref.open(o.pos[2], err))
        return fail(err);
    const bool sameDB = o.pos[0] == o.pos[1];
    const bool includeIdentical = o.geti("--add-self-matches", 0) != 0;
    fshost_params par;
    fillParams(o, par);
    par.alnCompBiasScale = (float) o.getd("--comp-bias-corr-scale", 0.5);
    if (!resolveStructureBits(o, o.pos[0], o.pos[1], err)) return fail(err);
    if (!checkMaxSeqLen(o, q3, "query", err) || !checkMaxSeqLen(o, t3, "target", err)) return fail(err);
    Matrix m3, mA;
    m3.builtin(FSHOST_MAT_3DI, 2.1f, 0.0f);
    mA.builtin(FSHOST_MAT_BLOSUM62, par.alignmentType == 2 ? 1.4f : 0.0f, 0.0f);
    PaddedTarget pt;
    if (!loadPadded(t3, &tA, m3, &mA, pt, err)) return fail(err);
    DeviceSet ds;
    if (!ds.open(o, pt, true, 3, err, true)) { ds.close(); return fail(err); }
    DbWriter w;
    if (!w.open(o.pos[3], DBTYPE_ALIGNMENT_RES, err)) { ds.close(); return fail(err); }
    const int nthreads = ds.threads();
    std::vector<std::string> results(pref.size());
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::atomic<int64_t> nBtDev(0), nBtAll(0);
    std::string firstErr;
    // each host thread takes groups of prefilter entries: their hit lists go through ONE multi-query SW launch
    // (fshost_search_align_batch), then gates / backtrace / formatting per query
    const size_t group = (size_t) std::max(1, std::min(o.geti("--align-batch", 64), 64));
    auto work = [&](int tix) {
        bool owned = false;
        fsgpu_ctx *ctx = ds.forThread(tix, owned);
        if (!ctx) { bad++; return; }
        fshost_search *s = fshost_search_create(ctx, &par, pt.keys.data(), nullptr, pt.d3, pt.dA, pt.offsets.data(), pt.lengths.data());
        std::vector<std::vector<uint8_t>> cA(group), c3(group);
        std::vector<std::vector<uint32_t>> ids(group);
        std::vector<std::vector<fshost_result>> res(group);
        std::vector<size_t> entry(group);
        std::vector<const uint8_t *> pA(group), p3(group);
        std::vector<const uint32_t *> pT(group);
        std::vector<fshost_result *> pR(group);
        std::vector<int> Ls(group), ns(group), nres(group);
        std::vector<int64_t> ident(group);
        std::vector<char> line(1024 + 2 * 65536 * 2);
        for (;;) {
            const size_t b0 = next.fetch_add(group);
            if (b0 >= pref.size() || bad) break;
            size_t m = 0;
            for (size_t id = b0; id < std::min(pref.size(), b0 + group) && !bad; id++) {
                const uint32_t queryKey = pref.key(id);
                const char *data = pref.data(id), *const dataEnd = data + pref.entryLen(id);      // an entry that lost its terminator ends at its length
                if (data == dataEnd || *data == '\0') continue;
                const int64_t qid = q3.idOf(queryKey);
                if (qid < 0 || qA.idOf(queryKey) < 0) { if (!bad++) firstErr = "query key missing in query database"; break; }
                const uint32_t L = q3.seqLen((size_t) qid);
                if (qA.seqLen((size_t) qA.idOf(queryKey)) != L) { if (!bad++) firstErr = "query AA / 3Di entries do not match"; break; }
                cA[m].resize(L); c3[m].resize(L);
                const char *sA = qA.data((size_t) qA.idOf(queryKey)), *s3 = q3.data((size_t) qid);
                for (uint32_t i = 0; i < L; i++) { cA[m][i] = mA.aa2num[(unsigned char) sA[i]]; c3[m][i] = m3.aa2num[(unsigned char) s3[i]]; }
                // prefilter entry: lines "targetKey \t score \t diagonal" (Util::parseKey, structurealign.cpp:351-355)
                ids[m].clear();
                while (data < dataEnd && *data != '\0') {
                    const char *lineEnd = data;
                    while (lineEnd < dataEnd && *lineEnd != '\n' && *lineEnd != '\0') lineEnd++;
                    const std::string line(data, lineEnd);                  // a terminated copy: strtoul stops inside the entry
                    const uint32_t dbKey = (uint32_t) strtoul(line.c_str(), nullptr, 10);
                    const int64_t tid = t3.idOf(dbKey);
                    if (tid < 0) { if (!bad++) firstErr = "target key missing in target database"; break; }
                    ids[m].push_back((uint32_t) tid);
                    data = (lineEnd < dataEnd && *lineEnd == '\n') ? lineEnd + 1 : lineEnd;
                }
                if (bad) break;
                res[m].resize(ids[m].size() * (size_t) (1 + par.altAlignment) + 1);
                entry[m] = id; pA[m] = cA[m].data(); p3[m] = c3[m].data(); pT[m] = ids[m].data(); pR[m] = res[m].data();
                Ls[m] = (int) L; ns[m] = (int) ids[m].size();
                ident[m] = (sameDB || includeIdentical) ? qid : -1;      // queryId == targetId, reader indices (structurealign.cpp:359)
                m++;
            }
            if (bad) break;
            if (m == 0) continue;
            if (fshost_search_align_batch(s, (int) m, pA.data(), p3.data(), Ls.data(), ident.data(), pT.data(), ns.data(), pR.data(), nres.data()) != FSGPU_OK) {
                if (!bad++) firstErr = fshost_search_error(s);
                break;
            }
            { int64_t bd = 0, ba = 0; fshost_search_backtrace_counts(s, &bd, &ba); nBtDev += bd; nBtAll += ba; }
            for (size_t k = 0; k < m; k++) {
                std::string &out = results[entry[k]];
                for (int r = 0; r < nres[k]; r++)
                    out.append(line.data(), fshost_format_result(line.data(), &res[k][r], fshost_search_backtrace(s, &res[k][r]), par.addBacktrace));
            }
        }
        fshost_search_free(s);
        if (owned) fsgpu_destroy(ctx);
    };
    std::vector<std::thread> ths;
    for (int i = 1; i < nthreads; i++) ths.emplace_back(work, i);
    work(0);
    for (auto &th : ths) th.join();
    ds.close();
    if (bad) return fail("structurealign failed: " + firstErr);
    for (size_t id = 0; id < pref.size(); id++) w.write(pref.key(id), results[id].data(), results[id].size());
    if (!w.close(err)) return fail(err);
    if (moduleTiming()) fprintf(stderr, "structurealign timing: backtrace 0.00 (%lld of %lld on the device)\n", (long long) nBtDev.load(), (long long) nBtAll.load());
    return EXIT_SUCCESS;
}


// structurerescorediagonal (a.k.a. structureungappedalign): rescoring of kmermatcher / prefilter hits along their diagonal
int fsmod_structurerescorediagonal(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "structurerescorediagonal", {kAlignFlags, kRescoreOnlyFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 4) return fail("usage: structurerescorediagonal <queryDB> <targetDB> <prefDB> <outAlnDB> [-e E] [-c C --cov-mode M] [--alignment-type 0|2] [-a] [--threads T] ...");
    if (o.geti("--alt-ali", 0) != 0) return fail("structurerescorediagonal: --alt-ali is not read by this module");
    std::string err;
    DbReader qA, q3, tA, t3, pref;
    if (!qA.open(o.pos[0], err) || !q3.open(dbPathWithSuffix(o.pos[0], "_ss"), err) || !tA.open(o.pos[1], err) || !t3.open(dbPathWithSuffix(o.pos[1], "_ss"), err) ||
        !pref.open(o.pos[2], err))
        return fail(err);
    const bool sameDB = o.pos[0] == o.pos[1];
    const bool includeIdentical = o.geti("--add-self-matches", 0) != 0;
    fshost_params par;
    fillParams(o, par);
    par.skipUndefinedDiagonals = o.gets("--undefined-diagonals", "fail") == "skip";
    if (!checkMaxSeqLen(o, q3, "query", err) || !checkMaxSeqLen(o, t3, "target", err)) return fail(err);
    Matrix m3, mA;
    m3.builtin(FSHOST_MAT_3DI, 2.1f, 0.0f);
    mA.builtin(FSHOST_MAT_BLOSUM62, par.alignmentType == 2 ? 1.4f : 0.0f, 0.0f);
    PaddedTarget pt;
    if (!loadPadded(t3, &tA, m3, &mA, pt, err)) return fail(err);
    DeviceSet ds;
    if (!ds.open(o, pt, true, 2, err)) { ds.close(); return fail(err); }
    DbWriter w;
    if (!w.open(o.pos[3], DBTYPE_ALIGNMENT_RES, err)) { ds.close(); return fail(err); }
    const int nthreads = ds.threads();
    std::vector<std::string> results(pref.size());
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::string firstErr;
    const size_t group = 2048;       // prefilter entries per device call: a handful of pairs each
    auto work = [&](int tix) {
        bool owned = false;
        fsgpu_ctx *ctx = ds.forThread(tix, owned);
        if (!ctx) { bad++; return; }
        fshost_search *s = fshost_search_create(ctx, &par, pt.keys.data(), nullptr, pt.d3, pt.dA, pt.offsets.data(), pt.lengths.data());
        std::vector<std::vector<uint8_t>> cA(group), c3(group);
        std::vector<std::vector<uint32_t>> ids(group);
        std::vector<std::vector<int16_t>> diags(group);
        std::vector<std::vector<fshost_result>> res(group);
        std::vector<size_t> entry(group);
        std::vector<const uint8_t *> pA(group), p3(group);
        std::vector<const uint32_t *> pT(group);
        std::vector<const int16_t *> pD(group);
        std::vector<fshost_result *> pR(group);
        std::vector<int> Ls(group), ns(group), nres(group);
        std::vector<int64_t> ident(group);
        std::vector<char> line(1024 + 2 * 65536 * 2);
        for (;;) {
            const size_t b0 = next.fetch_add(group);
            if (b0 >= pref.size() || bad) break;
            size_t m = 0;
            for (size_t id = b0; id < std::min(pref.size(), b0 + group) && !bad; id++) {
                const uint32_t queryKey = pref.key(id);
                const char *data = pref.data(id), *const dataEnd = data + pref.entryLen(id);      // an entry that lost its terminator ends at its length
                if (data == dataEnd || *data == '\0') continue;
                const int64_t qid = q3.idOf(queryKey);
                if (qid < 0 || qA.idOf(queryKey) < 0) { if (!bad++) firstErr = "query key missing in query database"; break; }
                const uint32_t L = q3.seqLen((size_t) qid);
                if (qA.seqLen((size_t) qA.idOf(queryKey)) != L) { if (!bad++) firstErr = "query AA / 3Di entries do not match"; break; }
                cA[m].resize(L); c3[m].resize(L);
                const char *sA = qA.data((size_t) qA.idOf(queryKey)), *s3 = q3.data((size_t) qid);
                for (uint32_t i = 0; i < L; i++) { cA[m][i] = mA.aa2num[(unsigned char) sA[i]]; c3[m][i] = m3.aa2num[(unsigned char) s3[i]]; }
                ids[m].clear(); diags[m].clear();
                while (data < dataEnd && *data != '\0') {          // QueryMatcher::parsePrefilterHit: exactly three columns
                    const char *lineEnd = data;
                    while (lineEnd < dataEnd && *lineEnd != '\n' && *lineEnd != '\0') lineEnd++;
                    const std::string line(data, lineEnd);                  // a terminated copy: the number parsers stop inside the entry
                    const char *l0 = line.c_str();
                    char *e1, *e2, *e3;
                    const uint32_t dbKey = (uint32_t) strtoul(l0, &e1, 10);
                    (void) strtol(e1, &e2, 10);
                    const long dg = strtol(e2, &e3, 10);
                    if (e1 == l0 || e2 == e1 || e3 == e2) { if (!bad++) firstErr = "Invalid prefilter input"; break; }
                    const int64_t tid = t3.idOf(dbKey);
                    if (tid < 0) { if (!bad++) firstErr = "target key missing in target database"; break; }
                    ids[m].push_back((uint32_t) tid);
                    diags[m].push_back((int16_t) dg);
                    data = (lineEnd < dataEnd && *lineEnd == '\n') ? lineEnd + 1 : lineEnd;
                }
                if (bad) break;
                if (L == 0 || ids[m].empty()) continue;
                res[m].resize(ids[m].size() + 1);
                entry[m] = id; pA[m] = cA[m].data(); p3[m] = c3[m].data(); pT[m] = ids[m].data(); pD[m] = diags[m].data(); pR[m] = res[m].data();
                Ls[m] = (int) L; ns[m] = (int) ids[m].size();
                ident[m] = (sameDB || includeIdentical) ? qid : -1;      // queryId == targetId (structurerescorediagonal.cpp:310)
                m++;
            }
            if (bad) break;
            if (m == 0) continue;
            if (fshost_search_rescore_diagonal_batch(s, (int) m, pA.data(), p3.data(), Ls.data(), ident.data(), pT.data(), pD.data(), ns.data(), pR.data(), nres.data()) != FSGPU_OK) {
                if (!bad++) firstErr = fshost_search_error(s);
                break;
            }
            for (size_t k = 0; k < m; k++) {
                std::string &out = results[entry[k]];
                for (int r = 0; r < nres[k]; r++) {
                    const size_t need = 1024 + (size_t) res[k][r].backtraceLen + 64;
                    if (line.size() < need) line.resize(need);
                    out.append(line.data(), fshost_format_result(line.data(), &res[k][r], fshost_search_backtrace(s, &res[k][r]), par.addBacktrace));
                }
            }
        }
        fshost_search_free(s);
        if (owned) fsgpu_destroy(ctx);
    };
    std::vector<std::thread> ths;
    for (int i = 1; i < nthreads; i++) ths.emplace_back(work, i);
    work(0);
    for (auto &th : ths) th.join();
    ds.close();
    if (bad) return fail("structurerescorediagonal failed: " + firstErr);
    for (size_t id = 0; id < pref.size(); id++) w.write(pref.key(id), results[id].data(), results[id].size());
    if (!w.close(err)) return fail(err);
    return EXIT_SUCCESS;
}


// ---- indexdb / createindex: the precomputed index the search workflow hands to its modules ----------------
// File format = PrefilteringIndexReader::createIndexFile (M/src/prefiltering/PrefilteringIndexReader.cpp:53-307): an MMseqs DB of type
// DBTYPE_INDEX_DB (9) whose entries are keyed by the constants below, every entry padded to the page size.  The k-mer table (ENTRIES /
// ENTRIESOFFSETS), the sequence lookup and the extended 3-mer matrix are what IndexBuilder::fillDatabase / ExtendedSubstitutionMatrix
// leave in memory; here they come from the index this library builds ON THE DEVICE (fsgpu_kmer_index_build), renumbered from the
// device's first-3-mer-major k-mer order to the reference's (first3 + 8000 * last3).  Consumers: the reference's CPU prefilter /
// structurealign (`prefilter q db_ss.idx`), and this repository's modules, which read the sequence DBs out of it (DbReader::open).
namespace {
enum { IDX_VERSION = 0, IDX_META = 1, IDX_SCOREMATRIXNAME = 2, IDX_SCOREMATRIX2MER = 3, IDX_SCOREMATRIX3MER = 4, IDX_DBR1INDEX = 5, IDX_DBR1DATA = 6,
       IDX_DBR2INDEX = 7, IDX_DBR2DATA = 8, IDX_ENTRIES = 9, IDX_ENTRIESOFFSETS = 10, IDX_ENTRIESNUM = 12, IDX_SEQCOUNT = 13, IDX_SEQINDEXDATA = 14,
       IDX_SEQINDEXDATASIZE = 15, IDX_SEQINDEXSEQOFFSET = 16, IDX_HDR1INDEX = 18, IDX_HDR1DATA = 19, IDX_HDR2INDEX = 20, IDX_HDR2DATA = 21,
       IDX_GENERATOR = 22, IDX_SPACEDPATTERN = 23 };
enum { IDX_SUBSET_NO_HEADERS = 1, IDX_SUBSET_NO_PREFILTER = 2, IDX_SUBSET_NO_ALIGNMENT = 4, IDX_SUBSET_NO_SEQUENCE_LOOKUP = 8 };
const int DBTYPE_INDEX_DB = 9;

// DBWriter in the mode createIndexFile uses it: one data file, entry = bytes + '\0', then zero padding to the next page
// (DBWriter::writeData / alignToPageSize, DBWriter.cpp:412-443), index sorted by key at close
struct IdxWriter {
    FILE *f = nullptr;
    uint64_t off = 0;
    bool failed = false;
    std::vector<DbReader::Entry> entries;
    void raw(const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) failed = true; off += n; }
    uint64_t begin() const { return off; }
    void end(uint32_t key, uint64_t start) {
        const char z = 0;
        raw(&z, 1);
        if (off - start > 0xffffffffull) failed = true;                     // the index line carries a 32-bit length (DBWriter.cpp:489)
        entries.push_back({key, start, (uint32_t) (off - start)});
        static const char zeros[4096] = {0};
        raw(zeros, (size_t) ((4096 - (off & 4095)) & 4095));
    }
    void put(uint32_t key, const void *p, size_t n) { const uint64_t s0 = begin(); raw(p, n); end(key, s0); }
    void alias(uint32_t key, uint64_t o, uint64_t len) { entries.push_back({key, o, (uint32_t) len}); }
};

// DBReader::serialize (DBReader.cpp:812-840): size, dataSize (sum of the entry lengths), lastKey, dbtype, maxSeqLen (largest entry length),
// then the 24-byte index records {u32 key, u64 offset, u32 length} with their struct padding (zeroed here, uninitialised in the reference)
void serializeReader(const DbReader &r, std::vector<char> &out) {
    const uint64_t n = r.size();
    uint64_t dataSize = 0;
    uint32_t lastKey = 0, maxLen = 0;
    for (size_t i = 0; i < n; i++) { dataSize += r.entryLen(i); lastKey = std::max(lastKey, r.key(i)); maxLen = std::max(maxLen, r.entryLen(i)); }
    out.assign(28 + 24 * n, 0);
    char *p = out.data();
    memcpy(p, &n, 8); memcpy(p + 8, &dataSize, 8); memcpy(p + 16, &lastKey, 4);
    const int32_t type = r.rawDbtype();
    memcpy(p + 20, &type, 4); memcpy(p + 24, &maxLen, 4);
    p += 28;
    for (size_t i = 0; i < n; i++, p += 24) {
        const uint32_t key = r.key(i), len = r.entryLen(i);
        const uint64_t o = r.offset(i);
        memcpy(p, &key, 4); memcpy(p + 8, &o, 8); memcpy(p + 16, &len, 4);
    }
}

// Masker::maskSequence with tantan off (Masker.cpp:15-56): runs of more than `nRepeats` equal codes -> X, then lower-case letters -> X
void maskedCodes(const Matrix &m, const char *letters, int L, int nRepeats, bool lowerCase, uint8_t *out) {
    const uint8_t X = m.aa2num[(unsigned char) 'X'];
    for (int i = 0; i < L; i++) out[i] = m.aa2num[(unsigned char) letters[i]];
    if (nRepeats > 0) {
        int run = 0, start = 0;
        for (int i = 0; i <= L; i++) {
            if (i < L && i > 0 && out[i] == out[i - 1]) { run++; continue; }
            if (i > 0 && run > nRepeats) for (int k = start; k < i; k++) out[k] = X;
            run = 1; start = i;
        }
    }
    if (lowerCase) for (int i = 0; i < L; i++) if (islower((unsigned char) letters[i])) out[i] = X;
}

// ExtendedSubstitutionMatrix::calcScoreMatrix for k = 2 (ExtendedSubstitutionMatrix.cpp:20-69): row of 2-mer a = all 400 2-mers sorted by score
// descending, ties in the permutation order (first position most significant), rows padded to a multiple of 64 (+64) with (-255, 0)
void scoreMatrix2mer(const int16_t *sub /*21x21*/, std::vector<int16_t> &score, std::vector<uint32_t> &index, size_t &rowSize) {
    const size_t size = 400;
    rowSize = (size / 64 + 1) * 64;
    score.assign(size * rowSize, (int16_t) -255); index.assign(size * rowSize, 0);
    std::vector<std::pair<int16_t, uint32_t>> tmp(size);
    for (int a1 = 0; a1 < 20; a1++)
        for (int a0 = 0; a0 < 20; a0++) {
            const size_t row = (size_t) a0 + 20 * (size_t) a1;
            for (int b0 = 0; b0 < 20; b0++)
                for (int b1 = 0; b1 < 20; b1++)
                    tmp[(size_t) b0 * 20 + b1] = {(int16_t) (sub[a0 * 21 + b0] + sub[a1 * 21 + b1]), (uint32_t) (b0 + 20 * b1)};
            std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int16_t, uint32_t> &x, const std::pair<int16_t, uint32_t> &y) { return x.first > y.first; });
            for (size_t z = 0; z < size; z++) { score[row * rowSize + z] = tmp[z].first; index[row * rowSize + z] = tmp[z].second; }
        }
}
} // namespace

int fsmod_indexdb(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "indexdb", {kIndexdbFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 2) return fail("usage: indexdb <sequenceDB> <sequenceDB> [--index-subset N] [--index-dbsuffix S] [-s S] [--k-score T] [--mask-lower-case 0|1] "
                                       "[--mask-n-repeat N] [--spaced-kmer-mode 0|1] [--comp-bias-corr 0|1] [--max-seq-len N]");
    if (o.pos[0] != o.pos[1]) return fail("indexdb: a separate source database (<db1> != <db2>) is not implemented");
    const std::string db = o.pos[0];
    std::string err;
    // profile / cluster databases (<db>_aln or <db>_clu next to <db>_seq, indexdb.cpp:49-75) index a different pair of readers: not covered
    {
        std::string base = db;
        const std::string suffix = o.gets("--index-dbsuffix", "");
        if (!suffix.empty()) { const size_t pos = base.find(suffix); if (pos != std::string::npos) base = base.substr(0, pos); }
        auto exists = [](const std::string &f) { struct stat st; return stat(f.c_str(), &st) == 0; };
        if ((exists(db + "_aln.dbtype") || exists(db + "_clu.dbtype")) && exists(base + "_seq" + suffix + ".dbtype"))
            return fail("indexdb: cluster / profile databases (<db>_aln | <db>_clu with <db>_seq) are not implemented");
    }
    DbReader r;
    if (!r.open(db, err)) return fail(err);
    if (r.dbtype() != DBTYPE_AMINO_ACIDS) return fail("indexdb: only amino-acid typed sequence databases (AA or 3Di letters) are implemented");
    if ((r.extended() & DBTYPE_EXTENDED_GPU) != 0) return fail("indexdb: padded GPU databases carry no k-mer index in the reference either");
    if (!checkMaxSeqLen(o, r, "database", err)) return fail(err);
    const int subset = o.geti("--index-subset", 0);
    const bool needKmerIndex = (subset & IDX_SUBSET_NO_PREFILTER) == 0;
    const bool needLookup = (subset & IDX_SUBSET_NO_SEQUENCE_LOOKUP) == 0;
    const bool noHeaders = (subset & IDX_SUBSET_NO_HEADERS) != 0;
    const int kmerSize = needKmerIndex ? 6 : 0;                                  // indexdb.cpp:128-135: no k-mer index -> k = 0, score 0
    if (needKmerIndex && r.residues() >= 3350000000ull) return fail("indexdb: database needs k = 7, which is not implemented on the device path");
    const float sens = (float) o.getd("-s", 5.7);                               // setIndexDbDefaults (indexdb.cpp:13-15)
    const int kmerThr = !needKmerIndex ? 0 : (o.geti("--k-score", INT_MAX) != INT_MAX ? o.geti("--k-score", 0) : fshost_kmer_threshold(sens, 6));
    const int spaced = o.geti("--spaced-kmer-mode", 1);
    const int maskLower = o.geti("--mask-lower-case", 0), maskNrepeats = o.geti("--mask-n-repeat", 0);
    fshost_matrix *m8 = fshost_matrix_create(FSHOST_MAT_3DI, 8.0f, -0.2f);
    Matrix m3;
    if (!m8 || !m3.builtin(FSHOST_MAT_3DI, 8.0f, -0.2f)) return fail("matrix construction failed");
    DbReader hdr;
    if (!noHeaders && !hdr.open(db + "_h", err)) return fail("Database " + db + " needs header information (" + err + ")");

    const std::string out = db + ".idx";
    for (const char *ext : {"", ".index", ".dbtype"}) remove((out + ext).c_str());
    IdxWriter w;
    w.f = fopen(out.c_str(), "wb");
    if (!w.f) return fail("cannot write " + out);
    w.put(IDX_VERSION, "fs1", 3);                                                // index_version_compatible of the Foldseek binary (F/src/foldseek.cpp:11)
    {
        const int32_t meta[12] = {o.geti("--max-seq-len", 65535), kmerSize, o.geti("--comp-bias-corr", 1) ? 1 : 0, 21, 0 /* --mask 0 */, spaced ? 1 : 0, kmerThr,
                                  r.rawDbtype(), r.rawDbtype(), noHeaders ? 0 : 1, 0, 1};
        w.put(IDX_META, meta, sizeof(meta));
    }
    {
        size_t len = 0;
        const char *text = fshost_matrix_text(FSHOST_MAT_3DI, &len);
        std::string sm = std::string("3di.out:") + std::string(text, len);       // BaseMatrix::serialize: name ':' file text
        w.put(IDX_SCOREMATRIXNAME, sm.data(), sm.size());
    }
    w.put(IDX_SPACEDPATTERN, "", 0);                                             // written exactly when the user pattern is EMPTY (:103, sic)
    {
        const std::string gen = "fsgpu-modules indexdb (foldseek_amd)";
        w.put(IDX_GENERATOR, gen.data(), gen.size());
    }
    std::vector<char> ser;
    auto putReader = [&](const DbReader &x, uint32_t keyIndex, uint32_t keyData, uint32_t aliasIndex, uint32_t aliasData) {
        serializeReader(x, ser);
        const uint64_t oi = w.begin();
        w.put(keyIndex, ser.data(), ser.size());
        const uint64_t od = w.begin();
        w.put(keyData, x.dataBase(), x.dataSize());
        w.alias(aliasIndex, oi, ser.size() + 1);                                 // dbr2 == NULL: the second reader is the first (:128-131)
        w.alias(aliasData, od, x.dataSize() + 1);
    };
    putReader(r, IDX_DBR1INDEX, IDX_DBR1DATA, IDX_DBR2INDEX, IDX_DBR2DATA);
    if (!noHeaders) putReader(hdr, IDX_HDR1INDEX, IDX_HDR1DATA, IDX_HDR2INDEX, IDX_HDR2DATA);

    // the sequence lookup: numeric codes with the masking baked in (IndexBuilder.cpp:134-160), offsets = running sequence lengths
    const size_t n = r.size();
    std::vector<uint64_t> seqOff(n + 1, 0);
    for (size_t i = 0; i < n; i++) seqOff[i + 1] = seqOff[i] + r.seqLen(i);
    std::vector<uint8_t> lookup(seqOff[n] + 1, 0);
    parallelRanges(n, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) maskedCodes(m3, r.data(i), (int) r.seqLen(i), maskNrepeats, maskLower != 0, lookup.data() + seqOff[i]);
    });

    if (needKmerIndex) {
        PaddedTarget pt;
        if (!loadPadded(r, nullptr, m3, nullptr, pt, err)) { fclose(w.f); remove(out.c_str()); return fail(err); }
        DeviceSet ds;
        if (!ds.open(o, pt, false, 1, err)) { ds.close(); fclose(w.f); remove(out.c_str()); return fail(err); }
        fsgpu_ctx *ctx = ds.root[0];
        fsgpu_kmer_index_params ip;
        ip.kmerSize = 6; ip.spaced = spaced; ip.kmerThr = kmerThr; ip.maskLowerCase = maskLower; ip.maskNrepeats = maskNrepeats;
        auto gpuFail = [&](const std::string &what) { const std::string m = what + ": " + fsgpu_last_error(ctx); ds.close(); fclose(w.f); remove(out.c_str()); return fail(m); };
        if (fsgpu_kmer_index_build(ctx, &ip, fshost_matrix_scores(m8)) != FSGPU_OK) return gpuFail("GPU k-mer index build");
        // SCOREMATRIX3MER (:220-228): 8000 rows of 8064 (score int16 | index uint32), the rows this library sorts on the device (k_kmer_rows3)
        {
            const size_t size = 8000, rowSize = (size / 64 + 1) * 64;
            std::vector<int16_t> sc(size * rowSize, (int16_t) -255);
            std::vector<uint32_t> ix(size * rowSize, 0);
            std::vector<uint16_t> row16(size);
            for (size_t a = 0; a < size; a++) {
                if (fsgpu_kmer_row_copy(ctx, (int) a, sc.data() + a * rowSize, row16.data()) != FSGPU_OK) return gpuFail("GPU 3-mer rows");
                for (size_t z = 0; z < size; z++) ix[a * rowSize + z] = row16[z];
            }
            const uint64_t s0 = w.begin();
            w.raw(sc.data(), sc.size() * sizeof(int16_t)); w.raw(ix.data(), ix.size() * sizeof(uint32_t));
            w.end(IDX_SCOREMATRIX3MER, s0);
        }
        {
            std::vector<int16_t> sc; std::vector<uint32_t> ix; size_t rowSize = 0;
            scoreMatrix2mer(fshost_matrix_scores(m8), sc, ix, rowSize);
            const uint64_t s0 = w.begin();
            w.raw(sc.data(), sc.size() * sizeof(int16_t)); w.raw(ix.data(), ix.size() * sizeof(uint32_t));
            w.end(IDX_SCOREMATRIX2MER, s0);
        }
        // ENTRIES / ENTRIESOFFSETS / ENTRIESNUM (:261-283): lists of {uint32 seqId, uint16 first position} per k-mer, sorted by (seqId, position)
        const uint64_t table = 64000000ull, nE = fsgpu_kmer_index_entries(ctx);
        std::vector<uint32_t> devOff(table + 1);
        std::vector<uint64_t> devEnt(std::max<uint64_t>(nE, 1));
        std::vector<uint8_t> devMasked(std::max<uint64_t>(pt.bytes, 1));
        if (fsgpu_kmer_index_copy(ctx, devOff.data(), devEnt.data(), devMasked.data()) != FSGPU_OK) return gpuFail("GPU k-mer index copy");
        ds.close();
        // the lookup written below is the host's; the device masked the same letters with its own kernel: they must agree
        for (size_t i = 0; i < n; i++)
            if (memcmp(devMasked.data() + pt.offsets[i], lookup.data() + seqOff[i], r.seqLen(i)) != 0) { fclose(w.f); remove(out.c_str()); return fail("indexdb: internal error: device and host masking differ at entry " + std::to_string(r.key(i))); }
        std::vector<uint64_t> refOff(table + 1, 0);
        parallelRanges(8000, [&](size_t l0, size_t l1) {                          // reference k-mer number = first3 + 8000 * last3, device = first3 * 8000 + last3
            for (size_t last3 = l0; last3 < l1; last3++)
                for (size_t first3 = 0; first3 < 8000; first3++) { const size_t p = first3 * 8000 + last3; refOff[first3 + 8000 * last3 + 1] = devOff[p + 1] - devOff[p]; }
        });
        for (uint64_t k = 0; k < table; k++) refOff[k + 1] += refOff[k];
        if (refOff[table] != nE) { fclose(w.f); remove(out.c_str()); return fail("indexdb: internal error: k-mer table sizes differ"); }
        std::vector<uint8_t> ent((size_t) nE * 6 + 1);
        parallelRanges(8000, [&](size_t l0, size_t l1) {
            for (size_t last3 = l0; last3 < l1; last3++)
                for (size_t first3 = 0; first3 < 8000; first3++) {
                    const size_t p = first3 * 8000 + last3;
                    uint8_t *dst = ent.data() + refOff[first3 + 8000 * last3] * 6;
                    for (uint32_t e = devOff[p]; e < devOff[p + 1]; e++, dst += 6) {
                        const uint32_t seqId = (uint32_t) (devEnt[e] >> 16);
                        const uint16_t pos = (uint16_t) (devEnt[e] & 0xffffu);
                        memcpy(dst, &seqId, 4); memcpy(dst + 4, &pos, 2);           // IndexEntryLocal, packed (IndexTable.h:25-41)
                    }
                }
        });
        if ((uint64_t) nE * 6 + 1 > 0xffffffffull || (table + 1) * sizeof(uint64_t) + 1 > 0xffffffffull) {
            fclose(w.f); remove(out.c_str()); return fail("indexdb: the k-mer table of this database exceeds the 32-bit entry length of the index DB format (DBWriter.cpp:489); split the database");
        }
        w.put(IDX_ENTRIES, ent.data(), (size_t) nE * 6);
        w.put(IDX_ENTRIESOFFSETS, refOff.data(), (table + 1) * sizeof(uint64_t));
        w.put(IDX_ENTRIESNUM, &nE, sizeof(nE));
    }
    if (needLookup) {
        const uint64_t cnt = n;
        const int64_t dataSize = (int64_t) seqOff[n];
        w.put(IDX_SEQCOUNT, &cnt, sizeof(cnt));
        w.put(IDX_SEQINDEXDATASIZE, &dataSize, sizeof(dataSize));
        w.put(IDX_SEQINDEXSEQOFFSET, seqOff.data(), (n + 1) * sizeof(uint64_t));
        w.put(IDX_SEQINDEXDATA, lookup.data(), seqOff[n] + 1);                   // dataSize + 1 bytes (:300; the last one is uninitialised in the reference)
    }
    fshost_matrix_free(m8);
    if (fclose(w.f) != 0 || w.failed) return fail("write error on " + out);
    std::sort(w.entries.begin(), w.entries.end(), [](const DbReader::Entry &a, const DbReader::Entry &b) { return a.key < b.key; });
    FILE *fi = fopen((out + ".index").c_str(), "w");
    if (!fi) return fail("cannot write " + out + ".index");
    for (const DbReader::Entry &e : w.entries) fprintf(fi, "%u\t%llu\t%u\n", e.key, (unsigned long long) e.offset, e.length);
    if (fclose(fi) != 0) return fail("write error on " + out + ".index");
    FILE *ft = fopen((out + ".dbtype").c_str(), "wb");
    const int32_t t = DBTYPE_INDEX_DB;
    if (!ft || fwrite(&t, 4, 1, ft) != 1 || fclose(ft) != 0) return fail("cannot write " + out + ".dbtype");
    return EXIT_SUCCESS;
}

// createindex <db> <tmpDir>: F/data/structureindex.sh -- the AA database without k-mer table (--index-subset 2), header links for <db>_ss,
// the 3Di database with the k-mer table (--index-subset 5, --index-dbsuffix _ss), then the C-alpha database appended to <db>.idx under the
// keys 500 / 501 (appenddbtoindex, M/src/util/appenddbtoindex.cpp:9-150) when <db>_ca exists.
int fsmod_createindex(int argc, const char **argv) {
    std::vector<std::string> pos, rest;
    for (int i = 0; i < argc; i++) {
        const std::string a = argv[i];
        if (a.size() > 1 && a[0] == '-' && !isdigit((unsigned char) a[1])) {
            if (a == "--index-subset" || a == "--index-dbsuffix" || a == "--index-exclude" || a == "--remove-tmp-files") { i++; continue; }   // set per call below / workflow housekeeping
            rest.push_back(a);
            if (i + 1 < argc) rest.push_back(argv[++i]);
        } else pos.push_back(a);
    }
    if (pos.size() != 2) return fail("usage: createindex <sequenceDB> <tmpDir> [indexdb options]");
    const std::string db = pos[0];
    auto call = [&](const std::string &d, const char *subset, const char *suffix) {
        std::vector<const char *> av = {d.c_str(), d.c_str()};
        for (const std::string &x : rest) av.push_back(x.c_str());
        av.push_back("--index-subset"); av.push_back(subset);
        if (suffix) { av.push_back("--index-dbsuffix"); av.push_back(suffix); }
        return fsmod_indexdb((int) av.size(), av.data());
    };
    if (call(db, "2", nullptr) != EXIT_SUCCESS) return EXIT_FAILURE;
    struct stat st;
    if (stat((db + "_ss_h.dbtype").c_str(), &st) != 0)                           // lndb <db>_h <db>_ss_h
        for (const char *ext : {"", ".index", ".dbtype"}) {
            const std::string from = db + "_h" + ext, to = db + "_ss_h" + ext;
            const size_t slash = from.rfind('/');
            if (symlink((slash == std::string::npos ? from : from.substr(slash + 1)).c_str(), to.c_str()) != 0 && errno != EEXIST) return fail("cannot link " + to);
        }
    if (call(db + "_ss", "5", "_ss") != EXIT_SUCCESS) return EXIT_FAILURE;
    if (stat((db + "_ca.dbtype").c_str(), &st) == 0) {
        DbReader idx, ca;
        std::string err;
        if (!ca.open(db + "_ca", err)) return fail(err);
        // append: serialised reader under key 500, the data under 501, offsets continue at the end of <db>.idx
        FILE *f = fopen((db + ".idx").c_str(), "ab");
        FILE *fi = fopen((db + ".idx.index").c_str(), "a");
        if (!f || !fi) return fail("cannot append to " + db + ".idx");
        fseek(f, 0, SEEK_END);
        uint64_t off = (uint64_t) ftell(f);
        std::vector<char> ser;
        serializeReader(ca, ser);
        const char z = 0;
        bool ok = fwrite(ser.data(), 1, ser.size(), f) == ser.size() && fwrite(&z, 1, 1, f) == 1;
        fprintf(fi, "%u\t%llu\t%llu\n", 500u, (unsigned long long) off, (unsigned long long) ser.size() + 1);
        off += ser.size() + 1;
        ok = ok && fwrite(ca.dataBase(), 1, ca.dataSize(), f) == ca.dataSize() && fwrite(&z, 1, 1, f) == 1;
        fprintf(fi, "%u\t%llu\t%llu\n", 501u, (unsigned long long) off, (unsigned long long) ca.dataSize() + 1);
        if (fclose(f) != 0 || fclose(fi) != 0 || !ok) return fail("write error on " + db + ".idx");
    }
    return EXIT_SUCCESS;
}

// gpuserver: load the target DB once, then answer gapless scans until SIGINT/SIGTERM (gpuserver.cpp:24-101).  The scan
// is this library's (scores capped at 255 - bias like the CPU path, result order = score desc, id asc); the cap is
// recovered from the profile the client sends (profile[a][i] - mat[a][q_i] = rounded composition bias of position i).
int fsmod_gpuserver(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "gpuserver", {kServerFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 1) return fail("usage: gpuserver <targetDB_ss[_pad]> [--max-seqs N] [--max-seq-len L] [--gpu-device D] [--gpu-server-version V | --shm-name NAME]");
    std::string err;
    DbReader t;
    if (!t.open(o.pos[0], err)) return fail(err);
    Matrix m3;
    if (!m3.builtin(FSHOST_MAT_3DI, 2.0f, 0.0f)) return fail("matrix construction failed");
    PaddedTarget pt;
    if (!loadPadded(t, nullptr, m3, nullptr, pt, err)) return fail(err);
    fsgpu_ctx *ctx = nullptr;
    if (fsgpu_create(o.geti("--gpu-device", 0), &ctx) != FSGPU_OK) return fail(std::string("GPU: ") + fsgpu_last_error(nullptr));
    if (fsgpu_db_load(ctx, pt.d3, nullptr, pt.offsets.data(), pt.lengths.data(), pt.lengths.size(), pt.bytes) != FSGPU_OK)
        return fail(std::string("GPU: ") + fsgpu_last_error(ctx));
    const unsigned int maxSeqLen = (unsigned int) std::max(1, std::min(o.geti("--max-seq-len", 65535), (int) FSGPU_MAX_SEQ_LEN));
    const unsigned int maxRes = (unsigned int) std::max(1, o.geti("--max-seqs", 1000));
    installSignalHandlers();
    const std::string name = shmNameFor(o, o.pos[0]);
    GpuShm *shm = gpuShmCreate(name, maxSeqLen, maxRes, err);
    if (!shm) { fsgpu_destroy(ctx); return fail(err); }
    fprintf(stderr, "%s\n", name.c_str());
    int matMin = 0;
    for (int i = 0; i < m3.n * m3.n; i++) matMin = std::min(matMin, (int) m3.tiny[i]);
    std::vector<fsgpu_hit> hits(maxRes);
    int rc = EXIT_SUCCESS;
    while (gKeepRunning) {
        if (shm->state.load(std::memory_order_acquire) != GpuShm::READY) { std::this_thread::yield(); continue; }
        std::atomic_thread_fence(std::memory_order_acquire);
        const int L = (int) shm->queryLen;
        int nout = 0;
        if (L > 0 && (unsigned int) L <= maxSeqLen) {
            const int8_t *qcodes = shm->query(), *prof = shm->profile();
            int cbMin = 0;
            for (int i = 0; i < L; i++) {
                const int c = qcodes[i];
                if (c < 0 || c >= m3.n) { cbMin = 0; nout = -1; break; }
                cbMin = std::min(cbMin, (int) prof[i] - (int) m3.sub[c]);      // row a = 0: prof[0*L + i] - mat[0][q_i]
            }
            if (nout == 0) {
                const int cap = 255 - (std::abs(matMin) + std::abs(cbMin));
                if (fsgpu_gapless_scan(ctx, prof, L, cap, -1, -1, (int) maxRes, hits.data(), &nout) != FSGPU_OK) {
                    fprintf(stderr, "gpuserver: %s\n", fsgpu_last_error(ctx));
                    nout = 0; rc = EXIT_FAILURE; gKeepRunning = 0;
                }
            }
            nout = std::max(nout, 0);
        }
        GpuShmResult *res = shm->results();
        for (int k = 0; k < nout; k++) { res[k].id = hits[k].id; res[k].score = hits[k].score; res[k].qEndPos = 0; res[k].dbEndPos = 0; }
        shm->resultLen = (unsigned int) nout;
        std::atomic_thread_fence(std::memory_order_release);
        shm->state.store(GpuShm::DONE, std::memory_order_release);
    }
    shm->serverExit.store(true, std::memory_order_release);
    std::atomic_thread_fence(std::memory_order_release);
    gpuShmDestroy(shm, name);
    fsgpu_destroy(ctx);
    return rc;
}

} // extern "C"

// ---- convertalis -------------------------------------------------------------------------------------------------
// F/src/strucclustutils/structureconvertalis.cpp:253-1445 for the BLAST-tab family (--format-mode 0, 2, 4) and every
// --format-output column that is a function of the alignment record, the sequences and the headers.  Columns computed from
// C-alpha coordinates (lddt, lddtfull, alntmscore, qtmscore, ttmscore, rmsd, u, t, qca, tca), taxonomy and multimer
// columns are refused by name.  `prob` only reads the score (CalcProbTP.h), so it is answered without the _ca DB the reference
// insists on opening for it.
namespace {

enum ConvCol { C_QUERY, C_TARGET, C_QKEY, C_TKEY, C_EVALUE, C_GAPOPEN, C_PIDENT, C_FIDENT, C_NIDENT, C_QSTART, C_QEND, C_QLEN, C_TSTART, C_TEND,
               C_TLEN, C_ALNLEN, C_BITS, C_CIGAR, C_QSEQ, C_TSEQ, C_Q3DI, C_T3DI, C_QHEADER, C_THEADER, C_QALN, C_TALN, C_Q3DIALN, C_T3DIALN,
               C_MISMATCH, C_QCOV, C_TCOV, C_EMPTY, C_PROB, C_QSET, C_QSETID, C_TSET, C_TSETID };

struct ConvColSpec { const char *name; ConvCol col; bool needSeq, need3Di, needBt; int needSets = 0; };   // needSets: 1 = .lookup, 2 = .source
const ConvColSpec kConvCols[] = {        // LocalParameters::getOutputFormat (LocalParameters.cpp:464-553)
    {"query", C_QUERY, false, false, false}, {"target", C_TARGET, false, false, false}, {"qkey", C_QKEY, false, false, false}, {"tkey", C_TKEY, false, false, false},
    {"evalue", C_EVALUE, false, false, false}, {"gapopen", C_GAPOPEN, false, false, false}, {"pident", C_PIDENT, false, false, false},
    {"fident", C_FIDENT, false, false, false}, {"nident", C_NIDENT, false, false, false}, {"qstart", C_QSTART, false, false, false},
    {"qend", C_QEND, false, false, false}, {"qlen", C_QLEN, false, false, false}, {"tstart", C_TSTART, false, false, false}, {"tend", C_TEND, false, false, false},
    {"tlen", C_TLEN, false, false, false}, {"alnlen", C_ALNLEN, false, false, false}, {"bits", C_BITS, false, false, false}, {"cigar", C_CIGAR, false, false, true},
    {"qseq", C_QSEQ, true, false, false}, {"tseq", C_TSEQ, true, false, false}, {"q3di", C_Q3DI, false, true, false}, {"t3di", C_T3DI, false, true, false},
    {"qheader", C_QHEADER, false, false, false}, {"theader", C_THEADER, false, false, false}, {"qaln", C_QALN, true, false, true}, {"taln", C_TALN, true, false, true},
    {"q3dialn", C_Q3DIALN, false, true, true}, {"t3dialn", C_T3DIALN, false, true, true}, {"mismatch", C_MISMATCH, false, false, false},
    {"qcov", C_QCOV, false, false, false}, {"tcov", C_TCOV, false, false, false}, {"empty", C_EMPTY, false, false, false}, {"prob", C_PROB, false, false, false},
    // set columns: `tset` asks for the lookup only (LocalParameters.cpp:519), so on its own it prints the empty source name -- reproduced
    {"qset", C_QSET, false, false, false, 3}, {"qsetid", C_QSETID, false, false, false, 3}, {"tset", C_TSET, false, false, false, 1},
    {"tsetid", C_TSETID, false, false, false, 3}};
const char *const kConvRefused[] = {"qca", "tca", "u", "t", "alntmscore", "qtmscore", "ttmscore", "rmsd", "lddt", "lddtfull",
                                    "taxid", "taxname", "taxlineage", "complexqtmscore", "multimerqtmscore", "complexttmscore", "multimerttmscore",
                                    "complexassignid", "multimerassignid", "complexu", "multimeru", "complext", "multimert", "qcomplexcoverage",
                                    "qmultimercoverage", "tcomplexcoverage", "tmultimercoverage", "qchaintms", "tchaintms", "qchains", "tchains", "interfacelddt"};

// one line of an alignment DB (Matcher::parseAlignmentRecord, Matcher.cpp:205-282; 10 columns, 11 with the run-length backtrace)
struct AlnRecord {
    uint32_t dbKey; int score; float seqId; double eval; int qStart, qEnd, qLen, dbStart, dbEnd, dbLen;
    float qcov, dbcov; unsigned int alnLength; std::string backtrace;
};

// <db>.lookup: key \t name \t set id (structureReadKeyToSet, structureconvertalis.cpp:204-227); <db>.source: set id \t source name, the
// name being the rest of the line (structureReadSetToSource, :230-251)
bool readWholeFile(const std::string &path, std::string &out, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}
bool readKeyToSet(const std::string &path, std::map<unsigned int, unsigned int> &m, std::string &err) {
    std::string text;
    if (!readWholeFile(path, text, err)) return false;
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(p, e - p);
        p = e + 1;
        unsigned int key = 0, set = 0;
        char name[4096];
        if (sscanf(line.c_str(), "%u %4095s %u", &key, name, &set) == 3) m.emplace(key, set);
    }
    return true;
}
bool readSetToSource(const std::string &path, std::map<unsigned int, std::string> &m, std::string &err) {
    std::string text;
    if (!readWholeFile(path, text, err)) return false;
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(p, e - p);
        p = e + 1;
        size_t a = 0;
        while (a < line.size() && (line[a] == ' ' || line[a] == '\t')) a++;
        size_t b = a;
        while (b < line.size() && line[b] != ' ' && line[b] != '\t') b++;
        size_t c = b;
        while (c < line.size() && (line[c] == ' ' || line[c] == '\t')) c++;
        if (a == b || c >= line.size()) continue;
        m.emplace((unsigned int) strtoul(line.c_str() + a, nullptr, 10), line.substr(c));
    }
    return true;
}

bool parseAlnRecord(const char *line, const char *end, AlnRecord &r, std::string &err) {
    const char *w[16];
    size_t n = 0;
    const char *p = line;
    while (p < end && n < 16) {
        while (p < end && (*p == ' ' || *p == '\t')) p++;
        if (p >= end || *p == '\n') break;
        w[n++] = p;
        while (p < end && *p != ' ' && *p != '\t' && *p != '\n') p++;
    }
    if (n < 10) { err = "Invalid alignment result record."; return false; }
    if (n != 10 && n != 11) { err = "alignment records with ORF columns are not produced on this path"; return false; }
    auto wordEnd = [&](size_t i) { const char *e = w[i]; while (e < end && *e != ' ' && *e != '\t' && *e != '\n') e++; return e; };
    // every field is parsed from a terminated copy: the record may be the last bytes of an entry that lost its terminator
    auto field = [&](size_t i) { return std::string(w[i], wordEnd(i)); };
    auto toInt = [&](size_t i) { return (int) strtol(field(i).c_str(), nullptr, 10); };
    r.dbKey = (uint32_t) strtoul(field(0).c_str(), nullptr, 10);
    r.score = toInt(1);
    const double seqId = strtod(field(2).c_str(), nullptr);
    r.eval = strtod(field(3).c_str(), nullptr);
    r.qStart = toInt(4); r.qEnd = toInt(5); r.qLen = toInt(6); r.dbStart = toInt(7); r.dbEnd = toInt(8); r.dbLen = toInt(9);
    const int aq = r.qStart == -1 ? 0 : r.qStart, ad = r.dbStart == -1 ? 0 : r.dbStart;
    auto cov = [](unsigned int s, unsigned int e, unsigned int len) { return (std::min(len, std::max(s, e)) - std::min(s, e) + 1) / (float) len; };   // SmithWaterman::computeCov
    r.qcov = (float) (double) cov((unsigned) aq, (unsigned) r.qEnd, (unsigned) r.qLen);
    r.dbcov = (float) (double) cov((unsigned) ad, (unsigned) r.dbEnd, (unsigned) r.dbLen);
    r.alnLength = (unsigned int) (std::max(abs(r.qEnd - aq), abs(r.dbEnd - ad)) + 1);
    r.seqId = (float) seqId;
    r.backtrace = n == 11 ? std::string(w[10], wordEnd(10)) : std::string();
    return true;
}

// Matcher::uncompressAlignment (Matcher.cpp:189-203).  false: the run lengths add up to more than `limit` columns (a damaged record: no
// alignment of two database entries is longer than their lengths together) -- the reference would expand it anyway and read past its
// sequences, here the record is refused by the caller.
bool expandBacktrace(const std::string &cbt, size_t limit, std::string &bt) {
    bt.clear();
    size_t count = 0;
    for (char c : cbt) {
        if (c >= '0' && c <= '9') { count = count * 10 + (size_t) (c - '0'); if (count > limit) return false; }
        else { const size_t n = count == 0 ? 1 : count; if (bt.size() + n > limit) return false; bt.append(n, c); count = 0; }
    }
    return true;
}

// structurePrintSeqBasedOnAln (structureconvertalis.cpp:133-171) without the nucleotide branches; target = the `reverse` argument
void appendAlignedSeq(std::string &out, const char *seq, unsigned int offset, const std::string &bt, bool target) {
    unsigned int pos = 0;
    for (char c : bt) {
        const char ch = seq[offset + pos];
        if (c == 'M') { out.push_back(ch); pos++; }
        else if (c == 'I') { if (target) out.push_back('-'); else { out.push_back(ch); pos++; } }
        else if (c == 'D') { if (target) { out.push_back(ch); pos++; } else out.push_back('-'); }
    }
}

void appendF3(std::string &out, float x) { char b[64]; out.append(b, (size_t) snprintf(b, sizeof(b), "%.3f", (double) x)); }        // SSTR(float): fmt "{:.3f}"
void appendE3(std::string &out, double x) { char b[64]; out.append(b, (size_t) snprintf(b, sizeof(b), "%.3E", x)); }               // SSTR(double): fmt "{:.3E}"

float probTruePositive(float score) {          // CalcProbTP::calculate (F/src/commons/CalcProbTP.h:8-32), float arithmetic as written there
    if (score <= 10) return 0;
    if (score >= 100) return 1.0;
    auto gammaPdf = [](const float alpha, const float beta, const float x) -> float {
        return std::exp(alpha * std::log(beta) + (alpha - 1) * std::log(x) + (-beta * x) - std::lgamma(alpha));    // float overloads, as <cmath> resolves them there
    };
    float p_tp = (0.8279 * gammaPdf(1.8123, 1 / 46.0042, score) + 0.1721 * gammaPdf(1.0057, 1 / 563.5014, score)) * 0.1023;
    float p_fp = (0.34 * gammaPdf(4.9259, 1 / 4.745, score) + 0.66 * gammaPdf(9.4834, 1 / 1.3136, score)) * 0.8977;
    return 1 / (1 + (p_fp / p_tp));
}

} // namespace

extern "C" int fsmod_convertalis(int argc, const char **argv) {
    Options o;
    {
        std::string perr;
        if (!parseArgs(argc, argv, "convertalis", {kConvertFlags, kCommonFlags}, o, perr)) return fail(perr);
    }
    if (o.pos.size() != 4) return fail("usage: convertalis <queryDB> <targetDB> <alignmentDB> <outFile> [--format-mode 0|2|4] [--format-output col,col,...] [--db-output 0|1]");
    int format = o.geti("--format-mode", 0);
    const bool columnHeaders = format == 4;
    if (columnHeaders) format = 0;
    const bool dbOut = o.geti("--db-output", 0) != 0;
    if (dbOut && columnHeaders) return fail("convertalis: --format-mode 4 with --db-output 1 is not implemented");
    const std::string outfmt = o.has("--format-output") ? o.kv["--format-output"] : "query,target,fident,alnlen,mismatch,gapopen,qstart,qend,tstart,tend,evalue,bits";
    std::vector<ConvCol> cols;
    std::vector<std::string> colNames;
    bool needSeq = false, need3Di = false, needBt = false, alignedCols = false;
    int needSets = 0;
    for (size_t b = 0; b <= outfmt.size();) {                      // Util::split(outfmt, ","): empty fields are skipped
        size_t e = outfmt.find(',', b);
        if (e == std::string::npos) e = outfmt.size();
        const std::string name = outfmt.substr(b, e - b);
        b = e + 1;
        if (name.empty()) continue;
        const ConvColSpec *spec = nullptr;
        for (const ConvColSpec &c : kConvCols) if (name == c.name) spec = &c;
        if (!spec) {
            for (const char *r : kConvRefused)
                if (name == r) return fail("convertalis: column " + name + " is not implemented on this path (needs the C-alpha, taxonomy or multimer data)");
            return fail("Format code " + name + " does not exist.");
        }
        cols.push_back(spec->col); colNames.push_back(name);
        needSeq = needSeq || spec->needSeq; need3Di = need3Di || spec->need3Di; needBt = needBt || spec->needBt;
        alignedCols = alignedCols || (spec->needBt && (spec->needSeq || spec->need3Di));
        needSets |= spec->needSets;
    }
    std::map<unsigned int, unsigned int> qKeyToSet, tKeyToSet;
    std::map<unsigned int, std::string> qSetToSource, tSetToSource;
    {
        std::string serr;
        if ((needSets & 1) && (!readKeyToSet(o.pos[0] + ".lookup", qKeyToSet, serr) || !readKeyToSet(o.pos[1] + ".lookup", tKeyToSet, serr))) return fail(serr);
        if ((needSets & 2) && (!readSetToSource(o.pos[0] + ".source", qSetToSource, serr) || !readSetToSource(o.pos[1] + ".source", tSetToSource, serr))) return fail(serr);
    }
    const bool sameDB = o.pos[0] == o.pos[1];
    std::string err;
    DbReader qSeq, tSeqOwn, q3, t3Own, qHdr, tHdrOwn, aln;
    if (!qHdr.openHeaders(o.pos[0], err)) return fail(err);
    if (!sameDB && !tHdrOwn.openHeaders(o.pos[1], err)) return fail(err);
    if (needSeq && (!qSeq.open(o.pos[0], err) || (!sameDB && !tSeqOwn.open(o.pos[1], err)))) return fail(err);
    if (need3Di && (!q3.open(dbPathWithSuffix(o.pos[0], "_ss"), err) || (!sameDB && !t3Own.open(dbPathWithSuffix(o.pos[1], "_ss"), err)))) return fail(err);
    if (!aln.open(o.pos[2], err)) return fail(err);
    const DbReader &tHdr = sameDB ? qHdr : tHdrOwn, &tSeq = sameDB ? qSeq : tSeqOwn, &t3 = sameDB ? q3 : t3Own;
    // entries of a padded (GPU layout) database are handed out as letters, soft-masked residues in lower case (DBReader::getUnpadded,
    // M/src/commons/DBReader.cpp:349-371)
    auto seqText = [](const DbReader &r, size_t id, std::string &buf) -> const char * {
        if (!(r.extended() & DBTYPE_EXTENDED_GPU)) return r.data(id);
        static const char letters[] = "ACDEFGHIKLMNPQRSTVWYX";
        const unsigned char *d = (const unsigned char *) r.data(id);
        const size_t n = r.seqLen(id);
        buf.resize(n);
        for (size_t k = 0; k < n; k++) {
            const unsigned char code = d[k], base = code >= 32 ? (unsigned char) (code - 32) : code;
            const char ch = letters[base <= 20 ? base : 20];
            buf[k] = code >= 32 ? (char) (ch | ' ') : ch;
        }
        return buf.c_str();
    };
    std::string qSeqBuf, q3Buf, tSeqBuf, t3Buf;

    // the reference walks the alignment DB in data-file order (DBReader::LINEAR_ACCCESS, structureconvertalis.cpp:451) and
    // concatenates the per-query blocks
    std::vector<size_t> order(aln.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return aln.offset(a) < aln.offset(b); });

    FILE *plain = nullptr;
    DbWriter w;
    if (dbOut) { if (!w.open(o.pos[3], 12 /* DBTYPE_GENERIC_DB */, err)) return fail(err); }
    else if (!(plain = fopen(o.pos[3].c_str(), "wb"))) return fail("cannot open " + o.pos[3] + " for writing");
    bool ioOk = true;
    if (columnHeaders && !cols.empty()) {
        std::string h;
        for (size_t i = 0; i < colNames.size(); i++) { if (i) h.push_back('\t'); h += colNames[i]; }
        h.push_back('\n');
        ioOk = fwrite(h.data(), 1, h.size(), plain) == h.size();
    }
    std::string result, bt;
    char buffer[1024];
    for (size_t oi = 0; oi < order.size(); oi++) {
        const size_t i = order[oi];
        const uint32_t queryKey = aln.key(i);
        const int64_t qh = qHdr.idOf(queryKey);
        if (qh < 0) return fail("convertalis: query key " + std::to_string(queryKey) + " has no header entry");
        const char *qHeader = qHdr.data((size_t) qh);
        const size_t qHeaderLen = qHdr.seqLen((size_t) qh);
        const std::string queryId = fastaHeaderName(qHeader, qHdr.entryLen((size_t) qh));
        const char *qSeqData = nullptr, *q3Data = nullptr;
        size_t qHave = SIZE_MAX;               // residues the loaded query entries really hold: what a record may address
        if (needSeq) { const int64_t id = qSeq.idOf(queryKey); if (id < 0) return fail("convertalis: query key " + std::to_string(queryKey) + " is not in " + o.pos[0]); qSeqData = seqText(qSeq, (size_t) id, qSeqBuf); qHave = std::min(qHave, (size_t) qSeq.seqLen((size_t) id)); }
        if (need3Di) { const int64_t id = q3.idOf(queryKey); if (id < 0) return fail("convertalis: query key " + std::to_string(queryKey) + " is not in the query 3Di database"); q3Data = seqText(q3, (size_t) id, q3Buf); qHave = std::min(qHave, (size_t) q3.seqLen((size_t) id)); }
        result.clear();
        const char *data = aln.data(i), *dataEnd = data + aln.entryLen(i);
        while (data < dataEnd && *data != '\0') {
            const char *lineEnd = data;
            while (lineEnd < dataEnd && *lineEnd != '\n' && *lineEnd != '\0') lineEnd++;
            AlnRecord res;
            if (!parseAlnRecord(data, lineEnd, res, err)) return fail(err);
            data = (lineEnd < dataEnd && *lineEnd == '\n') ? lineEnd + 1 : lineEnd;
            if (res.backtrace.empty() && needBt)
                return fail("Backtrace cigar is missing in the alignment result. Please recompute the alignment with the -a flag.");
            const int64_t th = tHdr.idOf(res.dbKey);
            if (th < 0) return fail("convertalis: target key " + std::to_string(res.dbKey) + " has no header entry");
            const char *tHeader = tHdr.data((size_t) th);
            const size_t tHeaderLen = tHdr.seqLen((size_t) th);
            const std::string targetId = fastaHeaderName(tHeader, tHdr.entryLen((size_t) th));
            // alignment length, gap opens, identities, mismatches (structureconvertalis.cpp:731-768)
            unsigned int gapOpenCount = 0, alnLen = res.alnLength, missMatchCount = 0, identical = 0;
            if (!res.backtrace.empty()) {
                size_t matchCount = 0;
                alnLen = 0;
                const std::string &b = res.backtrace;
                for (size_t pos = 0; pos < b.size(); pos++) {
                    int cnt = 0;
                    while (pos < b.size() && isdigit((unsigned char) b[pos])) {       // atoi there; saturating here (a damaged run length must not overflow)
                        if (cnt < 100000000) cnt = cnt * 10 + (b[pos] - '0');
                        pos++;
                    }
                    alnLen += (unsigned int) cnt;
                    if (pos >= b.size()) break;
                    if (b[pos] == 'M') matchCount += (size_t) cnt;
                    else if (b[pos] == 'D' || b[pos] == 'I') gapOpenCount += 1;
                }
                identical = static_cast<unsigned int>(res.seqId * static_cast<float>(alnLen) + 0.5);
                missMatchCount = static_cast<unsigned int>(matchCount - identical);
            } else {
                const int adjustQstart = (res.qStart == -1) ? 0 : res.qStart;
                const int adjustDBstart = (res.dbStart == -1) ? 0 : res.dbStart;
                const float bestMatchEstimate = static_cast<float>(std::min(abs(res.qEnd - adjustQstart), abs(res.dbEnd - adjustDBstart)));
                missMatchCount = static_cast<unsigned int>(bestMatchEstimate * (1.0f - res.seqId) + 0.5);
            }
            if (format == 2) {                     // FORMAT_ALIGNMENT_BLAST_WITH_LEN (:1198-1227)
                const int count = snprintf(buffer, sizeof(buffer), "%s\t%s\t%1.3f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.2E\t%d\t%d\t%d\n", queryId.c_str(), targetId.c_str(),
                                           res.seqId, alnLen, missMatchCount, gapOpenCount, res.qStart + 1, res.qEnd + 1, res.dbStart + 1, res.dbEnd + 1, res.eval,
                                           res.score, res.qLen, res.dbLen);
                if (count < 0 || (size_t) count >= sizeof(buffer)) { fprintf(stderr, "Truncated line in entry%zu!\n", i); continue; }
                result.append(buffer, (size_t) count);
                continue;
            }
            const char *tSeqData = nullptr, *t3Data = nullptr;
            size_t tHave = SIZE_MAX;
            if (needSeq) { const int64_t id = tSeq.idOf(res.dbKey); if (id < 0) return fail("convertalis: target key " + std::to_string(res.dbKey) + " is not in " + o.pos[1]); tSeqData = seqText(tSeq, (size_t) id, tSeqBuf); tHave = std::min(tHave, (size_t) tSeq.seqLen((size_t) id)); }
            if (need3Di) { const int64_t id = t3.idOf(res.dbKey); if (id < 0) return fail("convertalis: target key " + std::to_string(res.dbKey) + " is not in the target 3Di database"); t3Data = seqText(t3, (size_t) id, t3Buf); tHave = std::min(tHave, (size_t) t3.seqLen((size_t) id)); }
            // A record that addresses residues its entries do not have (lengths / start positions / backtrace that do not belong to these
            // databases): the reference prints whatever lies behind the entry; here it is an error naming the record.
            const auto misfit = [&](const char *what) {
                return fail("convertalis: the alignment of query " + std::to_string(queryKey) + " with target " + std::to_string(res.dbKey) + " does not fit the databases (" + what + ")");
            };
            if (needSeq || need3Di) {
                if (res.qLen < 0 || (size_t) res.qLen > qHave) return misfit("query length");
                if (res.dbLen < 0 || (size_t) res.dbLen > tHave) return misfit("target length");
            }
            if (needBt) {
                if (!expandBacktrace(res.backtrace, (size_t) std::max(res.qLen, 0) + (size_t) std::max(res.dbLen, 0) + 1, bt)) return misfit("backtrace longer than both entries");
                if (alignedCols) {
                    size_t useQ = 0, useT = 0;
                    for (char c : bt) { useQ += (c == 'M' || c == 'I'); useT += (c == 'M' || c == 'D'); }
                    if (res.qStart < 0 || (size_t) res.qStart + useQ > qHave) return misfit("backtrace leaves the query");
                    if (res.dbStart < 0 || (size_t) res.dbStart + useT > tHave) return misfit("backtrace leaves the target");
                }
            }
            if (cols.empty()) {                    // no column survived Util::split: the reference's fixed 12-column BLAST line (:776-800)
                const int count = snprintf(buffer, sizeof(buffer), "%s\t%s\t%1.3f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.2E\t%d\n", queryId.c_str(), targetId.c_str(),
                                           res.seqId, alnLen, missMatchCount, gapOpenCount, res.qStart + 1, res.qEnd + 1, res.dbStart + 1, res.dbEnd + 1, res.eval, res.score);
                if (count < 0 || (size_t) count >= sizeof(buffer)) { fprintf(stderr, "Truncated line in entry%zu!\n", i); continue; }
                result.append(buffer, (size_t) count);
                continue;
            }
            for (size_t c = 0; c < cols.size(); c++) {
                switch (cols[c]) {
                    case C_QUERY: result += queryId; break;
                    case C_TARGET: result += targetId; break;
                    case C_QKEY: result += std::to_string(queryKey); break;
                    case C_TKEY: result += std::to_string(res.dbKey); break;
                    case C_EVALUE: appendE3(result, res.eval); break;
                    case C_GAPOPEN: result += std::to_string(gapOpenCount); break;
                    case C_FIDENT: appendF3(result, res.seqId); break;
                    case C_PIDENT: appendF3(result, res.seqId * 100); break;
                    case C_NIDENT: result += std::to_string(identical); break;
                    case C_QSTART: result += std::to_string(res.qStart + 1); break;
                    case C_QEND: result += std::to_string(res.qEnd + 1); break;
                    case C_QLEN: result += std::to_string(res.qLen); break;
                    case C_TSTART: result += std::to_string(res.dbStart + 1); break;
                    case C_TEND: result += std::to_string(res.dbEnd + 1); break;
                    case C_TLEN: result += std::to_string(res.dbLen); break;
                    case C_ALNLEN: result += std::to_string(alnLen); break;
                    case C_BITS: result += std::to_string(res.score); break;
                    case C_CIGAR: result += res.backtrace; break;
                    case C_QSEQ: result.append(qSeqData, (size_t) res.qLen); break;
                    case C_TSEQ: result.append(tSeqData, (size_t) res.dbLen); break;
                    case C_Q3DI: result.append(q3Data, (size_t) res.qLen); break;
                    case C_T3DI: result.append(t3Data, (size_t) res.dbLen); break;
                    case C_QHEADER: result.append(qHeader, qHeaderLen); break;
                    case C_THEADER: result.append(tHeader, tHeaderLen); break;
                    case C_QALN: appendAlignedSeq(result, qSeqData, (unsigned int) res.qStart, bt, false); break;
                    case C_Q3DIALN: appendAlignedSeq(result, q3Data, (unsigned int) res.qStart, bt, false); break;
                    case C_TALN: appendAlignedSeq(result, tSeqData, (unsigned int) res.dbStart, bt, true); break;
                    case C_T3DIALN: appendAlignedSeq(result, t3Data, (unsigned int) res.dbStart, bt, true); break;
                    case C_MISMATCH: result += std::to_string(missMatchCount); break;
                    case C_QCOV: appendF3(result, res.qcov); break;
                    case C_TCOV: appendF3(result, res.dbcov); break;
                    case C_EMPTY: result.push_back('-'); break;
                    case C_PROB: appendF3(result, probTruePositive((float) res.score)); break;
                    case C_QSET: result += qSetToSource[qKeyToSet[queryKey]]; break;          // std::map::operator[] like the reference: absent -> 0 / ""
                    case C_QSETID: result += std::to_string(qKeyToSet[queryKey]); break;
                    case C_TSET: result += tSetToSource[tKeyToSet[res.dbKey]]; break;
                    case C_TSETID: result += std::to_string(tKeyToSet[res.dbKey]); break;
                }
                if (c + 1 < cols.size()) result.push_back('\t');
            }
            result.push_back('\n');
        }
        if (dbOut) w.write(queryKey, result.data(), result.size());
        else ioOk = ioOk && fwrite(result.data(), 1, result.size(), plain) == result.size();
    }
    if (dbOut) { if (!w.close(err)) return fail(err); }
    else {
        ioOk = (fclose(plain) == 0) && ioOk;
        if (!ioOk) return fail("write error on " + o.pos[3]);
    }
    return EXIT_SUCCESS;
}

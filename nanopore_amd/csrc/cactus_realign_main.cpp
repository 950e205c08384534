// cactus_realign -- the reference's process boundary, served by libnprealign.so.
//
// The reference does not link a realigner, it forks one: sonLib's system() runs
//     echo <exonerate cigar> | cactus_realign ref.fa read.fa --diagonalExpansion=10 --splitMatrixBiggerThanThis=3000
//          [--loadHmm=F] --gapGamma=G --matchGamma=M > out.cig                         (nanopore/analyses/utils.py:586-587)
//     cat cigars | cactus_realign ref.fa reads.fa --rescoreByPosteriorProbIgnoringGaps --rescoreOriginalAlignment
//          --diagonalExpansion=10 --splitMatrixBiggerThanThis=100 --outputPosteriorProbs=F --loadHmm=F > out
//                                                                                      (alignmentUncertainty.py:41)
//     ... --outputAllPosteriorProbs=F ...                                              (marginAlignSnpCaller.py:136-146)
// This is that program: argv[1], argv[2] are FASTA files (sequences are looked up BY NAME from the cigar, so their order
// does not matter), stdin carries one or more exonerate cigar lines
//     cigar: <query> <qstart> <qend> <strand> <target> <tstart> <tend> <strand> <score> (M|I|D <len>)*
// (query = read, target = reference, both '+': utils.py:173-177), stdout gets one cigar line per input line with the same
// names and coordinates, the new score and the new operations, and the posterior files get `refPos readPos prob` lines
// (marginAlignSnpCaller.py:149).  ALL cigars of stdin go to the GPU as ONE batch through the C ABI.  Exit status is
// non-zero on any failure, which is what makes the reference's system() raise (pipeline.py:209-210).
// No arithmetic lives here: it is argument parsing, FASTA / cigar text, and calls into include/nprealign.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "nprealign.h"

namespace {

struct Cigar {
    std::string query, target;
    long qstart = 0, qend = 0, tstart = 0, tend = 0;
    std::vector<int32_t> ops;  // (op, len) pairs
};

[[noreturn]] void die(const std::string &msg) {
    std::fprintf(stderr, "cactus_realign: %s\n", msg.c_str());
    std::exit(1);
}

void read_fasta(const char *path, std::map<std::string, std::string> &out) {
    std::ifstream in(path);
    if (!in) die(std::string("cannot open ") + path);
    std::string line, name;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            std::istringstream hs(line.substr(1));
            hs >> name;  // first word of the header, as getFastaDictionary keys them (utils.py:233-238)
            out[name].clear();
        } else if (!name.empty()) {
            out[name] += line;
        }
    }
}

bool parse_cigar(const std::string &line, Cigar &c) {
    std::istringstream ls(line);
    std::string tag, qs, ts;
    double score;
    if (!(ls >> tag) || tag != "cigar:") return false;
    if (!(ls >> c.query >> c.qstart >> c.qend >> qs >> c.target >> c.tstart >> c.tend >> ts >> score)) return false;
    if (qs != "+" || ts != "+") die("only '+' '+' cigars are supported (the reference reverse-complements SEQ itself, utils.py:327-331): " + line);
    std::string op;
    long len, qspan = 0, tspan = 0;
    while (ls >> op >> len) {
        const int code = op == "M" ? NPR_OP_M : (op == "I" ? NPR_OP_I : (op == "D" ? NPR_OP_D : -1));
        if (code < 0 || len < 0) die("bad cigar operation in: " + line);
        c.ops.push_back(code);
        c.ops.push_back(static_cast<int32_t>(len));
        if (code != NPR_OP_D) qspan += len;
        if (code != NPR_OP_I) tspan += len;
    }
    if (qspan != c.qend - c.qstart || tspan != c.tend - c.tstart) die("the operations do not span the coordinates of: " + line);
    return true;
}

bool load_hmm(const std::string &path, double *T, double *E) {
    std::ifstream in(path);
    if (!in) return false;
    std::string l1, l2;
    if (!std::getline(in, l1) || !std::getline(in, l2)) return false;
    std::istringstream a(l1), b(l2);
    double type, v;
    if (!(a >> type)) return false;
    for (int i = 0; i < 25; ++i)
        if (!(a >> T[i])) return false;
    (void)(a >> v);  // likelihood
    for (int i = 0; i < 80; ++i)
        if (!(b >> E[i])) return false;
    return true;
}

bool flag_value(const char *arg, const char *name, std::string &value) {
    const size_t n = std::strlen(name);
    if (std::strncmp(arg, name, n) != 0 || arg[n] != '=') return false;
    value = arg + n + 1;
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    npr_params P{};
    P.band_mode = NPR_BAND_ANCHOR;
    P.diagonal_expansion = 20;   // cactus_realign's own defaults; the reference always overrides the first two
    P.split_threshold = 3000;
    P.constraint_trim = 14;
    P.gap_gamma = 0.5;
    P.match_gamma = 0.0;
    P.posterior_threshold = 0.01;
    P.mode = NPR_MODE_REALIGN;
    std::vector<const char *> files;
    std::string hmm_file, post_file, all_post_file, v;
    bool rescore = false;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        if (a[0] == '\0') continue;  // nameValue() yields "" for an absent option (utils.py:586)
        if (flag_value(a, "--diagonalExpansion", v)) P.diagonal_expansion = std::atoi(v.c_str());
        else if (flag_value(a, "--splitMatrixBiggerThanThis", v)) P.split_threshold = std::atol(v.c_str());
        else if (flag_value(a, "--constraintDiagonalTrim", v)) P.constraint_trim = std::atoi(v.c_str());
        else if (flag_value(a, "--gapGamma", v)) P.gap_gamma = std::atof(v.c_str());
        else if (flag_value(a, "--matchGamma", v)) P.match_gamma = std::atof(v.c_str());
        else if (flag_value(a, "--loadHmm", v)) hmm_file = v;
        else if (flag_value(a, "--outputPosteriorProbs", v)) post_file = v;
        else if (flag_value(a, "--outputAllPosteriorProbs", v)) all_post_file = v;
        else if (flag_value(a, "--device", v)) device = std::atoi(v.c_str());
        else if (std::strcmp(a, "--rescoreOriginalAlignment") == 0) rescore = true;
        else if (std::strcmp(a, "--rescoreByPosteriorProbIgnoringGaps") == 0) {}  // the only rescoring this build has
        else if (std::strncmp(a, "--", 2) == 0) die(std::string("unknown option ") + a);
        else files.push_back(a);
    }
    if (files.size() != 2) die("usage: cactus_realign seqFile1 seqFile2 [options] < exonerate cigars");
    if (rescore) P.mode = NPR_MODE_RESCORE_ORIGINAL;
    else if (!all_post_file.empty() || !post_file.empty()) P.mode = NPR_MODE_ALL_POSTERIORS;

    std::map<std::string, std::string> seqs;
    read_fasta(files[0], seqs);
    read_fasta(files[1], seqs);
    std::vector<Cigar> cigars;
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.find_first_not_of(" \t\r") == std::string::npos) continue;
        Cigar c;
        if (!parse_cigar(line, c)) die("not an exonerate cigar line: " + line);
        cigars.push_back(c);
    }
    const int64_t n = static_cast<int64_t>(cigars.size());

    // one batch: the distinct target sequences as the reference table, one read buffer per cigar
    std::map<std::string, int32_t> ref_id;
    std::vector<const std::string *> ref_seq;
    std::vector<int32_t> ref_index(n), guide_ops;
    std::vector<int64_t> guide_off(n + 1, 0), guide_start(2 * n), read_off(n + 1, 0);
    std::string reads;
    for (int64_t i = 0; i < n; ++i) {
        const Cigar &c = cigars[i];
        const auto t = seqs.find(c.target), q = seqs.find(c.query);
        if (t == seqs.end()) die("sequence " + c.target + " is in neither FASTA file");
        if (q == seqs.end()) die("sequence " + c.query + " is in neither FASTA file");
        const auto ins = ref_id.emplace(c.target, static_cast<int32_t>(ref_seq.size()));
        if (ins.second) ref_seq.push_back(&t->second);
        ref_index[i] = ins.first->second;
        reads += q->second;
        read_off[i + 1] = static_cast<int64_t>(reads.size());
        guide_ops.insert(guide_ops.end(), c.ops.begin(), c.ops.end());
        guide_off[i + 1] = static_cast<int64_t>(guide_ops.size() / 2);
        guide_start[2 * i] = c.tstart, guide_start[2 * i + 1] = c.qstart;
    }
    std::string refs;
    std::vector<int64_t> ref_off(ref_seq.size() + 1, 0);
    for (size_t k = 0; k < ref_seq.size(); ++k) refs += *ref_seq[k], ref_off[k + 1] = static_cast<int64_t>(refs.size());

    char err[512] = {0};
    npr_ctx *ctx = nullptr;
    int32_t rc = npr_create(device, &ctx, err, sizeof(err));
    if (rc != NPR_OK) die(std::string(err[0] ? err : npr_strerror(rc)));
    if (!hmm_file.empty()) {
        double T[25], E[80];
        if (!load_hmm(hmm_file, T, E)) die("cannot read the hmm file " + hmm_file);
        if ((rc = npr_set_hmm(ctx, 0, T, E)) != NPR_OK) die(std::string("--loadHmm: ") + npr_last_error(ctx));
    }
    npr_batch *b = nullptr;
    rc = npr_batch_create_at(ctx, &P, n, static_cast<int64_t>(ref_seq.size()), reinterpret_cast<const uint8_t *>(refs.data()), ref_off.data(),
                             ref_index.data(), reinterpret_cast<const uint8_t *>(reads.data()), read_off.data(), guide_ops.data(),
                             guide_off.data(), guide_start.data(), nullptr, &b);
    if (rc == NPR_OK) rc = npr_batch_run(b, nullptr);
    if (rc == NPR_OK) rc = npr_batch_finish(b);
    if (rc != NPR_OK) die(std::string("realignment failed: ") + npr_last_error(ctx) + " (" + npr_strerror(rc) + ")");
    std::vector<npr_read_result> res(n);
    std::vector<int64_t> ops_off(n + 1, 0);
    if (n) {
        if ((rc = npr_batch_results(b, res.data())) != NPR_OK || (rc = npr_batch_ops(b, ops_off.data(), nullptr, 0)) != NPR_OK) die(npr_strerror(rc));
    }
    std::vector<int32_t> ops(2 * ops_off[n] + 2);
    if (n && (rc = npr_batch_ops(b, ops_off.data(), ops.data(), ops_off[n])) != NPR_OK) die(npr_strerror(rc));
    for (int64_t i = 0; i < n; ++i)
        if (res[i].status != NPR_OK) {  // the reference asserts exactly one cigar per input cigar (utils.py:588-589)
            std::fprintf(stderr, "cactus_realign: %s against %s: %s\n", cigars[i].query.c_str(), cigars[i].target.c_str(), npr_strerror(res[i].status));
            return 1;
        }
    static const char letter[3] = {'M', 'I', 'D'};
    for (int64_t i = 0; i < n; ++i) {
        const Cigar &c = cigars[i];
        std::printf("cigar: %s %ld %ld + %s %ld %ld + %f", c.query.c_str(), c.qstart, c.qend, c.target.c_str(), c.tstart, c.tend, res[i].score);
        for (int64_t q = ops_off[i]; q < ops_off[i + 1]; ++q) std::printf(" %c %d", letter[ops[2 * q]], ops[2 * q + 1]);
        std::printf("\n");
    }
    if (!all_post_file.empty() || !post_file.empty()) {
        std::vector<int64_t> poff(n + 1, 0);
        if ((rc = npr_batch_pairs(b, poff.data(), nullptr, nullptr, nullptr, 0)) != NPR_OK) die(npr_strerror(rc));
        std::vector<int32_t> px(poff[n] + 1), py(poff[n] + 1);
        std::vector<float> pp(poff[n] + 1);
        if ((rc = npr_batch_pairs(b, poff.data(), px.data(), py.data(), pp.data(), poff[n])) != NPR_OK) die(npr_strerror(rc));
        if (!all_post_file.empty()) {
            FILE *f = std::fopen(all_post_file.c_str(), "w");
            if (!f) die("cannot write " + all_post_file);
            for (int64_t k = 0; k < poff[n]; ++k) std::fprintf(f, "%d\t%d\t%.9g\n", px[k], py[k], static_cast<double>(pp[k]));
            std::fclose(f);
        }
        if (!post_file.empty()) {  // the pairs of the alignment that is printed: walk its M columns through the sorted list
            FILE *f = std::fopen(post_file.c_str(), "w");
            if (!f) die("cannot write " + post_file);
            for (int64_t i = 0; i < n; ++i) {
                int64_t x = cigars[i].tstart, y = cigars[i].qstart, k = poff[i];
                for (int64_t q = ops_off[i]; q < ops_off[i + 1]; ++q) {
                    const int32_t op = ops[2 * q], len = ops[2 * q + 1];
                    if (op == NPR_OP_M) {
                        for (int32_t t = 0; t < len; ++t, ++x, ++y) {
                            while (k < poff[i + 1] && (px[k] < x || (px[k] == x && py[k] < y))) ++k;
                            if (k < poff[i + 1] && px[k] == x && py[k] == y) std::fprintf(f, "%d\t%d\t%.9g\n", px[k], py[k], static_cast<double>(pp[k]));
                        }
                    } else if (op == NPR_OP_I) {
                        y += len;
                    } else {
                        x += len;
                    }
                }
            }
            std::fclose(f);
        }
    }
    npr_batch_destroy(b);
    npr_destroy(ctx);
    return 0;
}

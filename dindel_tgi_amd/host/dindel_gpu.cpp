// dindel_gpu — the --analysis indels --doDiploid window loop on the GPU likelihood path:
//   (BAM + .bai, window file, haplotype fixture[, library file])  ->  PREFIX.glf.txt
//
// Restates DetInDel::detectIndels (reference DInDel.cpp:1265-1424) as prepare-N / compute / reduce-N:
//   prepare  per window, in file order: getReads (get_reads.cpp) and the window's candidate haplotypes.  The reference BUILDS
//            those (getHaplotypes, DInDel.cpp:1526-1645: HaplotypeDistribution, SeqAn alignment) — out of this repository's
//            scope; they are read from a fixture file (window_io.hpp) instead;
//   compute  LikelihoodEngine::computeLikelihoodsBatch over the N prepared windows (one launch sequence on the GPU);
//   reduce   per window, in file order: diploidGLF (diploid_glf.cpp) -> the window's lines, or the skipped-window line with
//            the message the reference would print ("error_" + what was thrown).
// One BAM file: with a single pool the read buffer's reset after a skipped window (DInDel.cpp:1401-1408) does not change which
// reads a window sees, so preparing windows ahead of their predecessors' likelihood step is exact.
//
// Options (names follow the reference's CLI, DInDel.cpp:4079-4170):
//   --bamFile F --varFile F [--varFileIsOneBased] --hapFile F --outputFile PREFIX [--libFile F] [--faster] [--filterHaplotypes]
//   [--maxRead N] [--maxReadLength N] [--minReadOverlap N] [--mapQualThreshold X] [--pError X] [--pMut X] [--maxLengthIndel N]
//   [--flankRefSeq N] [--flankMaxMismatch N] [--priorSNP X] [--priorIndel X] [--capMapQualThreshold X] [--capMapQualFast X]
//   [--maxHapReadProd N] [--batchWindows N] [--device D] [--quiet]
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include "compute_likelihoods.hpp"
#include "diploid_glf.hpp"
#include "get_reads.hpp"
#include "glf_output.hpp"
#include "window_io.hpp"

using namespace dindel;

namespace {
struct WindowTask {
    int index; std::string tid; uint32_t pos, fileLeftPos, fileRightPos, leftPos, rightPos;
    AlignedCandidates candidates;
    std::vector<Read> reads;
    const std::vector<Haplotype> *haps;
    std::string message;                 // "ok" or the skipped message
    bool skipped;
};
}

int main(int argc, char **argv)
{
    std::map<std::string, std::string> opt;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.compare(0, 2, "--") != 0) { std::cerr << "Unknown argument " << a << "\n"; return 2; }
        a = a.substr(2);
        if (a == "varFileIsOneBased" || a == "faster" || a == "filterHaplotypes" || a == "quiet" || a == "doDiploid") opt[a] = "1";
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { std::cerr << "Option --" << a << " needs a value\n"; return 2; }
    }
    auto has = [&](const char *k) { return opt.find(k) != opt.end(); };
    auto num = [&](const char *k, double dflt) { return has(k) ? atof(opt[k].c_str()) : dflt; };
    for (const char *need : {"bamFile", "varFile", "hapFile", "outputFile"})
        if (!has(need)) { std::cerr << "Please specify --" << need << "\n"; return 1; }
    try {
        ObservationModelParameters obs;
        obs.setCLIDefaultValues();
        obs.pError = num("pError", obs.pError); obs.pMut = num("pMut", obs.pMut);
        obs.maxLengthIndel = obs.maxLengthDel = int(num("maxLengthIndel", obs.maxLengthIndel));
        obs.padCover = int(num("flankRefSeq", obs.padCover)); obs.maxMismatch = int(num("flankMaxMismatch", obs.maxMismatch));
        obs.mapQualThreshold = num("capMapQualThreshold", obs.mapQualThreshold); obs.capMapQualFast = num("capMapQualFast", obs.capMapQualFast);
        ReadSelectionParameters rsp;
        rsp.maxReads = size_t(num("maxRead", double(rsp.maxReads))); rsp.maxReadLength = size_t(num("maxReadLength", double(rsp.maxReadLength)));
        rsp.minReadOverlap = int(num("minReadOverlap", rsp.minReadOverlap)); rsp.mapQualThreshold = num("mapQualThreshold", rsp.mapQualThreshold);
        rsp.quiet = has("quiet");
        DiploidParameters dip;
        dip.priorSNP = num("priorSNP", dip.priorSNP); dip.priorIndel = num("priorIndel", dip.priorIndel);
        dip.filterHaplotypes = has("filterHaplotypes"); dip.quiet = has("quiet");
        const double maxHapReadProd = num("maxHapReadProd", 10000000.0);
        const int batchWindows = int(num("batchWindows", 256));
        const bool faster = has("faster"), oneBased = has("varFileIsOneBased");

        LibraryCollection libraries;
        if (has("libFile")) {                    // the reference: --libFile switches mapUnmappedReads on (DInDel.cpp:4268-4272)
            libraries.addFromFile(opt["libFile"]);
            rsp.mapUnmappedReads = true;
            obs.mapUnmappedReads = true;
        }
        BamFile bam(opt["bamFile"]);
        std::vector<BamFile *> bams(1, &bam);
        HaplotypeFixture fixture(opt["hapFile"]);
        ReadFetcher fetcher(bams, libraries, rsp);
        LikelihoodEngine engine(obs, int(num("device", 0)));
        engine.setThrowOnPositiveLikelihood(false);

        const std::string glfFile = opt["outputFile"] + ".glf.txt";
        std::ofstream glfOutput(glfFile.c_str());
        if (!glfOutput.is_open()) throw std::string("Cannot open file ").append(glfFile).append(" for writing.");
        OutputData glfData = makeGLFOutputData(glfOutput);
        glfData.outputLine(glfData.headerString());                               // DInDel.cpp:1290-1291

        VariantFile vf(opt["varFile"]);
        int index = 0;
        std::string oldTid("-1");
        std::vector<WindowTask> batch;
        long nWindows = 0, nSkipped = 0;

        auto flush = [&]() {
            // ---- compute: every prepared window of the batch in one call ----
            std::vector<WindowJob> jobs;
            std::vector<size_t> jobOf(batch.size(), size_t(-1));
            for (size_t i = 0; i < batch.size(); i++) if (!batch[i].skipped) {
                WindowJob J;
                J.haps = batch[i].haps; J.reads = &batch[i].reads; J.leftPos = batch[i].leftPos; J.rightPos = batch[i].rightPos;
                jobOf[i] = jobs.size();
                jobs.push_back(J);
            }
            if (!jobs.empty()) { if (faster) engine.computeLikelihoodsFasterBatch(jobs); else engine.computeLikelihoodsBatch(jobs); }
            // ---- reduce, in window order ----
            for (size_t i = 0; i < batch.size(); i++) {
                WindowTask &T = batch[i];
                if (!T.skipped) {
                    const WindowJob &J = jobs[jobOf[i]];
                    try {
                        if (!J.error.empty()) throw std::string(J.error);
                        // like the reference, diploidGLF writes its lines as it goes: if it throws half-way ("genotyping error"),
                        // the lines already written stay and the skipped-window line follows them
                        diploidGLF(*T.haps, T.reads, J.result, T.pos, T.leftPos, T.rightPos, glfData, T.index, T.tid, T.candidates, dip, "dip");
                    } catch (std::string &s) {
                        T.message = skippedMessage(s);
                        T.skipped = true;
                    }
                }
                if (T.skipped) {
                    std::cerr << "skipped " << T.tid << " " << T.pos << " reason: " << T.message << std::endl;     // DInDel.cpp:1383
                    glfData.output(skippedWindowLine(glfData, T.message, T.index, T.tid, T.fileLeftPos, T.fileRightPos));
                    nSkipped++;
                }
                nWindows++;
            }
            batch.clear();
        };

        while (!vf.eof()) {
            AlignedCandidates cand = vf.getLineVector(oneBased);
            if (cand.variants.size() == 0) continue;
            WindowTask T;
            T.candidates = cand; T.tid = cand.tid; T.pos = uint32_t(cand.centerPos);
            T.fileLeftPos = T.leftPos = uint32_t(cand.leftPos); T.fileRightPos = T.rightPos = uint32_t(cand.rightPos);
            T.haps = NULL; T.skipped = false; T.message = "ok";
            if (T.tid != oldTid) {                                                  // DInDel.cpp:1327-1333
                if (!batch.empty()) flush();
                fetcher.newChromosome();
                oldTid = T.tid;
            }
            if (T.fileLeftPos < fetcher.previousLeftPos()) {                          // :1335-1339
                std::cerr << "leftPos: " << T.fileLeftPos << " oldLeftPos: " << fetcher.previousLeftPos() << std::endl;
                std::cerr << "Candidate variant files must be sorted on left position of window!" << std::endl;
                return 1;
            }
            T.index = ++index;
            try {
                fetcher.getReads(T.tid, T.fileLeftPos, T.fileRightPos, T.reads);
                const WindowHaplotypes *wh = fixture.find(T.index);
                if (!wh) throw std::string("no haplotypes for this window in the haplotype file");
                T.haps = &wh->haps; T.leftPos = wh->leftPos; T.rightPos = wh->rightPos;
                if (double(T.reads.size() * T.haps->size()) > maxHapReadProd) {     // :395-399
                    std::stringstream os;
                    os << "skipped_numhap_times_numread>" << long(maxHapReadProd);
                    throw os.str();
                }
            } catch (std::string &s) {
                T.message = skippedMessage(s);
                T.skipped = true;
            }
            fetcher.windowDone(T.skipped, T.fileLeftPos);                             // :1401-1408
            batch.push_back(T);
            if (int(batch.size()) >= batchWindows) flush();
        }
        if (!batch.empty()) flush();
        glfOutput.close();
        if (!has("quiet")) std::cout << "windows: " << nWindows << " skipped: " << nSkipped << " -> " << glfFile << std::endl;
    } catch (std::string &s) {
        std::cerr << "Exception: " << s << std::endl;
        return 1;
    }
    return 0;
}

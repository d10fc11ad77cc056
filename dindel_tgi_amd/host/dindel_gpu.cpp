// dindel_gpu — the --analysis indels --doDiploid window loop on the GPU likelihood path:
//   (BAM + .bai, window file, haplotype fixture[, library file])  ->  PREFIX.glf.txt
//
// Restates DetInDel::detectIndels (reference DInDel.cpp:1265-1424) as prepare-N / compute / reduce-N:
//   prepare  per window, in file order: getReads (get_reads.cpp) and the window's candidate haplotypes.  The reference BUILDS
//            those (getHaplotypes, DInDel.cpp:1526-1645: HaplotypeDistribution, SeqAn alignment) — out of this repository's
//            scope; they are read from a fixture file (window_io.hpp) instead;
//   compute  LikelihoodEngine::computeLikelihoodsBatch over the N prepared windows (one launch sequence on the GPU);
//   reduce   per window: diploidGLF (diploid_glf.cpp) -> the window's lines, or the skipped-window line with the message the
//            reference would print ("error_" + what was thrown); lines are written in file order.
// The stages run as a pipeline: the main thread reads the window file and cuts it into batches; --prepareThreads workers, each
// with its own handle on the BAM file and its own read buffer, prepare whole batches side by side; one thread feeds the GPU in
// batch order; one thread (with --reduceThreads helpers, each window into its own buffer) reduces and writes in window order.
// --bamFiles LIST (one path per line, first word of the line; reference DInDel.cpp:64-88): the files are pools of one read buffer
// (Read::fetchFuncVectorPooled, poolID = position in the list).  With several pools the buffer's ORDER depends on its history — a
// window's new records are appended pool after pool behind the survivors of the windows before — and std::sort's order inside a tie of
// mapping qualities follows it, so a worker does not start a batch with an empty buffer: it first replays read selection over the
// batch's look-back (the preceding windows of the chromosome back to one whose reads have all left the buffer by the batch's first
// window), which leaves the buffer in the state the window-by-window loop would have.  What a worker cannot see is a window skipped
// AFTER read selection (an exception of the likelihood or genotyping step: "hapSize error.", "Nan detected", ...): the reference empties
// the buffer after it (DInDel.cpp:1404-1405).  The writer sees it: from the window behind such a LATE skip it re-prepares, from an empty
// buffer and window by window, every window whose buffer could still hold a record of that moment (same chromosome, leftPos less than
// 2 maxInsert + 200 behind the first re-prepared window's rightPos) with a read fetcher, an engine and the reduce step of its own, and
// writes those results instead of the ones prepared ahead — beyond that reach both histories hold the same records in the same order.
// One BAM file: with a single pool the read buffer's reset (after a skipped window in the reference, DInDel.cpp:1401-1408; at
// the head of every batch here) does not change which reads a window sees — the buffer always holds the file's reads starting in
// [leftPos - maxInsert - 200, rightPos + maxInsert), in file order — so preparing windows ahead of their predecessors'
// likelihood step, and batches side by side, is exact.  (One counter does depend on the buffer's history: "Too many reads in
// region" fires on buffer size + records fetched > 100 * maxRead, which at a batch's first window is counted as after a reset.)
//
// Options (names follow the reference's CLI, DInDel.cpp:4079-4170):
//   --bamFile F | --bamFiles LIST   --varFile F [--varFileIsOneBased] --hapFile F --outputFile PREFIX [--libFile F] [--faster] [--filterHaplotypes]
//   [--maxRead N] [--maxReadLength N] [--minReadOverlap N] [--mapQualThreshold X] [--pError X] [--pMut X] [--maxLengthIndel N]
//   [--filterReadAux STR] [--flankRefSeq N] [--flankMaxMismatch N] [--priorSNP X] [--priorIndel X] [--capMapQualThreshold X] [--capMapQualFast X]
//   [--maxHapReadProd N] [--batchWindows N] [--prepareThreads N] [--computeThreads N] [--packThreads N] [--reduceThreads N] [--device D | --devices D0,D1,...] [--quiet]
//   [--outputRealignedBAM]   per window PREFIX.ra.INDEX_TID_LEFT_RIGHT.bam with the reads realigned through the most likely haplotype
//                     pair (DInDel.cpp:589-620; main model only, like the reference; the haplotype file needs its A records)
//   [--timing]        one "timing:" line on stdout with the busy time of each stage
//   [--prepareOnly]   stop after the prepare stage (no likelihoods, no calls: profiling the read selection on a GPU-less host)
//   [--windowByWindow] tests: the writer re-does every window one after the other with a read buffer of its own (the reference's loop as it stands)
//   [--windowByWindow] tests: the writer re-does every window one after the other with a read buffer of its own (the reference's loop as it stands)
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <thread>
#include "compute_likelihoods.hpp"
#include "diploid_glf.hpp"
#include "get_reads.hpp"
#include "glf_output.hpp"
#include "realigned_bam.hpp"
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
    bool lateSkip;                       // skipped by the likelihood or genotyping step, i.e. after read selection (DInDel.cpp:1369-1408)
    std::string lines;                   // what the reduce stage wrote for this window
};
struct Batch {
    long seq;                            // position in the file: batches are computed and written in this order
    // A finished batch goes back to the reader and is filled again: its windows' read vectors (one string and one vector of
    // base qualities per read) are overwritten in place instead of being freed by one thread and allocated anew by another.
    std::vector<WindowTask> tasks;
    std::vector<WindowJob> jobs;
    std::vector<size_t> jobOf;
    std::vector<int> toRelease;          // windows of the batch's previous use whose parsed haplotypes are no longer needed
    // several BAM pools: the windows in front of the batch (same chromosome, file order) whose read selection is replayed first
    struct Before { std::string tid; uint32_t leftPos, rightPos; int index; };
    std::vector<Before> lookBack;
};
typedef std::unique_ptr<Batch> BatchPtr;

// hand-over between two pipeline stages: at most `cap` batches wait in it
class Channel {
public:
    explicit Channel(size_t cap) : cap_(cap), closed_(false) {}
    bool push(BatchPtr &b)               // false: the consumer is gone
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return false;
        q_.push_back(std::move(b));
        cv_.notify_all();
        return true;
    }
    bool pop(BatchPtr &b)                // false: closed and drained
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        b = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    void close() { std::lock_guard<std::mutex> lk(m_); closed_ = true; cv_.notify_all(); }
    void abort() { std::lock_guard<std::mutex> lk(m_); closed_ = true; q_.clear(); cv_.notify_all(); }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<BatchPtr> q_; size_t cap_; bool closed_;
};

// hand-over that restores file order: batches may arrive in any order, leave by sequence number; a batch more than `cap`
// ahead of the next one to leave waits at the door (the one that is next never does, so the stages cannot lock up)
class OrderedChannel {
public:
    explicit OrderedChannel(long cap) : cap_(cap), next_(0), closed_(false), aborted_(false) {}
    bool push(BatchPtr &b)
    {
        std::unique_lock<std::mutex> lk(m_);
        const long seq = b->seq;
        cv_.wait(lk, [&] { return seq < next_ + cap_ || aborted_; });
        if (aborted_) return false;
        held_[seq] = std::move(b);
        cv_.notify_all();
        return true;
    }
    bool pop(BatchPtr &b)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return held_.find(next_) != held_.end() || closed_ || aborted_; });
        std::map<long, BatchPtr>::iterator it = held_.find(next_);
        if (aborted_ || it == held_.end()) return false;
        b = std::move(it->second);
        held_.erase(it);
        next_++;
        cv_.notify_all();
        return true;
    }
    bool tryPop(BatchPtr &b)             // the next batch if it is already here; never waits
    {
        std::lock_guard<std::mutex> lk(m_);
        std::map<long, BatchPtr>::iterator it = held_.find(next_);
        if (aborted_ || it == held_.end()) return false;
        b = std::move(it->second);
        held_.erase(it);
        next_++;
        cv_.notify_all();
        return true;
    }
    void close() { std::lock_guard<std::mutex> lk(m_); closed_ = true; cv_.notify_all(); }
    void abort() { std::lock_guard<std::mutex> lk(m_); aborted_ = true; held_.clear(); cv_.notify_all(); }
private:
    std::mutex m_; std::condition_variable cv_; std::map<long, BatchPtr> held_; long cap_, next_; bool closed_, aborted_;
};

class BatchPool {
public:
    BatchPtr take()
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.empty()) return BatchPtr(new Batch);
        BatchPtr b = std::move(free_.back());
        free_.pop_back();
        return b;
    }
    void give(BatchPtr &b) { std::lock_guard<std::mutex> lk(m_); free_.push_back(std::move(b)); }
private:
    std::mutex m_; std::vector<BatchPtr> free_;
};

double seconds_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}

int main(int argc, char **argv)
{
    std::map<std::string, std::string> opt;
    for (int i = 1; i < argc; i++) if (!strcmp(argv[i], "--help") || !strcmp(argv[i], "-h")) {
        std::cout <<
            "dindel_gpu: the --analysis indels --doDiploid window loop with the likelihood step on the GPU\n"
            "  required: --bamFile F --varFile F --hapFile F --outputFile PREFIX          (writes PREFIX.glf.txt)\n"
            "  model:    [--faster] [--libFile F] [--filterHaplotypes] [--outputRealignedBAM] [--varFileIsOneBased]\n"
            "            [--maxRead N] [--maxReadLength N] [--minReadOverlap N] [--mapQualThreshold X] [--filterReadAux STR] [--pError X] [--pMut X] [--maxLengthIndel N]\n"
            "            [--flankRefSeq N] [--flankMaxMismatch N] [--priorSNP X] [--priorIndel X] [--capMapQualThreshold X] [--capMapQualFast X] [--maxHapReadProd N]\n"
            "  running:  [--batchWindows N] [--mergeBatches N] [--device D | --devices D0,D1,...] [--prepareThreads N] [--computeThreads N] [--packThreads N] [--reduceThreads N]\n"
            "            [--quiet] [--timing] [--prepareOnly]\n"
            "  files:    --varFile: the reference's window file; --hapFile: W / H / V / A records (host/window_io.hpp)\n";
        return 0;
    }
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.compare(0, 2, "--") != 0) { std::cerr << "Unknown argument " << a << "\n"; return 2; }
        a = a.substr(2);
        if (a == "varFileIsOneBased" || a == "faster" || a == "filterHaplotypes" || a == "quiet" || a == "doDiploid" || a == "timing" || a == "outputRealignedBAM" ||
            a == "prepareOnly" || a == "noLookBack" || a == "lateSkipsKnown" || a == "windowByWindow") opt[a] = "1";
        else if (i + 1 < argc) opt[a] = argv[++i];
        else { std::cerr << "Option --" << a << " needs a value\n"; return 2; }
    }
    auto has = [&](const char *k) { return opt.find(k) != opt.end(); };
    auto num = [&](const char *k, double dflt) { return has(k) ? atof(opt[k].c_str()) : dflt; };
    for (const char *need : {"varFile", "hapFile", "outputFile"})
        if (!has(need)) { std::cerr << "Please specify --" << need << "\n"; return 1; }
    if (!has("bamFile") && !has("bamFiles")) { std::cerr << "Error: Specify either --bamFile or --bamFiles." << std::endl; return 1; }   // DInDel.cpp:4215-4218
    try {
        const std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();
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
        if (has("filterReadAux")) rsp.filterReadAux = opt["filterReadAux"];
        DiploidParameters dip;
        dip.priorSNP = num("priorSNP", dip.priorSNP); dip.priorIndel = num("priorIndel", dip.priorIndel);
        dip.filterHaplotypes = has("filterHaplotypes"); dip.quiet = has("quiet");
        const double maxHapReadProd = num("maxHapReadProd", 10000000.0);
        const int batchWindows = std::max(1, int(num("batchWindows", 256)));
        const bool faster = has("faster"), oneBased = has("varFileIsOneBased"), prepareOnly = has("prepareOnly");
        const bool realignedBAM = has("outputRealignedBAM") && !faster;                  // `params.outputRealignedBAM && params.slower`, :589
        rsp.keepRecords = realignedBAM;
        unsigned hw = std::thread::hardware_concurrency();
        if (!hw) hw = 1;
        // defaults measured on a 16-CPU share of an MI355X host (profiles/r03/n2_pipeline.md): per window the read selection costs
        // 0.10-0.20 ms of CPU, diploidGLF 0.09 ms, packing 0.025 ms; two engines per GPU keep it busy while one of them packs.  They
        // scale with the number of devices (set below, once --devices is known) up to what the host has.
        // --devices 0,1,...: the engines are dealt out over these GPUs (batches are independent: no exchange between devices)
        std::vector<int> devices;
        {
            std::string list = has("devices") ? opt["devices"] : (has("device") ? opt["device"] : std::string("0"));
            for (size_t i = 0; i <= list.size();) {
                size_t e = list.find(',', i);
                if (e == std::string::npos) e = list.size();
                if (e > i) devices.push_back(atoi(list.substr(i, e - i).c_str()));
                i = e + 1;
            }
            if (devices.empty()) devices.push_back(0);
        }
        const unsigned nDev = unsigned(devices.size());
        const int computeThreads = std::max(1, int(num("computeThreads", (hw >= 8 ? 2.0 : 1.0) * double(nDev))));
        const int packThreads = int(num("packThreads", double(std::min(4u, std::max(1u, hw / (4 * nDev))))));   // host threads of each engine's packing (0: the engine's default)
        const int reduceThreads = std::max(1, int(num("reduceThreads", double(std::min(4u * nDev, std::max(1u, hw / 4))))));
        const int prepareThreads = std::max(1, int(num("prepareThreads", double(std::min(8u * nDev, std::max(1u, hw / 2))))));

        LibraryCollection libraries;
        if (has("libFile")) {                    // the reference: --libFile switches mapUnmappedReads on (DInDel.cpp:4268-4272)
            libraries.addFromFile(opt["libFile"]);
            rsp.mapUnmappedReads = true;
            obs.mapUnmappedReads = true;
        }
        std::vector<std::string> bamPaths;       // --bamFile wins when both are given (DInDel.cpp:4220-4226)
        if (has("bamFile")) bamPaths.push_back(opt["bamFile"]);
        else {
            std::ifstream list(opt["bamFiles"].c_str());
            if (!list.is_open()) { std::cout << "Cannot open file with BAM files:  " << opt["bamFiles"] << std::endl; throw std::string("File open error."); }
            std::string line;
            while (std::getline(list, line)) {
                std::istringstream is(line);
                std::string fname;
                is >> fname;
                if (!fname.empty()) bamPaths.push_back(fname);
            }
            if (bamPaths.empty()) throw std::string("No BAM file in ").append(opt["bamFiles"]);
        }
        const BamFile headerBam(bamPaths[0]);    // "Cannot open BAM file." / "Cannot open BAM index." before anything else happens; the header for --outputRealignedBAM
        for (size_t i = 1; i < bamPaths.size(); i++) { const BamFile probe(bamPaths[i]); (void)probe; }
        HaplotypeFixture fixture(opt["hapFile"]);
        // tests: --injectLateSkip I,J,... makes the reduce step of these windows throw "hapSize error." (a late skip, also under
        // --prepareOnly); with --lateSkipsKnown the prepare stage is told in advance and resets its buffers behind them — the
        // window-by-window history by construction, against which the writer's re-preparation is checked
        std::set<int> injectedLateSkips;
        if (has("injectLateSkip")) {
            const std::string list = opt["injectLateSkip"];
            for (size_t i = 0; i <= list.size();) {
                size_t e = list.find(',', i);
                if (e == std::string::npos) e = list.size();
                if (e > i) injectedLateSkips.insert(atoi(list.substr(i, e - i).c_str()));
                i = e + 1;
            }
        }
        const bool lateSkipsKnown = has("lateSkipsKnown");
        // --windowByWindow (tests, diagnostics): the writer re-does EVERY window — read selection from its own buffer, likelihoods, genotyping —
        // one after the other, i.e. the reference's loop as it stands (DInDel.cpp:1310-1411); what the pipeline prepared ahead is ignored
        const bool windowByWindow = has("windowByWindow");

        const std::string outputPrefix = opt["outputFile"];
        const char *dumpReads = getenv("DINDEL_DUMP_READS");                       // diagnostics: what each window hands to the likelihood step
        const std::string glfFile = outputPrefix + ".glf.txt";
        std::ofstream glfOutput(glfFile.c_str());
        if (!glfOutput.is_open()) throw std::string("Cannot open file ").append(glfFile).append(" for writing.");
        OutputData glfData = makeGLFOutputData(glfOutput);
        glfData.outputLine(glfData.headerString());                               // DInDel.cpp:1290-1291
        const double t_setup = seconds_since(t_start);

        long nWindows = 0, nSkipped = 0;
        std::vector<std::pair<double, long> > progress;                          // (seconds since start, windows written) after every batch
        double t_reduce_work = 0.0, t_prepare = 0.0, t_compute = 0.0, t_pack = 0.0, t_device = 0.0, t_unpack = 0.0, t_reduce = 0.0;
        std::mutex fatal_m, done_m;
        std::condition_variable done_cv;
        bool reduceDone = false;
        std::string fatal;
        std::atomic<int> fatalExit(1);
        BatchPool recycled;
        Channel toPrepare(size_t(prepareThreads) + 1);
        // prepared batches wait for the GPU in file order: room for one per prepare worker, so that a worker that was slow with the batch
        // at the head of the line (a descheduled thread on a busy host) does not idle the GPU while its successors are ready
        const long ahead = has("computeAhead") ? long(num("computeAhead", 0)) : long(std::max(computeThreads + 1, prepareThreads));
        // batches that are already waiting when an engine becomes free ride in one launch with the batch it takes (up to --mergeBatches of
        // them): the host stages keep their small batches, the GPU gets the larger launches it runs better (42.5 us per window in a launch of
        // 256 windows, 41.3 in one of 1,024, and one kernel boundary instead of four)
        // (default 4; 2 when the per-base alignments come back too — --faster, --outputRealignedBAM —: the page-locked result blocks of such a
        // launch are 0.4 GB each, and making them costs a 100,000-window run more than the launches save it)
        const int mergeBatches = std::max(1, int(num("mergeBatches", (has("faster") || has("outputRealignedBAM")) ? 2 : 4)));
        OrderedChannel toCompute(std::max(1L, ahead)), toReduce(long(computeThreads) * mergeBatches + 1);
        auto fail = [&](const std::string &s) {
            { std::lock_guard<std::mutex> lk(fatal_m); if (fatal.empty()) fatal = s; }
            toPrepare.abort(); toCompute.abort(); toReduce.abort();
        };

        const bool pooled = bamPaths.size() > 1 && !has("noLookBack");        // --noLookBack: diagnostics (every batch starts with an empty buffer)
        const uint32_t bufferSpan = 2u * uint32_t(libraries.getMaxInsertSize()) + 200u;    // a record fetched for a window has left the buffer this far on
        // ---- one window's read selection and haplotypes (the prepare workers; the writer's re-preparation behind a late skip) ----
        auto prepareWindow = [&](ReadFetcher &fetcher, WindowTask &T) {
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
            if (dumpReads) {
                std::ofstream df((std::string(dumpReads) + "." + std::to_string(T.index)).c_str());
                df.precision(17);
                for (size_t r = 0; r < T.reads.size(); r++) {
                    const Read &R = T.reads[r];
                    df << R.qname << " " << R.poolID << " " << int32_t(R.pos) << " " << R.mapQual << " " << R.matePos << " " << R.mateLen << " " << R.isUnmapped() << " " << R.isPaired()
                       << " " << R.mateIsUnmapped() << " " << R.mateIsReverse() << " " << R.mateSameTid << " " << R.posStat.first << " "
                       << (R.library ? R.library->getMaxInsertSize() : -1) << " " << R.seq.seq << "\n";
                }
            }
        };

        // ---- prepare: whole batches side by side, handed on in file order ----
        std::vector<double> t_prepare_of(size_t(prepareThreads), 0.0);
        std::vector<std::thread> prepareWorkers;
        for (int pt = 0; pt < prepareThreads; pt++) prepareWorkers.push_back(std::thread([&, pt]() {
            try {
                std::vector<std::unique_ptr<BamFile> > handles;
                std::vector<BamFile *> bams;
                for (size_t i = 0; i < bamPaths.size(); i++) { handles.push_back(std::unique_ptr<BamFile>(new BamFile(bamPaths[i]))); bams.push_back(handles.back().get()); }
                ReadFetcher fetcher(bams, libraries, rsp);
                std::vector<Read> replayed;
                BatchPtr b;
                while (toPrepare.pop(b)) {
                    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                    for (size_t i = 0; i < b->toRelease.size(); i++) fixture.release(b->toRelease[i]);
                    b->toRelease.clear();
                    std::string oldTid;
                    bool primed = false;
                    if (!b->lookBack.empty()) {               // several pools: bring the buffer to the state the windows in front left it in
                        fetcher.newChromosome();
                        for (size_t k = 0; k < b->lookBack.size(); k++) {
                            const Batch::Before &W = b->lookBack[k];
                            bool skipped = false;
                            try { fetcher.getReads(W.tid, W.leftPos, W.rightPos, replayed); } catch (std::string &) { skipped = true; }
                            fetcher.windowDone(skipped || (lateSkipsKnown && injectedLateSkips.count(W.index)), W.leftPos);
                        }
                        oldTid = b->lookBack.back().tid;
                        primed = true;
                    }
                    for (size_t i = 0; i < b->tasks.size(); i++) {
                        WindowTask &T = b->tasks[i];
                        if ((i == 0 && !primed) || T.tid != oldTid) { fetcher.newChromosome(); oldTid = T.tid; }     // DInDel.cpp:1327-1333
                        prepareWindow(fetcher, T);
                        fetcher.windowDone(T.skipped || (lateSkipsKnown && injectedLateSkips.count(T.index)), T.fileLeftPos);   // :1401-1408
                    }
                    t_prepare_of[size_t(pt)] += seconds_since(t0);
                    if (!toCompute.push(b)) break;
                }
            } catch (std::string &s) { fail(s); }
            catch (ReadFetcher::FatalError &e) { fatalExit = e.exitCode; fail(e.message); }   // the reference's exit() paths of getReads end the run here too
            catch (HaplotypeFixture::Error &e) { fail(e.message); }     // a malformed haplotype file ends the run, whichever window met it
            catch (std::exception &e) { fail(e.what()); }
        }));

        // ---- compute: every prepared window of a batch in one call; --computeThreads engines take batches in turn, so that one
        //      packs its batch (host) while the other's is on the GPU ----
        std::vector<double> t_compute_of(size_t(computeThreads), 0.0), t_pack_of(size_t(computeThreads), 0.0), t_device_of(size_t(computeThreads), 0.0),
            t_unpack_of(size_t(computeThreads), 0.0);
        std::vector<std::thread> computeWorkers;
        std::atomic<int> computeLeft(computeThreads);
        std::atomic<long> nLaunches(0);
        std::vector<double> t_ready_of(size_t(computeThreads), 0.0);         // when each engine had its device context, arena and streams
        for (int ct = 0; ct < computeThreads; ct++) computeWorkers.push_back(std::thread([&, ct]() {
            BatchPtr b;
            try {
                LikelihoodEngine engine(obs, devices[size_t(ct) % devices.size()]);
                engine.setThrowOnPositiveLikelihood(false);
                // diploidGLF reads scalars and covered flags only; the --faster model's indel count (DInDel.cpp:3529) needs hpos
                engine.setKeepAlignments(faster || realignedBAM);
                if (packThreads > 0) engine.setHostThreads(packThreads);
                if (!prepareOnly) engine.warmUp(size_t(batchWindows) * size_t(mergeBatches) * 8 * 200);     // while the first batches are being prepared
                t_ready_of[size_t(ct)] = seconds_since(t_start);
                std::vector<BatchPtr> group;
                std::vector<WindowJob> merged;
                bool open = true;
                while (open && toCompute.pop(b)) {
                    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                    group.clear();
                    group.push_back(std::move(b));
                    while (int(group.size()) < mergeBatches && toCompute.tryPop(b)) group.push_back(std::move(b));
                    size_t nJobs = 0;
                    for (size_t g = 0; g < group.size(); g++) {
                        Batch &B = *group[g];
                        B.jobOf.assign(B.tasks.size(), size_t(-1));
                        for (size_t i = 0; i < B.tasks.size(); i++) if (!B.tasks[i].skipped) {
                            WindowJob J;
                            J.haps = B.tasks[i].haps; J.reads = &B.tasks[i].reads; J.leftPos = B.tasks[i].leftPos; J.rightPos = B.tasks[i].rightPos;
                            B.jobOf[i] = B.jobs.size();
                            B.jobs.push_back(J);
                        }
                        nJobs += B.jobs.size();
                    }
                    if (nJobs > 0 && !prepareOnly) {
                        // one call for the group: the jobs travel through one vector and go back to their batches with their views
                        std::vector<WindowJob> *jobs = &group[0]->jobs;
                        if (group.size() > 1) {
                            merged.clear();
                            for (size_t g = 0; g < group.size(); g++)
                                for (size_t j = 0; j < group[g]->jobs.size(); j++) merged.push_back(std::move(group[g]->jobs[j]));
                            jobs = &merged;
                        }
                        if (faster) engine.computeLikelihoodsFasterBatch(*jobs); else engine.computeLikelihoodsBatch(*jobs);
                        if (group.size() > 1) {
                            size_t at = 0;
                            for (size_t g = 0; g < group.size(); g++)
                                for (size_t j = 0; j < group[g]->jobs.size(); j++) group[g]->jobs[j] = std::move(merged[at++]);
                            merged.clear();
                        }
                        t_pack_of[size_t(ct)] += engine.lastPackSeconds; t_device_of[size_t(ct)] += engine.lastDeviceSeconds;
                        t_unpack_of[size_t(ct)] += engine.lastUnpackSeconds;
                        nLaunches++;
                    }
                    t_compute_of[size_t(ct)] += seconds_since(t0);
                    for (size_t g = 0; g < group.size() && open; g++) if (!toReduce.push(group[g])) open = false;
                }
                group.clear();
                b.reset();
                if (--computeLeft == 0) toReduce.close();
                // the batches still being reduced hold views into this engine's result blocks: wait for the writer
                std::unique_lock<std::mutex> lk(done_m);
                done_cv.wait(lk, [&] { return reduceDone; });
            } catch (std::string &s) { fail(s); }
            catch (std::exception &e) { fail(e.what()); }
        }));

        // ---- one window's lines: diploidGLF (+ the realigned BAM), or the skipped-window line (the reduce helpers; the writer's
        //      re-preparation behind a late skip) ----
        auto reduceWindow = [&](WindowTask &T, const WindowJob *J) {
            std::ostringstream os;
            OutputData local = glfData;
            local.out = &os;
            if (!T.skipped) {
                try {
                    if (injectedLateSkips.count(T.index)) throw std::string("hapSize error.");       // tests (--injectLateSkip)
                    if (J) {
                        if (!J->error.empty()) throw std::string(J->error);
                        // like the reference, diploidGLF writes its lines as it goes: if it throws half-way ("genotyping
                        // error"), the lines already written stay and the skipped-window line follows them
                        diploidGLF(*T.haps, T.reads, J->result, T.pos, T.leftPos, T.rightPos, local, T.index, T.tid, T.candidates, dip, "dip");
                        if (realignedBAM) {                                      // DInDel.cpp:589-620
                            const std::pair<int, int> best = maxLikelihoodPair(*T.haps, T.reads, J->result, int(T.leftPos), T.candidates, dip);
                            std::vector<CIGAR> cigars;
                            realignedCigars(*T.haps, T.reads, J->result, best, int(T.leftPos), cigars);
                            std::vector<int> onHap(T.reads.size());
                            for (size_t r = 0; r < onHap.size(); r++) onHap[r] = J->result.onHap(r);
                            writeRealignedBAMFile(realignedBAMFileName(outputPrefix, T.index, T.tid, T.leftPos, T.rightPos, rsp.minReadOverlap),
                                                  cigars, T.reads, onHap, headerBam);
                        }
                    }
                } catch (std::string &s) {
                    T.message = skippedMessage(s);
                    T.skipped = true;
                    T.lateSkip = true;                                           // after read selection: the reference resets its read buffer behind it (:1404-1405)
                }
            }
            if (T.skipped) local.output(skippedWindowLine(local, T.message, T.index, T.tid, T.fileLeftPos, T.fileRightPos));
            T.lines = os.str();
        };

        // ---- reduce: windows of a batch side by side, each into its own buffer; written out in window order ----
        long nRePrepared = 0;
        std::thread reduceThread([&]() {
            BatchPtr b;
            // Several pools: the windows behind a late skip, re-prepared window by window from an empty buffer (see the header).  The
            // fetcher, its BAM handles and the engine are made when the first late skip of the run is met.
            struct Redo {
                bool active, first;
                std::string tid; uint64_t reach;
                std::vector<std::unique_ptr<BamFile> > handles; std::vector<BamFile *> bams;
                std::unique_ptr<ReadFetcher> fetcher; std::unique_ptr<LikelihoodEngine> engine;
                Redo() : active(false), first(false), reach(0) {}
            } redo;
            auto rePrepare = [&](WindowTask &T) {
                if (!redo.fetcher) {
                    for (size_t i = 0; i < bamPaths.size(); i++) { redo.handles.push_back(std::unique_ptr<BamFile>(new BamFile(bamPaths[i]))); redo.bams.push_back(redo.handles.back().get()); }
                    redo.fetcher.reset(new ReadFetcher(redo.bams, libraries, rsp));
                }
                if (redo.first) { redo.fetcher->newChromosome(); redo.reach = uint64_t(T.fileRightPos) + bufferSpan; redo.first = false; }   // the reset of DInDel.cpp:1404-1405
                T.skipped = false; T.lateSkip = false; T.message = "ok";
                prepareWindow(*redo.fetcher, T);
                std::vector<WindowJob> one;
                if (!T.skipped && !prepareOnly) {
                    if (!redo.engine) {
                        redo.engine.reset(new LikelihoodEngine(obs, devices[0]));
                        redo.engine->setThrowOnPositiveLikelihood(false);
                        redo.engine->setKeepAlignments(faster || realignedBAM);
                    }
                    WindowJob J;
                    J.haps = T.haps; J.reads = &T.reads; J.leftPos = T.leftPos; J.rightPos = T.rightPos;
                    one.push_back(J);
                    if (faster) redo.engine->computeLikelihoodsFasterBatch(one); else redo.engine->computeLikelihoodsBatch(one);
                }
                reduceWindow(T, one.empty() ? NULL : &one[0]);
                redo.fetcher->windowDone(T.skipped, T.fileLeftPos);
                nRePrepared++;
            };
            try {
                while (toReduce.pop(b)) {
                    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                    Batch &B = *b;
                    std::atomic<size_t> next(0);
                    // a helper thread must not let anything escape (std::terminate): whatever diploidGLF, the CIGAR step or the BAM writer
                    // throws beside the reference's strings ends the run through fail(), and the helpers stop taking windows
                    auto work = [&]() {
                        const std::chrono::steady_clock::time_point w0 = std::chrono::steady_clock::now();
                        try {
                        for (;;) {
                            const size_t i = next.fetch_add(1);
                            if (i >= B.tasks.size()) break;
                            reduceWindow(B.tasks[i], (!B.tasks[i].skipped && !prepareOnly) ? &B.jobs[B.jobOf[i]] : NULL);
                        }
                        } catch (std::exception &e) { next.store(B.tasks.size()); fail(std::string("reduce: ") + e.what()); }
                        catch (...) { next.store(B.tasks.size()); fail("reduce: unknown exception"); }
                        const double dt = seconds_since(w0);
                        std::lock_guard<std::mutex> lk(fatal_m);
                        t_reduce_work += dt;
                    };
                    std::vector<std::thread> pool;
                    const int nt = int(std::min<size_t>(size_t(reduceThreads), B.tasks.size()));
                    for (int t = 1; t < nt; t++) pool.push_back(std::thread(work));
                    work();
                    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
                    for (size_t i = 0; i < B.tasks.size(); i++) {
                        WindowTask &T = B.tasks[i];
                        if (windowByWindow) {
                            if (T.tid != redo.tid || !redo.fetcher) { redo.first = true; redo.tid = T.tid; }     // DInDel.cpp:1327-1333
                            rePrepare(T);
                        } else if (redo.active) {
                            // (a new chromosome resets the buffer in both histories; beyond `reach` no record of the reset's moment is left)
                            if ((redo.first && T.tid != redo.tid) || (!redo.first && (T.tid != redo.tid || uint64_t(T.fileLeftPos) >= redo.reach))) redo.active = false;
                            else rePrepare(T);
                        }
                        if (T.lateSkip && pooled && !lateSkipsKnown && !windowByWindow) { redo.active = true; redo.first = true; redo.tid = T.tid; }
                        if (T.skipped) {
                            std::cerr << "skipped " << T.tid << " " << T.pos << " reason: " << T.message << std::endl;     // DInDel.cpp:1383
                            nSkipped++;
                        }
                        glfOutput << T.lines;
                        nWindows++;
                    }
                    glfOutput.flush();
                    progress.push_back(std::make_pair(seconds_since(t_start), nWindows));
                    // the haplotypes of these windows can go: noted here, dropped by the prepare worker that takes the batch next
                    // (side by side with the others, not in this thread's serial part)
                    for (size_t i = 0; i < B.tasks.size(); i++) if (B.tasks[i].haps) { B.toRelease.push_back(B.tasks[i].index); B.tasks[i].haps = NULL; }
                    B.jobs.clear();                                               // drops the batch's views: its result block can be reused
                    B.jobOf.clear();
                    recycled.give(b);
                    t_reduce += seconds_since(t0);
                }
            } catch (std::string &s) { fail(s); }
            catch (ReadFetcher::FatalError &e) { fatalExit = e.exitCode; fail(e.message); }   // (the re-preparation behind a late skip reads the BAM files too)
            catch (HaplotypeFixture::Error &e) { fail(e.message); }
            catch (std::exception &e) { fail(e.what()); }
        });

        // ---- the window file, in file order ----
        int rc = 0;
        try {
            VariantFile vf(opt["varFile"]);
            int index = 0;
            long seq = 0;
            std::string oldTid("-1");
            uint32_t oldLeftPos = 0;
            BatchPtr batch = recycled.take();
            size_t nTasks = 0;                                                                // batch->tasks[nTasks...] are left-overs of an earlier use
            std::deque<Batch::Before> recent;                                                 // the chromosome's windows so far that a later batch may need
            auto flush = [&]() {
                batch->seq = seq++;
                batch->tasks.resize(nTasks);
                batch->lookBack.clear();
                if (pooled && nTasks) {
                    // windows of recent[] in front of the batch's first one; the replay starts at the last of them whose own fetch
                    // (everything up to rightPos + maxInsert) has left the buffer when the batch's first window is selected
                    const WindowTask &first = batch->tasks[0];
                    auto gone = [&](const Batch::Before &W) {          // W's own fetch has left the buffer by the batch's first window (or never was in it)
                        return W.tid != first.tid || uint64_t(W.rightPos) + bufferSpan <= uint64_t(first.fileLeftPos);
                    };
                    const size_t n = recent.size() - nTasks;                                   // recent[] ends with this batch's windows
                    size_t from = n;
                    while (from > 0 && recent[from - 1].tid == first.tid) { from--; if (gone(recent[from])) break; }
                    batch->lookBack.assign(recent.begin() + long(from), recent.begin() + long(n));
                    while (recent.size() > nTasks + 1 && gone(recent[1])) recent.pop_front();   // recent[0] stays a start later batches can use
                }
                const bool ok = toPrepare.push(batch);
                batch = recycled.take();
                nTasks = 0;
                return ok;
            };
            while (!vf.eof()) {
                AlignedCandidates cand = vf.getLineVector(oneBased);
                if (cand.variants.size() == 0) continue;
                if (cand.tid != oldTid) { oldTid = cand.tid; oldLeftPos = 0; }                // DInDel.cpp:1327-1333
                if (uint32_t(cand.leftPos) < oldLeftPos) {                                    // :1335-1339
                    std::cerr << "leftPos: " << uint32_t(cand.leftPos) << " oldLeftPos: " << oldLeftPos << std::endl;
                    std::cerr << "Candidate variant files must be sorted on left position of window!" << std::endl;
                    rc = 1;
                    break;
                }
                oldLeftPos = uint32_t(cand.leftPos);
                if (nTasks == batch->tasks.size()) batch->tasks.push_back(WindowTask());
                WindowTask &T = batch->tasks[nTasks++];
                T.lines.clear();
                T.candidates = cand; T.tid = cand.tid; T.pos = uint32_t(cand.centerPos);
                T.fileLeftPos = T.leftPos = uint32_t(cand.leftPos); T.fileRightPos = T.rightPos = uint32_t(cand.rightPos);
                T.haps = NULL; T.skipped = false; T.lateSkip = false; T.message = "ok";
                T.index = ++index;
                if (pooled) {
                    Batch::Before W = { T.tid, T.fileLeftPos, T.fileRightPos, T.index };
                    recent.push_back(W);
                }
                // the first batches are small (an eighth, a quarter, half of --batchWindows): the GPU gets its first windows while the bulk is
                // still being prepared, and the writer its first lines
                const int want = seq >= 3 ? batchWindows : std::max(1, batchWindows >> (3 - int(seq)));
                if (int(nTasks) >= want && !flush()) break;
            }
            if (nTasks) flush();
        } catch (std::string &s) { fail(s); }
        toPrepare.close();
        for (size_t t = 0; t < prepareWorkers.size(); t++) prepareWorkers[t].join();
        for (size_t t = 0; t < t_prepare_of.size(); t++) t_prepare += t_prepare_of[t];
        toCompute.close();
        // the reduce stage ends when every batch has come through; the compute workers are released after it
        reduceThread.join();
        { std::lock_guard<std::mutex> lk(done_m); reduceDone = true; }
        done_cv.notify_all();
        for (size_t t = 0; t < computeWorkers.size(); t++) computeWorkers[t].join();
        for (int t = 0; t < computeThreads; t++) { t_compute += t_compute_of[size_t(t)]; t_pack += t_pack_of[size_t(t)]; t_device += t_device_of[size_t(t)]; t_unpack += t_unpack_of[size_t(t)]; }
        glfOutput.close();
        if (!fatal.empty()) { std::cerr << "Exception: " << fatal << std::endl; return fatalExit.load(); }
        if (rc) return rc;
        if (!has("quiet")) std::cout << "windows: " << nWindows << " skipped: " << nSkipped << " -> " << glfFile << std::endl;
        if (!has("quiet") && nRePrepared) std::cout << "re-prepared behind late skips: " << nRePrepared << " windows" << std::endl;
        if (has("timing")) {
            const double wall = seconds_since(t_start);
            long peakKb = 0;                                                      // VmHWM of /proc/self/status
            {
                std::ifstream st("/proc/self/status");
                std::string line;
                while (std::getline(st, line)) if (line.compare(0, 6, "VmHWM:") == 0) peakKb = atol(line.c_str() + 6);
            }
            std::cout << "timing: wall=" << wall << " setup=" << t_setup << " prepare_threads=" << prepareThreads << " prepare=" << t_prepare << " compute_threads=" << computeThreads << " compute=" << t_compute << " (pack=" << t_pack
                      << " device=" << t_device << " unpack=" << t_unpack << ") reduce_threads=" << reduceThreads << " reduce=" << t_reduce << " (work=" << t_reduce_work << " summed over the threads)" << " peak_rss_mb=" << peakKb / 1024 << " windows_per_s=" << double(nWindows) / wall;
            // the rate once the pipeline is full: from the batch that completed the first fifth of the windows to the last one
            size_t from = 0;
            while (from + 1 < progress.size() && progress[from].second * 5 < nWindows) from++;
            if (from + 1 < progress.size() && progress.back().first > progress[from].first)
                std::cout << " steady_windows_per_s=" << double(progress.back().second - progress[from].second) / (progress.back().first - progress[from].first);
            std::cout << " launches=" << nLaunches.load();
            // when the first batch and the first 1 / 5 / 20 / 50 / 100 % of the windows were written (seconds since start)
            std::cout << " engines_ready_at=";
            for (size_t i = 0; i < t_ready_of.size(); i++) std::cout << (i ? "," : "") << t_ready_of[i];
            std::cout << " written_at=";
            size_t at = 0;
            if (!progress.empty()) std::cout << progress[0].first << "(first)";
            for (int pc : {1, 5, 20, 50, 100}) {
                while (at + 1 < progress.size() && progress[at].second * 100 < nWindows * pc) at++;
                if (!progress.empty()) std::cout << "/" << progress[at].first;
            }
            std::cout << std::endl;
        }
    } catch (std::string &s) {
        std::cerr << "Exception: " << s << std::endl;
        return 1;
    }
    return 0;
}

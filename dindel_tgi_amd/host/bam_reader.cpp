// bam_reader.cpp — see bam_reader.hpp.
#include "bam_reader.hpp"
#include "fast_inflate.hpp"
#include <algorithm>
#include <cstring>
#include <zlib.h>

namespace dindel {

namespace {
inline uint16_t le16(const uint8_t *p) { return uint16_t(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
inline uint64_t le64(const uint8_t *p) { return uint64_t(le32(p)) | (uint64_t(le32(p + 4)) << 32); }
const int kMaxBin = 37450;               // ((1 << 18) - 1) / 7 + 1: the pseudo-bin holding the index metadata
const int kLidxShift = 14;
}

// ---------------- BGZF ----------------
BgzfReader::BgzfReader() : f(NULL), blockAddress(-1), blockLength(0), nextAddress(0), block(NULL), offset(0), cacheNext(0), zs(NULL)
{
    for (int i = 0; i < kCachedBlocks; i++) { cache[i].address = -1; cache[i].next = 0; cache[i].length = 0; }
}

BgzfReader::~BgzfReader()
{
    if (f) fclose(f);
    if (zs) { inflateEnd(static_cast<z_stream *>(zs)); delete static_cast<z_stream *>(zs); }
}

void BgzfReader::open(const std::string &path)
{
    f = fopen(path.c_str(), "rb");
    if (!f) throw std::string("Cannot open BAM file.");
    blockAddress = -1; blockLength = 0; offset = 0; nextAddress = 0;
}

bool BgzfReader::loadBlock(int64_t address)
{
    for (int i = 0; i < kCachedBlocks; i++) if (cache[i].address == address) {
        blockAddress = address; blockLength = cache[i].length; nextAddress = cache[i].next; block = cache[i].data.data(); offset = 0;
        return true;
    }
    uint8_t hdr[18];
    if (fseek(f, long(address), SEEK_SET) != 0) return false;
    if (fread(hdr, 1, 18, f) != 18) return false;
    if (hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) throw std::string("not a BGZF block");
    const int xlen = le16(hdr + 10);
    // extra subfields: find 'B','C' (the first one in every BGZF file: hdr[12..17] when xlen == 6)
    int bsize = -1;
    if (xlen == 6 && hdr[12] == 'B' && hdr[13] == 'C' && le16(hdr + 14) == 2) bsize = le16(hdr + 16);
    else {
        std::vector<uint8_t> extra(size_t(xlen), 0);
        memcpy(extra.data(), hdr + 12, size_t(std::min(xlen, 6)));
        if (xlen > 6 && fread(extra.data() + 6, 1, size_t(xlen - 6), f) != size_t(xlen - 6)) return false;
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = le16(extra.data() + i + 2);
            if (extra[size_t(i)] == 'B' && extra[size_t(i + 1)] == 'C' && slen == 2) bsize = le16(extra.data() + i + 4);
            i += 4 + slen;
        }
    }
    if (bsize < 0) throw std::string("BGZF block without a BC field");
    const int clen = bsize - xlen - 19;                   // compressed payload; then CRC32 and ISIZE
    if (clen < 0) throw std::string("corrupt BGZF block");
    comp.resize(size_t(clen) + 8);
    if (fread(comp.data(), 1, comp.size(), f) != comp.size()) return false;
    const uint32_t isize = le32(comp.data() + clen + 4);
    if (isize > 65536u) throw std::string("corrupt BGZF block");      // a BGZF block holds at most 64 KiB
    CachedBlock &C = cache[cacheNext];
    C.address = -1;                                       // not valid while it is being filled (an exception leaves it so)
    if (C.data.size() < size_t(isize) + 8) C.data.resize(size_t(isize) + 8);      // 8 bytes of slack for fastInflate's block copies
    if (isize) {
        const uint32_t want_crc = le32(comp.data() + clen);
        // the own decoder first (fast_inflate.hpp); whatever it declines, or decodes to bytes the CRC does not accept, goes to zlib
        bool done = fastInflate(comp.data(), size_t(clen), C.data.data(), size_t(isize)) && fastCrc32(0, C.data.data(), isize) == want_crc;
        if (!done) {
            if (!zs) {
                z_stream *z = new z_stream;
                memset(z, 0, sizeof(*z));
                if (inflateInit2(z, -15) != Z_OK) { delete z; throw std::string("zlib: inflateInit2 failed"); }
                zs = z;
            } else if (inflateReset(static_cast<z_stream *>(zs)) != Z_OK) throw std::string("zlib: inflateReset failed");
            z_stream *z = static_cast<z_stream *>(zs);
            z->next_in = comp.data(); z->avail_in = uInt(clen);
            z->next_out = C.data.data(); z->avail_out = uInt(isize);
            if (inflate(z, Z_FINISH) != Z_STREAM_END) throw std::string("zlib: corrupt BGZF block");
            if (fastCrc32(0, C.data.data(), isize) != want_crc) throw std::string("BGZF block: CRC mismatch");
        }
    }
    C.address = address; C.length = int(isize); C.next = address + bsize + 1;
    cacheNext = (cacheNext + 1) % kCachedBlocks;
    blockAddress = address; blockLength = int(isize); nextAddress = C.next; block = C.data.data(); offset = 0;
    return true;
}

void BgzfReader::seek(uint64_t voffset)
{
    const int64_t address = int64_t(voffset >> 16);
    const int within = int(voffset & 0xffff);
    if (address != blockAddress || block == NULL) {
        if (!loadBlock(address)) { blockAddress = address; blockLength = 0; nextAddress = address; }
    }
    offset = within;
}

bool BgzfReader::read(void *dst, size_t n)
{
    uint8_t *out = static_cast<uint8_t *>(dst);
    while (n > 0) {
        if (blockAddress < 0 || offset >= blockLength) {
            const int64_t address = blockAddress < 0 ? 0 : nextAddress;
            if (!loadBlock(address)) return false;
            if (blockLength == 0) { if (feof(f)) return false; continue; }   // empty block (the EOF marker, or padding)
        }
        const size_t take = std::min(n, size_t(blockLength - offset));
        memcpy(out, block + offset, take);
        out += take; offset += int(take); n -= take;
        if (offset == blockLength && n == 0) {
            // like bgzf_tell after a read that ends exactly at a block boundary: point at the start of the next block
            blockAddress = nextAddress; blockLength = 0; offset = 0;
            const int64_t keep = nextAddress;
            if (!loadBlock(keep)) { blockAddress = keep; blockLength = 0; nextAddress = keep; }
        }
    }
    return true;
}

// ---------------- BGZF output ----------------
void BgzfWriter::open(const std::string &path)
{
    f = fopen(path.c_str(), "wb");
    if (!f) throw std::string("Cannot open bamfile ").append(path).append(" for writing!");
    pending.clear();
}

void BgzfWriter::flushBlock()
{
    const size_t n = pending.size();                       // may be 0: the end-of-file marker
    std::vector<uint8_t> comp(compressBound(uLong(n)) + 64);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::string("zlib: deflateInit2 failed");
    zs.next_in = n ? pending.data() : NULL; zs.avail_in = uInt(n);
    zs.next_out = comp.data(); zs.avail_out = uInt(comp.size());
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = comp.size() - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26 > 65536) throw std::string("zlib: deflate failed on a BGZF block");
    const uint32_t bsize = uint32_t(clen + 25);           // total block size - 1
    const uint32_t crc = fastCrc32(0, pending.data(), n);
    uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, uint8_t(bsize & 255), uint8_t(bsize >> 8)};
    uint8_t tail[8] = {uint8_t(crc), uint8_t(crc >> 8), uint8_t(crc >> 16), uint8_t(crc >> 24), uint8_t(n), uint8_t(n >> 8), uint8_t(n >> 16), uint8_t(n >> 24)};
    if (fwrite(hdr, 1, 18, f) != 18 || fwrite(comp.data(), 1, clen, f) != clen || fwrite(tail, 1, 8, f) != 8) throw std::string("Error writing BGZF block.");
    pending.clear();
}

void BgzfWriter::write(const void *src, size_t n)
{
    const uint8_t *p = static_cast<const uint8_t *>(src);
    const size_t kBlock = 0xff00;
    while (n > 0) {
        const size_t take = std::min(n, kBlock - pending.size());
        pending.insert(pending.end(), p, p + take);
        p += take; n -= take;
        if (pending.size() == kBlock) flushBlock();
    }
}

void BgzfWriter::close()
{
    if (!f) return;
    if (!pending.empty()) flushBlock();
    flushBlock();                                          // the empty end-of-file block
    FILE *g = f;
    f = NULL;
    if (fclose(g) != 0) throw std::string("Error closing BGZF file.");
}

BamWriter::BamWriter(const std::string &path, const BamFile &like)
{
    out.open(path);
    uint8_t w[4];
    auto put32 = [&](uint32_t v) { w[0] = uint8_t(v); w[1] = uint8_t(v >> 8); w[2] = uint8_t(v >> 16); w[3] = uint8_t(v >> 24); out.write(w, 4); };
    out.write("BAM\1", 4);
    const std::string &text = like.headerTextAsStored();
    put32(uint32_t(text.size()));
    out.write(text.data(), text.size());
    put32(uint32_t(like.targetNames().size()));
    for (size_t i = 0; i < like.targetNames().size(); i++) {
        const std::string &nm = like.targetNames()[i];
        put32(uint32_t(nm.size() + 1));
        out.write(nm.c_str(), nm.size() + 1);
        put32(uint32_t(like.targetLengths()[i]));
    }
}

void BamWriter::write(const std::vector<uint8_t> &record)
{
    const uint32_t v = uint32_t(record.size());
    const uint8_t w[4] = {uint8_t(v), uint8_t(v >> 8), uint8_t(v >> 16), uint8_t(v >> 24)};
    out.write(w, 4);
    out.write(record.data(), record.size());
}

// ---------------- BAM record ----------------
uint32_t BamRecord::calend() const
{
    uint32_t end = uint32_t(pos);
    for (size_t k = 0; k < cigar.size(); k++) {
        const int op = int(cigar[k] & 15u);
        if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CREF_SKIP || op == BAM_CEQUAL || op == BAM_CDIFF) end += cigar[k] >> 4;
    }
    return end;
}

const char *BamRecord::auxString(const char tag[2]) const
{
    size_t i = 0;
    while (i + 3 <= aux.size()) {
        const bool hit = aux[i] == uint8_t(tag[0]) && aux[i + 1] == uint8_t(tag[1]);
        const char type = char(aux[i + 2]);
        i += 3;
        size_t len;
        if (type == 'A' || type == 'c' || type == 'C') len = 1;
        else if (type == 's' || type == 'S') len = 2;
        else if (type == 'i' || type == 'I' || type == 'f') len = 4;
        else if (type == 'd') len = 8;
        else if (type == 'Z' || type == 'H') {
            const size_t start = i;
            while (i < aux.size() && aux[i]) i++;
            if (hit && type == 'Z' && i < aux.size()) return reinterpret_cast<const char *>(aux.data() + start);
            i++;
            continue;
        } else if (type == 'B') {
            if (i + 5 > aux.size()) return NULL;
            const char sub = char(aux[i]);
            const uint32_t cnt = le32(aux.data() + i + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            len = 5 + es * cnt;
        } else return NULL;
        i += len;
    }
    return NULL;
}

// ---------------- BAM file + BAI ----------------
BamFile::BamFile(const std::string &path) : fileName(path)
{
    bgzf.open(path);
    uint8_t magic[4], w[4];
    if (!bgzf.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) throw std::string("Cannot open BAM file.");
    if (!bgzf.read(w, 4)) throw std::string("truncated BAM header");
    const uint32_t l_text = le32(w);
    text.resize(l_text);
    if (l_text && !bgzf.read(&text[0], l_text)) throw std::string("truncated BAM header");
    rawText = text;
    while (!text.empty() && text[text.size() - 1] == 0) text.erase(text.size() - 1);
    if (!bgzf.read(w, 4)) throw std::string("truncated BAM header");
    const uint32_t n_ref = le32(w);
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!bgzf.read(w, 4)) throw std::string("truncated BAM header");
        const uint32_t l_name = le32(w);
        std::string nm(l_name, 0);
        if (l_name && !bgzf.read(&nm[0], l_name)) throw std::string("truncated BAM header");
        while (!nm.empty() && nm[nm.size() - 1] == 0) nm.erase(nm.size() - 1);
        if (!bgzf.read(w, 4)) throw std::string("truncated BAM header");
        strToTID[nm] = int(names.size());
        names.push_back(nm);
        lengths.push_back(int32_t(le32(w)));
    }
    // @RG lines: ID -> LB (sam_header2tbl(dict, "RG", "ID", "LB"))
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(p, e - p);
        p = e + 1;
        if (line.compare(0, 3, "@RG") != 0) continue;
        std::string id, lb;
        size_t q = 3;
        while (q < line.size()) {
            size_t t = line.find('\t', q + 1);
            if (t == std::string::npos) t = line.size();
            const std::string fld = line.substr(q + 1, t - q - 1);
            if (fld.compare(0, 3, "ID:") == 0) id = fld.substr(3);
            if (fld.compare(0, 3, "LB:") == 0) lb = fld.substr(3);
            q = t;
        }
        if (!id.empty() && !lb.empty() && rg2lib.find(id) == rg2lib.end()) rg2lib[id] = lb;
    }
    loadIndex(path);
}

int BamFile::getTID(const std::string &name) const
{
    std::map<std::string, int>::const_iterator it = strToTID.find(name);
    if (it == strToTID.end()) throw std::string("Cannot find ID!");
    return it->second;
}

const char *BamFile::getLibrary(const BamRecord &b) const
{
    const char *rg = b.auxString("RG");
    if (!rg) return NULL;
    std::map<std::string, std::string>::const_iterator it = rg2lib.find(rg);
    return it == rg2lib.end() ? NULL : it->second.c_str();
}

void BamFile::loadIndex(const std::string &path)
{
    FILE *fi = fopen((path + ".bai").c_str(), "rb");
    if (!fi && path.size() > 4) fi = fopen((path.substr(0, path.size() - 4) + ".bai").c_str(), "rb");     // file.bam -> file.bai
    if (!fi) throw std::string("Cannot open BAM index.");
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fi)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(fi);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) throw std::string("not a BAI index");
    size_t o = 4;
    const uint32_t n_ref = le32(d.data() + o); o += 4;
    index.resize(n_ref);
    for (uint32_t r = 0; r < n_ref; r++) {
        if (o + 4 > d.size()) throw std::string("truncated BAI index");
        const uint32_t n_bin = le32(d.data() + o); o += 4;
        for (uint32_t b = 0; b < n_bin; b++) {
            if (o + 8 > d.size()) throw std::string("truncated BAI index");
            const uint32_t bin = le32(d.data() + o), n_chunk = le32(d.data() + o + 4); o += 8;
            if (o + 16ull * n_chunk > d.size()) throw std::string("truncated BAI index");
            std::vector<Chunk> &v = index[r].bins[bin];
            for (uint32_t c = 0; c < n_chunk; c++, o += 16) { Chunk ch; ch.beg = le64(d.data() + o); ch.end = le64(d.data() + o + 8); v.push_back(ch); }
        }
        if (o + 4 > d.size()) throw std::string("truncated BAI index");
        const uint32_t n_intv = le32(d.data() + o); o += 4;
        if (o + 8ull * n_intv > d.size()) throw std::string("truncated BAI index");
        for (uint32_t i = 0; i < n_intv; i++, o += 8) index[r].linear.push_back(le64(d.data() + o));
    }
}

std::vector<BamFile::Chunk> BamFile::chunksFor(int tid, int ibeg, int iend) const
{
    std::vector<Chunk> off;
    if (tid < 0 || size_t(tid) >= index.size()) return off;
    uint32_t beg = uint32_t(ibeg), end = uint32_t(iend);
    if (beg >= end) return off;                                    // reg2bins: nothing
    if (end >= 1u << 29) end = 1u << 29;
    const RefIndex &R = index[size_t(tid)];
    // bins overlapping [beg, end) — reg2bins
    std::vector<uint32_t> bins;
    {
        const uint32_t e = end - 1;
        bins.push_back(0);
        for (uint32_t k = 1 + (beg >> 26); k <= 1 + (e >> 26); ++k) bins.push_back(k);
        for (uint32_t k = 9 + (beg >> 23); k <= 9 + (e >> 23); ++k) bins.push_back(k);
        for (uint32_t k = 73 + (beg >> 20); k <= 73 + (e >> 20); ++k) bins.push_back(k);
        for (uint32_t k = 585 + (beg >> 17); k <= 585 + (e >> 17); ++k) bins.push_back(k);
        for (uint32_t k = 4681 + (beg >> 14); k <= 4681 + (e >> 14); ++k) bins.push_back(k);
    }
    uint64_t min_off = 0;
    if (!R.linear.empty()) {
        const size_t w = size_t(beg >> kLidxShift);
        min_off = w >= R.linear.size() ? R.linear.back() : R.linear[w];
        if (min_off == 0) {                                        // index files whose leading windows are empty
            size_t nn = std::min(w, R.linear.size());
            while (nn > 0 && R.linear[nn - 1] == 0) nn--;
            if (nn > 0) min_off = R.linear[nn - 1];
        }
    }
    for (size_t i = 0; i < bins.size(); i++) {
        if (int(bins[i]) >= kMaxBin) continue;
        std::map<uint32_t, std::vector<Chunk> >::const_iterator it = R.bins.find(bins[i]);
        if (it == R.bins.end()) continue;
        for (size_t j = 0; j < it->second.size(); j++)
            if (it->second[j].end > min_off) off.push_back(it->second[j]);
    }
    if (off.empty()) return off;
    std::sort(off.begin(), off.end());
    // chunks completely contained in their predecessor, overlaps from the indexer's merging, then adjacent chunks
    size_t l = 0;
    for (size_t i = 1; i < off.size(); i++) if (off[l].end < off[i].end) off[++l] = off[i];
    off.resize(l + 1);
    for (size_t i = 1; i < off.size(); i++) if (off[i - 1].end >= off[i].beg) off[i - 1].end = off[i].beg;
    l = 0;
    for (size_t i = 1; i < off.size(); i++) {
        if ((off[l].end >> 16) == (off[i].beg >> 16)) off[l].end = off[i].end; else off[++l] = off[i];
    }
    off.resize(l + 1);
    return off;
}

bool BamFile::nextRaw(std::vector<uint8_t> &d)
{
    uint8_t w[4];
    if (!bgzf.read(w, 4)) return false;
    const uint32_t block_size = le32(w);
    if (block_size < 32) throw std::string("corrupt BAM record");
    d.resize(block_size);
    return bgzf.read(d.data(), block_size);
}

void BamFile::decodeCore(const std::vector<uint8_t> &d, BamRecord &b)
{
    b.tid = int32_t(le32(d.data())); b.pos = int32_t(le32(d.data() + 4));
    const uint32_t l_read_name = d[8];
    b.qual = d[9]; b.bin = le16(d.data() + 10);
    const uint32_t n_cigar = le16(d.data() + 12);
    b.flag = le16(d.data() + 14);
    b.l_qseq = int32_t(le32(d.data() + 16)); b.mtid = int32_t(le32(d.data() + 20)); b.mpos = int32_t(le32(d.data() + 24)); b.isize = int32_t(le32(d.data() + 28));
    size_t o = 32;
    const size_t need = o + l_read_name + 4ull * n_cigar + (size_t(b.l_qseq) + 1) / 2 + size_t(b.l_qseq);
    if (b.l_qseq < 0 || need > d.size()) throw std::string("corrupt BAM record");
    o += l_read_name;
    b.cigar.resize(n_cigar);
    for (uint32_t k = 0; k < n_cigar; k++, o += 4) b.cigar[k] = le32(d.data() + o);
}

void BamFile::decodeRest(const std::vector<uint8_t> &d, BamRecord &b)
{
    const uint32_t l_read_name = d[8];
    size_t o = 32;
    b.qname.assign(reinterpret_cast<const char *>(d.data() + o), l_read_name ? l_read_name - 1 : 0);
    o += l_read_name + 4 * b.cigar.size();
    static const char nt16[] = "=ACMGRSVTWYHKDBN";               // bam_nt16_rev_table
    b.seq.resize(size_t(b.l_qseq));
    for (int32_t x = 0; x < b.l_qseq; x++) b.seq[size_t(x)] = nt16[(d[o + size_t(x >> 1)] >> ((~x & 1) << 2)) & 15];
    o += (size_t(b.l_qseq) + 1) / 2;
    b.qualities.assign(d.begin() + long(o), d.begin() + long(o) + b.l_qseq); o += size_t(b.l_qseq);
    b.aux.assign(d.begin() + long(o), d.end());
}

bool BamFile::next(BamRecord &b)
{
    if (!nextRaw(raw)) return false;
    decodeCore(raw, b);
    decodeRest(raw, b);
    return true;
}

} // namespace dindel

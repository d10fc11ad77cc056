"""TEST INFRASTRUCTURE ONLY: a Python-3 restatement of the reference's glf -> VCF converter, used to check
dindel_tgi_amd/host/glf_to_vcf.cpp.  It follows python/mergeOutputDiploid.py (getVCFString :35-154, processDiploidGLFFile
:158-231, mergeOutput :240-317) and its helpers python/utils/{Fasta.py :33-53, AnalyzeSequence.py, Variant.py :3-27,
FileUtils.py :33-125} statement by statement, written independently of the C++ (dict / list idioms of the script kept).

Parity unpinned: the reference's scripts are Python 2 and cannot be imported in this Python-3-only image, and the reference
ships no .glf.txt / VCF fixtures; the expected lines the tests spell out literally are hand-derived from the cited rules."""
import os


class Fasta:
    def __init__(self, fname):
        self.fa = open(fname, "rb")
        self.ft = {}
        for line in open(fname + ".fai"):
            dat = line.split()
            if len(dat) == 5:
                self.ft[dat[0]] = tuple(int(v) for v in dat[1:])

    def get(self, tid, pos1based, n):
        pos = pos1based - 1
        if tid not in self.ft:
            raise NameError("KeyError")
        _len, offset, blen, llen = self.ft[tid]
        fpos = offset + (pos // blen) * llen + (pos % blen)
        self.fa.seek(fpos, 0)
        seq = []
        while len(seq) < n:
            ch = self.fa.read(1)
            if ch == b"":
                break                       # the script would loop for ever at end of file
            if ch != b"\n":
                seq.append(ch.decode())
        return seq


def homopolymer_length(seq, pos):
    hp_len = 1
    for i in range(pos + 1, len(seq)):
        if seq[i] == seq[i - 1]:
            hp_len += 1
        else:
            break
    for i in range(pos - 1, 0, -1):
        if seq[i] == seq[i + 1]:
            hp_len += 1
        else:
            break
    return hp_len


class Variant:
    def __init__(self, s):
        n = len(s)
        if s[0] == "-" and n > 1:
            self.type, self.seq, self.length = "del", s[1:], n - 1
        elif s[0] == "+" and n > 1:
            self.type, self.seq, self.length = "ins", s[1:], n - 1
        elif n == 4 and s[1:3] == "=>":
            self.type, self.seq, self.length = "snp", s[3], 1
        elif s[0] == "*" or s.find("REF") != -1 or s.find("ref") != -1:
            self.type, self.seq, self.length = "ref", "", 0
        else:
            raise NameError("Unrecognized variant: " + s)


def get_vcf_string(glf, fa, maxHPLen=10, filterQual=0):
    filters = []
    pos = int(glf["pos"])
    chrom = glf["chr"]
    seq = fa.get(chrom, pos + 1 - 25, 50)
    hplen = homopolymer_length(seq, 25)
    report_pos = pos
    max_del_len = 0
    for gta in set(glf["nref_all"]):
        var = Variant(gta)
        if var.type == "del" and var.length > max_del_len:
            max_del_len = var.length
    refseq = "".join(fa.get(chrom, report_pos, 1 + max_del_len))
    altseqs, altseq_to_type = [], {}
    for gta in glf["nref_all"]:
        v = Variant(gta)
        g_code = -1
        if v.type == "del":
            g_altseq = refseq[0] + refseq[(1 + v.length):]
        elif v.type == "ins":
            g_altseq = refseq[0] + v.seq + refseq[1:]
        elif v.type == "snp":
            g_altseq = refseq[0] + v.seq[0] + refseq[2:]
        else:
            g_altseq = refseq[:]
            g_code = 0
        if g_code == -1 and g_altseq not in altseqs:
            altseqs.append(g_altseq)
            altseq_to_type[g_altseq] = v.type
    gtd = glf["genotype"].split(":")
    rec_gt = "%s:%d" % (gtd[0], int(float(gtd[1])))
    if all(altseq_to_type[a] == "snp" for a in altseqs):
        report_pos += 1
        refseq = "".join(fa.get(chrom, report_pos, 1))
        altseqs = [a[1:] for a in altseqs]
    if hplen > maxHPLen:
        filters.append("hp%d" % maxHPLen)
    if glf["qual"] < filterQual:
        filters.append("q%d" % filterQual)
    altseqs = ["<DEL>" if a.find("D") != -1 else a for a in altseqs]
    filterStr = "PASS" if not filters else ";".join(filters)
    infoStr = "DP=%d;NF=%d;NR=%d;NRS=%d;NFS=%d;HP=%d" % (int(glf["num_hap_reads"]), int(glf["num_cover_forward"]), int(glf["num_cover_reverse"]),
                                                         int(glf["num_cover_forward_old"]), int(glf["num_cover_reverse_old"]), hplen)
    rstr = "%s\t%s\t.\t%s\t%s\t%s\t%s\t%s\t%s\t%s" % (chrom, report_pos, refseq, ",".join(altseqs), "%s" % glf["qual"], filterStr, infoStr, "GT:GQ", rec_gt)
    return rstr, report_pos


def process_glf_file(glfFile, variants, fa, filterQual=20):
    f = open(glfFile)
    labels = f.readline().rstrip("\n").rstrip().split(" ")
    while True:
        line = f.readline().rstrip("\n").rstrip(" ").split(" ")
        if line == [""]:
            break
        assert len(line) == len(labels)
        dat = dict(zip(labels, line))
        if dat["msg"] != "ok" or dat["analysis_type"] != "dip.map" or dat["was_candidate_in_window"] != "1":
            continue
        glf = {"chr": dat["tid"], "pos": dat["realigned_position"], "qual": int(float(dat["qual"]))}
        if float(glf["qual"]) < 1.0:
            continue
        glf["nref_all"] = dat["nref_all"].split(",")
        if glf["nref_all"] == ["R=>D"]:
            continue
        glf["num_cover_forward"] = int(dat["var_coverage_forward"].split(",")[0])
        glf["num_cover_reverse"] = int(dat["var_coverage_reverse"].split(",")[0])
        glf["num_cover_forward_old"] = int(dat["num_cover_forward"])
        glf["num_cover_reverse_old"] = int(dat["num_cover_reverse"])
        glf["num_hap_reads"] = dat["num_reads"]
        glf["genotype"] = dat["glf"]
        vcf_str, report_pos = get_vcf_string(glf, fa, filterQual=filterQual)      # maxHPLen is not passed on: always 10 (:219)
        variants.setdefault(dat["tid"], {}).setdefault(report_pos, []).append(vcf_str)


HEADER = ['##fileformat=VCFv4.0', '##source=Dindel', '##reference=%(ref)s',
          '##INFO=<ID=DP,Number=1,Type=Integer,Description="Total number of reads in haplotype window">',
          '##INFO=<ID=HP,Number=1,Type=Integer,Description="Reference homopolymer tract length">',
          '##INFO=<ID=NF,Number=1,Type=Integer,Description="Number of reads covering non-ref variant on forward strand">',
          '##INFO=<ID=NR,Number=1,Type=Integer,Description="Number of reads covering non-ref variant on reverse strand">',
          '##INFO=<ID=NFS,Number=1,Type=Integer,Description="Number of reads covering non-ref variant site on forward strand">',
          '##INFO=<ID=NRS,Number=1,Type=Integer,Description="Number of reads covering non-ref variant site on reverse strand">',
          '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
          '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype quality">',
          '##ALT=<ID=DEL,Description="Deletion">',
          '##FILTER=<ID=q%(fq)d,Description="Quality below %(fq)d">',
          '##FILTER=<ID=hp%(hp)d,Description="Reference homopolymer length was longer than %(hp)d">',
          '##FILTER=<ID=fr0,Description="Non-ref allele is not covered by at least one read on both strands">',
          '##FILTER=<ID=wv,Description="Other indel in window had higher likelihood">',
          '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%(sample)s']


def merge_output(glfFilesFile, sampleID, refFile, maxHPLen, vcfFile, filterQual=20):
    files = [l.rstrip("\n").split()[0] for l in open(glfFilesFile)]
    assert all(os.path.exists(f) for f in files)
    out = [h % dict(ref=refFile, fq=filterQual, hp=maxHPLen, sample=sampleID) for h in HEADER]
    fa = Fasta(refFile)
    variants = {}
    for gf in files:
        process_glf_file(gf, variants, fa, filterQual=filterQual)
    chroms = [str(v) for v in range(1, 23)] + ["X", "Y"]
    chroms += sorted(c for c in variants if c not in chroms)      # the script: Python-2 dict order for these; the C++ sorts them
    for c in chroms:
        if c in variants:
            for pos in sorted(variants[c]):
                out.extend(variants[c][pos])
    open(vcfFile, "w").write("\n".join(out) + "\n")

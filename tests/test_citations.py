"""Every `File.ext:line[-line]` citation of a reference file in the public header, the kernels' headers, the oracle and the
docs must point inside that file (only checkable where the reference tree is mounted)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FILES = ["include/dindel_hmm.h", "oracle/dd_oracle.c", "oracle/dd_oracle.h", "oracle/ref_bits.cpp", "DESIGN.md", "INTEGRATION.md",
         "dindel_tgi_amd/csrc/hmm_kernel.hip", "dindel_tgi_amd/csrc/faster_kernel.hip", "dindel_tgi_amd/csrc/genotype_kernel.hip",
         "dindel_tgi_amd/csrc/capi.cpp", "dindel_tgi_amd/host/compute_likelihoods.hpp", "dindel_tgi_amd/host/compute_likelihoods.cpp",
         "dindel_tgi_amd/host/genotype.hpp", "dindel_tgi_amd/host/genotype.cpp", "dindel_tgi_amd/host/cigar.hpp",
         "dindel_tgi_amd/host/cigar.cpp", "dindel_tgi_amd/host/dindel_types.hpp", "dindel_tgi_amd/host/glf_output.hpp",
         "dindel_tgi_amd/host/glf_to_vcf.hpp", "dindel_tgi_amd/host/glf_to_vcf.cpp", "dindel_tgi_amd/host/bam_reader.hpp",
         "dindel_tgi_amd/host/window_io.hpp", "dindel_tgi_amd/host/window_io.cpp", "dindel_tgi_amd/host/get_reads.hpp",
         "dindel_tgi_amd/host/get_reads.cpp", "dindel_tgi_amd/host/diploid_glf.hpp", "dindel_tgi_amd/host/diploid_glf.cpp",
         "dindel_tgi_amd/host/dindel_gpu.cpp", "dindel_tgi_amd/host/dindel_glf2vcf.cpp", "tests/_vcf_oracle.py", "profiles/r02/instruction_mix.md",
         "dindel_tgi_amd/host/realigned_bam.hpp", "dindel_tgi_amd/host/realigned_bam.cpp", "dindel_tgi_amd/host/bam_reader.cpp",
         "tests/_getreads_oracle.py", "profiles/r02/n2_pipeline.md"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_reference_citations_point_inside_the_files():
    nlines = {}
    for f in os.listdir(REF):
        p = os.path.join(REF, f)
        if os.path.isfile(p):
            nlines[f] = sum(1 for _ in open(p, errors="ignore"))
    pydir = os.path.join(REF, "python")
    for f in os.listdir(pydir) if os.path.isdir(pydir) else []:
        if os.path.isfile(os.path.join(pydir, f)):
            nlines["python/" + f] = sum(1 for _ in open(os.path.join(pydir, f), errors="ignore"))
    utils = os.path.join(pydir, "utils")
    for f in os.listdir(utils) if os.path.isdir(utils) else []:      # python/utils/Fasta.py is cited as "Fasta.py:…" / "python/utils/Fasta.py:…"
        if os.path.isfile(os.path.join(utils, f)) and f not in nlines:
            nlines[f] = sum(1 for _ in open(os.path.join(utils, f), errors="ignore"))
    pat = re.compile(r"((?:python/)?[A-Za-z][A-Za-z0-9_]*\.(?:cpp|hpp|py|h))`?:(\d+)(?:-(\d+))?")
    bad, n = [], 0
    for rel in FILES:
        for m in pat.finditer(open(os.path.join(ROOT, rel)).read()):
            name, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            if name not in nlines:
                continue                                  # our own files (hmm_kernel.hip:…, capi.cpp …) are not reference citations
            n += 1
            if not (1 <= a <= b <= nlines[name]):
                bad.append((rel, m.group(0), nlines[name]))
    assert n > 150, n
    assert not bad, bad

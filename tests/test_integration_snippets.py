"""The code INTEGRATION.md tells a maintainer to write is compiled here, so the document cannot drift from the headers.
Blocks are marked `<!-- compile-test: NAME -->` in front of their ```cpp fence.
  batched-window-loop  compiled and linked against dindel_tgi_amd/host (the mirror types carry the reference's class names)
  haplotype-dump       the reference-side block that writes dindel_gpu's --hapFile: compiled against stand-ins holding the members it touches
                       (Haplotype::seq / indels / snps / ml.hpos, AlignedVariant's getters), run, and its output read back through HaplotypeFixture"""
import ctypes as C
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dindel_tgi_amd", "host")


def _block(name):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- compile-test: %s -->\s*```cpp\n(.*?)\n```" % re.escape(name), text, re.S)
    assert m, "INTEGRATION.md has no compile-test block called " + name
    return m.group(1)


def test_batched_window_loop_snippet_compiles_and_links(tmp_path):
    src = tmp_path / "loop.cpp"
    src.write_text(_block("batched-window-loop") + "\nint main() { return 0; }\n")
    exe = str(tmp_path / "loop")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", HOST, str(src), "-o", exe, "-L", HOST, "-ldindel_host",
                           "-L", os.path.join(ROOT, "dindel_tgi_amd", "csrc"), "-ldindel_hmm", "-Wl,-rpath," + HOST, "-Wl,-rpath," + os.path.join(ROOT, "dindel_tgi_amd", "csrc"),
                           "-Wl,--no-as-needed"])
    body = _block("batched-window-loop")
    for needed in ("computeLikelihoodsBatch", "computeLikelihoodsFasterBatch", "diploidGLF(", "skippedWindowLine", "jobs[i].error"):
        assert needed in body


def test_field_table_names_real_accessors():
    """Every accessor the §3c table sends a maintainer to exists in compute_likelihoods.hpp."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = text[text.index("What `diploidGLF` / `filterHaplotypes` read of"):text.index("## 4. Feeding the GPU")]
    header = open(os.path.join(HOST, "compute_likelihoods.hpp")).read()
    names = set(re.findall(r"liks\.(\w+)\(", table))
    assert {"ll", "offHap", "offHapHMQ", "numIndels", "indelCount", "nBQT", "nMMRight", "hapIndelCovered", "hapSNPCovered", "hapIndelFilterCovered", "varSlot", "coveredAt",
            "onHap", "get", "rows"} <= names
    for n in names:
        assert re.search(r"\b%s\(" % n, header), n


HARNESS = r'''
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#include "dindel_types.hpp"
using dindel::AlignedVariant;
struct RefMLAlignment { std::vector<int> hpos; };
struct RefHaplotype { std::string seq; std::map<int, AlignedVariant> indels, snps; RefMLAlignment ml; };      // the members the block touches (Haplotype.hpp:40-312)
static void window(int index, unsigned leftPos, unsigned rightPos, const std::vector<RefHaplotype> &haps, bool skip)
{
%s
}
int main()
{
    std::vector<RefHaplotype> haps(2);
    haps[0].seq = "ACGTACGTAC"; for (int b = 0; b < 10; b++) haps[0].ml.hpos.push_back(b);
    haps[0].indels[4] = AlignedVariant("*REF", 4, 4, 4, 4);
    haps[1].seq = "ACGTTTACGTAC"; for (int b = 0; b < 12; b++) haps[1].ml.hpos.push_back(b < 4 ? b : (b < 6 ? -1 : b - 2));
    AlignedVariant ins("+TT", 4, 4, 4, 5); ins.setFlanking(3, 4, 3, 6);
    haps[1].indels[4] = ins;
    haps[1].snps[7] = AlignedVariant("A=>C", 7, 7, 9, 9);
    window(1, 1000, 1009, haps, false);
    window(2, 2000, 2009, haps, true);                       // a window getHaplotypes asked to skip: no record
    haps.resize(1);
    window(3, 3000, 3009, haps, false);
    return 0;
}
'''


def test_haplotype_dump_snippet_writes_what_the_fixture_reader_reads(tmp_path):
    src = tmp_path / "dump.cpp"
    src.write_text(HARNESS % _block("haplotype-dump"))
    exe = str(tmp_path / "dump")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I", HOST, str(src), "-o", exe])
    out = str(tmp_path / "haps.txt")
    env = dict(os.environ, DINDEL_DUMP_HAPS=out)
    subprocess.check_call([exe], env=env)
    text = open(out).read()
    assert text.startswith("W 1 1000 1009\nH ACGTACGTAC\nA 0 1 2 3 4 5 6 7 8 9\nV I 4 *REF 4 4 4 4 4 4 4 4\nH ACGTTTACGTAC\nA 0 1 2 3 -1 -1 4 5 6 7 8 9\nV I 4 +TT 4 4 4 5 3 4 3 6\nV S 7 A=>C 7 7 9 9 7 7 9 9\nW 3 ")
    from dindel_tgi_amd import hostlib
    lib = hostlib.load()
    lib.ddh_fixture_json.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(1 << 20)
    assert lib.ddh_fixture_json(out.encode(), (C.c_int * 3)(1, 2, 3), 3, buf, len(buf)) > 0
    got = json.loads(buf.value.decode())
    assert got[1] is None and got[0][:3] == [1, 1000, 1009] and got[2][:3] == [3, 3000, 3009] and len(got[2][3]) == 1
    h0, h1 = got[0][3]
    assert h0 == ["ACGTACGTAC", [["I", 4, "*REF", 4, 4, 4, 4, 4, 4, 4, 4]], list(range(10))]
    assert h1 == ["ACGTTTACGTAC", [["I", 4, "+TT", 4, 4, 4, 5, 3, 4, 3, 6], ["S", 7, "A=>C", 7, 7, 9, 9, 7, 7, 9, 9]], [0, 1, 2, 3, -1, -1, 4, 5, 6, 7, 8, 9]]

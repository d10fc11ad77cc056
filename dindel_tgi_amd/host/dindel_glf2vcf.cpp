// dindel_glf2vcf — command-line front end of glf_to_vcf.cpp: the reference's `python mergeOutputDiploid.py`
// (python/mergeOutputDiploid.py:319-348) with the same options:
//   -i/--inputFiles FILE   file listing the Dindel '.glf.txt' files to merge
//   -o/--outputFile FILE   output VCF
//   -s/--sampleID ID       sample column name [SAMPLE]
//   -r/--refFile FILE      reference FASTA (with FILE.fai)
//   --maxHPLen N           homopolymer length named in the hp filter header [10]
//   -f/--filterQual N      calls below this quality get the qN filter [20]
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include "glf_to_vcf.hpp"

int main(int argc, char **argv)
{
    std::string inputFiles, outputFile, sampleID = "SAMPLE", refFile;
    int maxHPLen = 10, filterQual = 20;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        const char *v = i + 1 < argc ? argv[i + 1] : NULL;
        if ((a == "-i" || a == "--inputFiles") && v) { inputFiles = v; i++; }
        else if ((a == "-o" || a == "--outputFile") && v) { outputFile = v; i++; }
        else if ((a == "-s" || a == "--sampleID") && v) { sampleID = v; i++; }
        else if ((a == "-r" || a == "--refFile") && v) { refFile = v; i++; }
        else if (a == "--maxHPLen" && v) { maxHPLen = atoi(v); i++; }
        else if ((a == "-f" || a == "--filterQual") && v) { filterQual = atoi(v); i++; }
        else { std::cerr << "Unknown or incomplete option: " << a << "\n"; return 2; }
    }
    if (inputFiles.empty()) { std::cerr << "Please specify --inputFiles\n"; return 1; }
    if (outputFile.empty()) { std::cerr << "Please specify --outputFile\n"; return 1; }
    if (refFile.empty()) { std::cerr << "Please specify --refFile\n"; return 1; }
    try {
        dindel::mergeOutput(inputFiles, sampleID, refFile, maxHPLen, outputFile, filterQual);
    } catch (std::string &e) {
        std::cerr << "An error occurred!\n" << e << "\n";
        return 1;
    }
    return 0;
}

// TEST INFRASTRUCTURE ONLY: scratch driver linking the reference's own
// MultipleSequenceAlignmentSP (built by oracle/Makefile into oracle/_ref/) to
// produce golden vectors for the sum-of-pairs MSA.
// stdin: one sequence per line (already in the order the caller would pass).
// stdout: one padded row per input sequence, the same way
// /root/reference/StrainCall/PartialOrderGraph.cpp:510-518 reads them back.
#include "MultipleSequenceAlignment.hpp"
#include <iostream>
#include <string>
#include <vector>
int main() {
    std::vector<std::string> seqs;
    for (std::string line; std::getline(std::cin, line);)
        if (!line.empty()) seqs.push_back(line);
    if (seqs.empty()) return 0;
    MultipleSequenceAlignmentSP<Index2D, SimpleScoreModel, std::vector, std::string, char> msa;
    MSA<std::vector, char> result;
    msa.align(seqs, result);
    for (size_t t = 0; t < seqs.size(); ++t) {
        std::vector<char> res;
        result.get((int)t, res);
        std::cout << std::string(res.begin(), res.end()) << "\n";
    }
    std::cout << "#columns " << result.size() << "\n";
    return 0;
}

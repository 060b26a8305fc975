// TEST ONLY (CPU): builds the product's host graph (rambl_amd/csrc/sc_graph.cpp)
// from an oracle read dump and prints the -G text, so the host side of rows
// a5/a6/a9/a10/a11 can be compared with the oracle without a GPU.  The MSA
// callback is the ORACLE's (liboracle.so) -- on the GPU path it is k_msa.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include "../../rambl_amd/csrc/sc_graph.hpp"
extern "C" int oracle_msa_align(const char** seqs, int n, char* out, int out_cap, int* ncol);
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream in(argv[1]);
    std::string ref;
    std::vector<sc::AlignedRead> reads;
    for (std::string line; std::getline(in, line);) {
        std::vector<std::string> f;
        std::stringstream ss(line);
        for (std::string t; std::getline(ss, t, '\t');) f.push_back(t);
        if (f.empty()) continue;
        if (f[0] == "REF") ref = f.size() > 1 ? f[1] : "";
        else if (f[0] == "READ") reads.push_back({std::stoi(f[1]), f[2], f[3], std::stoi(f[4])});
    }
    sc::MsaFn msa = [](const std::vector<std::string>& seqs, std::vector<std::string>& rows) {
        std::vector<const char*> p;
        size_t tot = 0;
        for (auto& s : seqs) { p.push_back(s.c_str()); tot += s.size(); }
        std::vector<char> out((seqs.size()) * (tot + 2));
        int ncol = 0;
        oracle_msa_align(p.data(), (int)seqs.size(), out.data(), (int)out.size(), &ncol);
        rows.clear();
        for (size_t t = 0; t < seqs.size(); t++) rows.emplace_back(out.data() + t * (ncol + 1), (size_t)ncol);
        return ncol;
    };
    sc::PoGraph g(ref, reads, msa);
    std::cout << g.dump();
    sc::FlatGraph f;
    sc::flatten(g, (int)reads.size(), f);
    std::cerr << "nodes " << f.n_nodes << " levels " << f.n_levels << " K " << f.K << " msa_calls " << g.msa_calls
              << " unsupported '" << f.unsupported << "' sorted " << f.pools_sorted << "\n";
    return 0;
}

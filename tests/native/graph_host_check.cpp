// TEST ONLY (CPU): builds the product's host graph (rambl_amd/csrc/sc_graph.cpp)
// from an oracle read dump and prints the -G text, so the host side of rows
// a5/a6/a9/a10/a11 can be compared with the oracle without a GPU.  The MSA
// callback is the ORACLE's (liboracle.so) -- on the GPU path it is k_msa -- and the
// class tables of row a5 are produced here by plain loops (on the GPU path: k_thread_*).
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
    sc::ThreadFn thr = [](const std::string& G, const std::vector<sc::AlignedRead>& R,
                          const std::vector<std::vector<sc::CigarOp>>& cig, sc::ThreadTables& T) {
        const int glen = (int)G.size(), n = (int)R.size(), ncls = glen * 8, INF = 0x7fffffff;
        bool present[256] = {false};                          // read symbols only, as the device stage codes them
        for (auto& r : R) for (unsigned char c : r.seq) present[c] = true;
        for (int c = 0; c < 256; c++) T.lut[c] = 0xFF;
        for (char c : {'A', 'C', 'G', 'T'}) { T.lut[(unsigned char)c] = (uint8_t)T.sym.size(); T.sym.push_back(c); }
        for (int c = 0; c < 256; c++) if (present[c] && T.lut[c] == 0xFF) { T.lut[c] = (uint8_t)T.sym.size(); T.sym.push_back((char)c); }
        T.count.assign(ncls, 0); T.minrid.assign(ncls, INF); T.smin.assign(ncls, INF); T.emin.assign(ncls, INF);
        T.tmin.assign((size_t)ncls * 8, INF);
        std::vector<std::vector<int>> pools(ncls);
        for (int r = 0; r < n; r++) {
            int i = R[r].pos, j = 0; bool prev_m = false;
            for (size_t k = 0; k < cig[r].size(); k++) {
                const auto& c = cig[r][k];
                if (c.op == 'M') {
                    for (int t = 0; t < c.len; t++) {
                        const int code = T.lut[(unsigned char)R[r].seq[j + t]], cls = (i + t) * 8 + code;
                        T.count[cls]++; if (r < T.minrid[cls]) T.minrid[cls] = r; pools[cls].push_back(r);
                        if (t > 0 || prev_m) { int& x = T.tmin[(size_t)(i + t) * 64 + T.lut[(unsigned char)R[r].seq[j + t - 1]] * 8 + code]; if (r < x) x = r; }
                        else if (k == 0 && r < T.smin[cls]) T.smin[cls] = r;
                        if (t == c.len - 1 && k + 1 == cig[r].size() && r < T.emin[cls]) T.emin[cls] = r;
                    }
                    i += c.len; j += c.len; prev_m = true;
                } else if (c.op == 'I') { j += c.len; prev_m = false; }
                else if (c.op == 'D') { i += c.len; prev_m = false; }
            }
        }
        T.off.assign((size_t)ncls + 1, 0);
        for (int c = 0; c < ncls; c++) { T.off[c + 1] = T.off[c] + (int)pools[c].size(); for (int r : pools[c]) T.pool.push_back(r); }
    };
    sc::PoGraph g(ref, reads, msa, thr);
    std::cout << g.dump();
    sc::FlatGraph f;
    sc::flatten(g, (int)reads.size(), f);
    std::cerr << "nodes " << f.n_nodes << " levels " << f.n_levels << " K " << f.K << " msa_calls " << g.msa_calls
              << " unsupported '" << f.unsupported << "' sorted " << f.pools_sorted << "\n";
    return 0;
}

// Native read ingest of the StrainCall path, host side of libstraincall_hip.so.
//
// Replaces the text that crosses the reference's samtools boundary and what it does with it:
//   samtools view <aln> -q mq -F 1804 gene:p0-p1     /root/reference/StrainCall/StrainCall.cpp:496
//   samtools mpileup -q mq -Q0 -A -r gene:P-Q <aln>  :696 (only "is there a '+' / a '-' or '*' in column 5" is read, :712-735)
//   load_mapping_reads                               :480-670  (depth -> rho, filters, crop to the window,
//                                                     mt19937(1234) thinning, exact-duplicate collapse, mate table)
// The alignment file (SAM text, or BAM = BGZF-compressed records, SAM specification section 4) is read once and
// indexed by reference name; a region's reads come back as the packed arrays sc_roi_submit takes, so the caller
// hands them on without touching a read.  Nothing here uses the GPU.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/straincall_hip.h"
#include "sc_ingest.hpp"

namespace {
using sc_ingest::Rec;

// std::stoi on a field: leading blanks, sign, digits; anything after is ignored; no digits = error
bool lead_int(const char* s, int n, int& out) {
    int i = 0;
    while (i < n && (s[i] == ' ' || s[i] == '\t')) i++;
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    if (i >= n || s[i] < '0' || s[i] > '9') return false;
    long v = 0;
    while (i < n && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (s[i] - '0'); if (v > 0x7fffffffL) return false; i++; }
    out = (int)(neg ? -v : v);
    return true;
}

struct Op { char op; int len; };
// PartialOrderGraph.cpp:13-59: a number, then one of MIDNSHP (kept) or = X (become M); other characters join the number
bool parse_cigar(const char* c, int n, std::vector<Op>& out) {
    out.clear();
    int start = 0;
    for (int i = 0; i < n; i++) {
        const char ch = c[i];
        const bool plain = ch == 'M' || ch == 'I' || ch == 'D' || ch == 'N' || ch == 'S' || ch == 'H' || ch == 'P';
        if (plain || ch == '=' || ch == 'X') {
            int len;
            if (!lead_int(c + start, i - start, len)) return false;
            out.push_back({plain ? ch : 'M', len});
            start = i + 1;
        }
    }
    return true;
}
int ref_span_end(int pos, const char* c, int n) {      // samtools' notion: M D N = X consume the reference
    long tot = 0, v = 0;
    for (int i = 0; i < n; i++) {
        const char ch = c[i];
        if (ch >= '0' && ch <= '9') { v = v * 10 + (ch - '0'); continue; }
        if (ch == 'M' || ch == 'D' || ch == 'N' || ch == '=' || ch == 'X') tot += v;
        v = 0;
    }
    return pos + (int)(tot > 1 ? tot : 1) - 1;
}

struct Mt19937 {                                       // std::mt19937 + generate_canonical<double,53> (libstdc++)
    uint32_t x[624]; int p = 624;
    explicit Mt19937(uint32_t seed) {
        x[0] = seed;
        for (int i = 1; i < 624; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    }
    uint32_t next() {
        if (p >= 624) {
            const uint32_t UP = 0x80000000u, LO = 0x7fffffffu;
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (x[k] & UP) | (x[(k + 1) % 624] & LO);
                x[k] = x[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
            }
            p = 0;
        }
        uint32_t z = x[p++];
        z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);
        return z;
    }
    double canonical() {
        const double lo = (double)next(), hi = (double)next();
        double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
        if (r >= 1.0) r = 0x1.fffffffffffffp-1;
        return r;
    }
};

}  // namespace

namespace {

int inflate_threads();
// One stretch of the text (whole lines): the records of the references this handle keeps, per reference in file order,
// and the statistics of every reference.
struct SamPart {
    std::vector<std::pair<std::string, std::vector<Rec>>> buckets;      // in order of first appearance
    std::unordered_map<std::string, size_t> index;
    std::unordered_map<std::string, sc_aln::RefStat> stats;
    long n_records = 0;
    std::string error;
};
bool parse_sam_part(const sc_aln& a, const char* p, const char* end, SamPart& out) {
    std::string last_name;
    std::vector<Rec>* bucket = nullptr;
    sc_aln::RefStat* stat = nullptr;
    bool wanted = false;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* next = nl ? nl + 1 : end;
        if (le > p && le[-1] == '\r') le--;
        if (le > p && *p != '@') {
            const char* f[12];
            int nf = 0;
            f[0] = p;
            for (const char* q = p; q < le && nf < 11; q++)
                if (*q == '\t') f[++nf] = q + 1;
            // f[k] = start of field k; field k ends one before f[k+1] (or at the end of the line for the last one found)
            if (nf >= 10) {
                auto fend = [&](int k) { return k < nf ? f[k + 1] - 1 : le; };
                const char* qual_end = le;
                if (nf == 11) qual_end = f[11] - 1;
                Rec r{};
                r.qname = f[0]; r.qlen = (int)(fend(0) - f[0]);
                bool ok = lead_int(f[1], (int)(fend(1) - f[1]), r.flag) && lead_int(f[3], (int)(fend(3) - f[3]), r.pos) &&
                          lead_int(f[4], (int)(fend(4) - f[4]), r.mapq);
                if (!ok) { out.error = "SAM record with a non-numeric FLAG, POS or MAPQ"; return false; }
                r.cigar = f[5]; r.clen = (int)(fend(5) - f[5]);
                r.seq = f[9]; r.slen = (int)(fend(9) - f[9]);
                r.quallen = (int)(qual_end - f[10]);
                r.ref_end = ref_span_end(r.pos, r.cigar, r.clen);
                const size_t rl = (size_t)(fend(2) - f[2]);
                if (!stat || last_name.size() != rl || memcmp(last_name.data(), f[2], rl) != 0) {
                    last_name.assign(f[2], rl);
                    stat = &out.stats[last_name];
                    wanted = a.wants(f[2], rl);
                    bucket = nullptr;
                    if (wanted) {
                        auto it = out.index.find(last_name);
                        if (it == out.index.end()) { it = out.index.emplace(last_name, out.buckets.size()).first; out.buckets.emplace_back(last_name, std::vector<Rec>()); }
                        bucket = &out.buckets[it->second].second;
                    }
                }
                stat->n++; stat->bases += (long)(r.ref_end - r.pos + 1);
                if (wanted) bucket->push_back(r);
                out.n_records++;
            }
        }
        p = next;
    }
    return true;
}
// The text is cut at line ends into as many stretches as this rank has host threads; the stretches are parsed side by
// side and joined in file order (a reference's records keep the order of the file).
bool load_sam_text(sc_aln& a, const char* text, size_t len) {
    int nt = inflate_threads();
    size_t min_chunk = (size_t)1 << 20;                                   // a stretch below this is not worth a thread
    if (const char* e = getenv("SC_INGEST_MIN_CHUNK")) min_chunk = (size_t)std::max(atol(e), 64L);
    nt = (int)std::max<size_t>(std::min<size_t>((size_t)nt, len / min_chunk), 1);
    std::vector<const char*> cut((size_t)nt + 1, text + len);
    cut[0] = text;
    for (int k = 1; k < nt; k++) {
        const char* q = text + len / (size_t)nt * (size_t)k;
        if (q < cut[(size_t)k - 1]) q = cut[(size_t)k - 1];
        const char* nl = (const char*)memchr(q, '\n', (size_t)(text + len - q));
        cut[(size_t)k] = nl ? nl + 1 : text + len;
    }
    std::vector<SamPart> parts((size_t)nt);
    std::vector<char> ok((size_t)nt, 1);
    std::vector<std::thread> pool;
    for (int k = 1; k < nt; k++) pool.emplace_back([&, k] { ok[(size_t)k] = parse_sam_part(a, cut[(size_t)k], cut[(size_t)k + 1], parts[(size_t)k]) ? 1 : 0; });
    ok[0] = parse_sam_part(a, cut[0], cut[1], parts[0]) ? 1 : 0;
    for (auto& th : pool) th.join();
    for (int k = 0; k < nt; k++) {
        if (!ok[(size_t)k]) { a.error = parts[(size_t)k].error; return false; }
        for (auto& b : parts[(size_t)k].buckets) {
            std::vector<Rec>& dst = a.by_ref[b.first];
            if (dst.empty()) dst.swap(b.second); else dst.insert(dst.end(), b.second.begin(), b.second.end());
        }
        for (auto& st : parts[(size_t)k].stats) { sc_aln::RefStat& d = a.stats[st.first]; d.n += st.second.n; d.bases += st.second.bases; }
        a.n_records += parts[(size_t)k].n_records;
    }
    return true;
}

// BGZF: a series of gzip members, each with the BC extra field; the payload is raw deflate (SAM spec 4.1).
// The members are independent: one pass over the headers finds where each one's bytes go, then the host threads
// the process may use inflate them side by side.
int inflate_threads() {
    long quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {             // "<quota> <period>" or "max <period>"
        long q = 0, per = 0;
        if (fscanf(f, "%ld %ld", &q, &per) == 2 && q > 0 && per > 0) quota = (q + per - 1) / per;
        fclose(f);
    }
    long n = (long)std::thread::hardware_concurrency();
    if (quota > 0 && quota < n) n = quota;
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) { const long k = atol(e); if (k > 1) n = std::max<long>(n / k, 1); }     // ranks sharing the host
    if (const char* e = getenv("SC_INGEST_THREADS")) n = atol(e);
    return (int)std::min<long>(std::max<long>(n, 1), 32);
}
bool inflate_bgzf(const unsigned char* src, size_t n, std::vector<unsigned char>& out, std::string& err) {
    struct Member { size_t cdata, clen, at; unsigned isize, crc; };
    std::vector<Member> members;
    size_t o = 0, total = 0;
    while (o + 18 <= n) {
        if (src[o] != 0x1f || src[o + 1] != 0x8b || src[o + 2] != 8 || !(src[o + 3] & 4)) { err = "not a BGZF block"; return false; }
        const unsigned xlen = src[o + 10] | (src[o + 11] << 8);
        size_t x = o + 12, xe = x + xlen;
        long bsize = -1;
        if (xe + 8 > n) { err = "truncated BGZF block"; return false; }
        while (x + 4 <= xe) {
            const unsigned slen = src[x + 2] | (src[x + 3] << 8);
            if (x + 4 + slen > xe) break;                                 // a subfield that runs past the extra field: not ours to read
            if (src[x] == 'B' && src[x + 1] == 'C' && slen == 2) bsize = (src[x + 4] | (src[x + 5] << 8)) + 1L;
            x += 4 + slen;
        }
        if (bsize < (long)xlen + 20 || o + (size_t)bsize > n) { err = "BGZF block without a BC field or truncated"; return false; }
        const unsigned isize = src[o + bsize - 4] | (src[o + bsize - 3] << 8) | (src[o + bsize - 2] << 16) | ((unsigned)src[o + bsize - 1] << 24);
        const unsigned crc = src[o + bsize - 8] | (src[o + bsize - 7] << 8) | (src[o + bsize - 6] << 16) | ((unsigned)src[o + bsize - 5] << 24);
        if (isize > 65536u) { err = "BGZF block claims more than 64 KiB of data"; return false; }      // SAM spec 4.1
        members.push_back(Member{o + 12 + xlen, (size_t)bsize - xlen - 20, total, isize, crc});
        total += isize;
        if (total > ((size_t)1 << 40)) { err = "BGZF file larger than this reader accepts"; return false; }
        o += (size_t)bsize;
    }
    out.resize(total);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&]() {
        for (size_t k = next.fetch_add(1); k < members.size() && !bad.load(std::memory_order_relaxed); k = next.fetch_add(1)) {
            const Member& m = members[k];
            if (!m.isize) { if (m.crc != 0u) { bad = true; return; } continue; }
            z_stream zs{};
            if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
            zs.next_in = const_cast<unsigned char*>(src + m.cdata); zs.avail_in = (unsigned)m.clen;
            zs.next_out = out.data() + m.at; zs.avail_out = m.isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = true; return; }
            if ((unsigned)crc32(crc32(0L, Z_NULL, 0), out.data() + m.at, m.isize) != m.crc) { bad = true; return; }   // the member's CRC-32 of its data
        }
    };
    const int nt = (int)std::min<size_t>((size_t)inflate_threads(), std::max<size_t>(members.size() / 4, 1));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (bad) { err = "corrupt BGZF block (inflate or CRC-32)"; return false; }
    return true;
}

bool load_bam(sc_aln& a, const unsigned char* src, size_t n) {
    std::vector<unsigned char> d;
    if (!inflate_bgzf(src, n, d, a.error)) return false;
    auto i32 = [&](size_t o) { int32_t v; memcpy(&v, d.data() + o, 4); return v; };
    if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) { a.error = "not a BAM file"; return false; }
    size_t o = 8 + (size_t)i32(4);
    if (o + 4 > d.size()) { a.error = "truncated BAM header"; return false; }
    const int n_ref = i32(o);
    o += 4;
    std::vector<std::string> refs;
    for (int k = 0; k < n_ref; k++) {
        if (o + 4 > d.size()) { a.error = "truncated BAM header"; return false; }
        const int l_name = i32(o);
        o += 4;
        if (l_name < 1 || o + (size_t)l_name + 4 > d.size()) { a.error = "truncated BAM header"; return false; }
        refs.emplace_back((const char*)d.data() + o, (size_t)l_name - 1);
        o += (size_t)l_name + 4;
    }
    static const char SEQ[] = "=ACMGRSVTWYHKDBN", CIG[] = "MIDNSHP=X???????";
    const std::string star("*");
    while (o + 4 <= d.size()) {
        const int block = i32(o);
        o += 4;
        if (block < 32 || o + (size_t)block > d.size()) { a.error = "truncated BAM record"; return false; }
        const int ref_id = i32(o), pos = i32(o + 4);
        const unsigned l_read_name = d[o + 8], mapq = d[o + 9];
        const unsigned n_cigar = d[o + 12] | (d[o + 13] << 8), flag = d[o + 14] | (d[o + 15] << 8);
        const int l_seq = i32(o + 16);
        size_t p = o + 32;
        if (l_seq < 0 || p + l_read_name + 4ul * n_cigar + (size_t)(l_seq + 1) / 2 + (size_t)l_seq > o + (size_t)block) { a.error = "corrupt BAM record"; return false; }
        const std::string& rname = (ref_id >= 0 && ref_id < n_ref) ? refs[(size_t)ref_id] : star;
        if (!a.wants(rname.data(), rname.size())) {
            // not this rank's reference: its statistics only (reference span of the CIGAR, as ref_span_end counts it)
            long tot = 0;
            for (unsigned k = 0; k < n_cigar; k++) {
                uint32_t c; memcpy(&c, d.data() + p + l_read_name + 4ul * k, 4);
                const unsigned op = c & 15;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) tot += (long)(c >> 4);
            }
            sc_aln::RefStat& st = a.stats[rname];
            st.n++; st.bases += tot > 1 ? tot : 1;
            a.n_records++;
            o += (size_t)block;
            continue;
        }
        Rec r{};
        r.qlen = l_read_name ? (int)l_read_name - 1 : 0;
        char* qn = a.alloc((size_t)r.qlen + 1);
        memcpy(qn, d.data() + p, (size_t)r.qlen);
        r.qname = qn;
        p += l_read_name;
        std::string cig;
        for (unsigned k = 0; k < n_cigar; k++) {
            uint32_t c; memcpy(&c, d.data() + p + 4ul * k, 4);
            cig += std::to_string(c >> 4);
            cig += CIG[c & 15];
        }
        if (cig.empty()) cig = "*";
        p += 4ul * n_cigar;
        char* cg = a.alloc(cig.size() + 1);
        memcpy(cg, cig.data(), cig.size());
        r.cigar = cg; r.clen = (int)cig.size();
        char* sq = a.alloc((size_t)std::max(l_seq, 1) + 1);
        for (int k = 0; k < l_seq; k++) { const unsigned b = d[p + (size_t)k / 2]; sq[k] = SEQ[(k & 1) ? (b & 15) : (b >> 4)]; }
        if (l_seq == 0) sq[0] = '*';
        r.seq = sq; r.slen = std::max(l_seq, 1);
        p += (size_t)(l_seq + 1) / 2;
        r.quallen = (l_seq == 0 || d[p] == 0xff) ? 1 : l_seq;            // "*" when absent
        r.flag = (int)flag; r.pos = pos + 1; r.mapq = (int)mapq;
        r.ref_end = ref_span_end(r.pos, r.cigar, r.clen);
        a.by_ref[rname].push_back(r);
        { sc_aln::RefStat& st = a.stats[rname]; st.n++; st.bases += (long)(r.ref_end - r.pos + 1); }
        a.n_records++;
        o += (size_t)block;
    }
    return true;
}

// The front of a read that starts before the window, the back of one that ends after it, and soft clips are cut
// away; what is left is the read inside the window (crop_read_within_window, StrainCall.cpp:291-414).  A reference
// coordinate map of the operations decides: M and D operations are clipped to the window position by position, an
// insertion lying before the first / after the last kept reference position goes with its bases, operations that
// consume neither (H, P) or that the reference does not know to consume the reference (N) stay as they are.
struct Cropped { int lead = 0, trail = 0; std::vector<Op> ops; bool ok = true; };
void crop_to_window(int w0, int w1, const std::vector<Op>& ops, int r0, int r1, Cropped& out) {
    out.lead = out.trail = 0; out.ops.clear(); out.ok = true;
    const int n = (int)ops.size();
    if (n == 0) { out.ok = false; return; }
    int first = 0;
    if (ops[0].op == 'S') { out.lead = ops[0].len; first = 1; }
    if (first >= n) { out.ok = false; return; }
    // ---- front: walk the reference cursor up to the window start
    int k = first;
    if (r0 < w0 && r0 < w1) {
        int cur = r0, used_last = 0;
        Op last{0, 0};
        while (cur < w0 && cur < w1) {
            if (k >= n) { out.ok = false; return; }            // the read never reaches the window (the reference reads past its vector here)
            const Op o = ops[k++];
            int used = 0;
            if (o.op == 'M' || o.op == 'D') {
                used = std::max(0, std::min(o.len, w0 - cur));
                cur += used;
                if (o.op == 'M') out.lead += used;
            } else if (o.op == 'I') {
                out.lead += o.len;                               // an insertion in front of the window leaves with its bases
            }
            last = o; used_last = used;
        }
        if (used_last < last.len) out.ops.push_back(Op{last.op, last.len - used_last});      // the operation the window starts in
    } else {
        if (ops[k].len > 0) out.ops.push_back(ops[k]);
        k++;
    }
    for (; k < n; k++) out.ops.push_back(ops[k]);
    // ---- back: the same from the other end, on the operations still held
    int last = n - 1;
    if (ops[last].op == 'S') {
        out.trail = ops[last].len;
        last--;
        if (out.ops.empty()) { out.ok = false; return; }
        out.ops.pop_back();
    }
    int cur = r1;
    while (cur > w1 && cur > w0) {
        if (last < 0 || out.ops.empty()) { out.ok = false; return; }
        const Op o = ops[last--];
        int used = 0;
        if (o.op == 'M' || o.op == 'D') {
            used = std::min(o.len, cur - w1);
            if (used < 0) used = 0;
            cur -= used;
            if (o.op == 'M') out.trail += used;
        } else if (o.op == 'I') {
            out.trail += o.len;
        }
        if (used == o.len || o.op == 'I') out.ops.pop_back();
        else out.ops.back().len -= used;
    }
}

struct Kept {                      // a read that survived the filters, before duplicates collapse
    int relpos; uint32_t cig_off, cig_len; const char* seq; int seq_len; const char* name; int name_len; char suffix;   // suffix: 0, '1' or '2'
};

}  // namespace

struct sc_reads {
    std::vector<int> pos, cigar_off, seq_off, copies, mate_idx, mate_off;
    std::string cigar_text, seq_text;
    long n_input = 0;
    int depth = 0;
};

namespace {
// std::stable_sort on the rank's ingest threads: stretches sorted side by side, then merged pairwise (std::inplace_merge
// keeps equal elements in order, so the result is the one a single stable sort gives).  A million reads of a deep region
// are compared by position, CIGAR text and bases: two such sorts were half of the region's ingest.
template <class T, class Less>
static void stable_sort_mt(std::vector<T>& v, const Less& less) {
    const size_t n = v.size();
    int nt = std::min(inflate_threads(), 16);
    static const size_t min_stretch = [] {          // SC_INGEST_SORT_MIN: elements worth a thread (tests lower it to reach the merges)
        const char* e = getenv("SC_INGEST_SORT_MIN");
        const long v = e ? atol(e) : 65536;
        return (size_t)(v < 2 ? 2 : v);
    }();
    while (nt > 1 && n / (size_t)nt < min_stretch) nt--;
    if (nt <= 1) { std::stable_sort(v.begin(), v.end(), less); return; }
    std::vector<size_t> cut((size_t)nt + 1);
    for (int t = 0; t <= nt; t++) cut[(size_t)t] = n * (size_t)t / (size_t)nt;
    {
        std::vector<std::thread> ts;
        for (int t = 1; t < nt; t++) ts.emplace_back([&, t] { std::stable_sort(v.begin() + (long)cut[(size_t)t], v.begin() + (long)cut[(size_t)t + 1], less); });
        std::stable_sort(v.begin(), v.begin() + (long)cut[1], less);
        for (auto& th : ts) th.join();
    }
    for (int width = 1; width < nt; width *= 2) {
        std::vector<std::thread> ts;
        for (int lo = 0; lo + width < nt; lo += 2 * width) {
            const size_t b = cut[(size_t)lo], m = cut[(size_t)(lo + width)], e = cut[(size_t)std::min(lo + 2 * width, nt)];
            ts.emplace_back([&v, &less, b, m, e] { std::inplace_merge(v.begin() + (long)b, v.begin() + (long)m, v.begin() + (long)e, less); });
        }
        for (auto& th : ts) th.join();
    }
}

}  // namespace

extern "C" {

int sc_aln_open(const char* path, sc_aln** out) { return sc_aln_open_filtered(path, nullptr, -1, out); }

int sc_aln_open_filtered(const char* path, const char* const* names, int n_names, sc_aln** out) {
    if (!path || !out || (n_names > 0 && !names)) return SC_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<sc_aln> a(new sc_aln());
    if (n_names >= 0) {
        a->keep_all = false;
        for (int i = 0; i < n_names; i++) if (names[i]) a->keep.insert(names[i]);
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SC_ERR_ARG;
    struct stat stt;
    if (fstat(fd, &stt) != 0) { close(fd); return SC_ERR_ARG; }
    a->map_len = (size_t)stt.st_size;
    bool ok = true;
    if (a->map_len > 0) {
        a->map = mmap(nullptr, a->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (a->map == MAP_FAILED) { a->map = nullptr; close(fd); return SC_ERR_INTERNAL; }
        const unsigned char* b = (const unsigned char*)a->map;
        try {
            if (a->map_len >= 2 && b[0] == 0x1f && b[1] == 0x8b) {
                ok = load_bam(*a, b, a->map_len);
                munmap(a->map, a->map_len);            // every string was copied out
                a->map = nullptr;
            } else {
                ok = load_sam_text(*a, (const char*)a->map, a->map_len);
            }
        } catch (const std::bad_alloc&) { a->error = "out of memory while reading the alignments"; ok = false; }
        catch (const std::exception& ex) { a->error = ex.what(); ok = false; }
    }
    close(fd);
    *out = a.release();
    return ok ? SC_OK : SC_ERR_ARG;                // on a parse error the handle carries the message (sc_aln_error)
}

void sc_aln_close(sc_aln* a) { delete a; }

// The message of the last failing call of THIS thread on any handle (windows are ingested side by side on several
// threads: a message kept on the shared handle could be another window's); the handle itself keeps what sc_aln_open said.
static thread_local std::string tl_error;
const char* sc_aln_error(sc_aln* a) {
    if (!tl_error.empty()) return tl_error.c_str();
    return a ? a->error.c_str() : "";
}

long sc_aln_records(sc_aln* a) { return a ? a->n_records : 0; }

int sc_aln_ref_stats(sc_aln* a, const char* gene, long* n_records, long* aligned_bases) {
    if (!a || !gene) return SC_ERR_ARG;
    long n = 0, b = 0;
    auto it = a->stats.find(gene);
    if (it != a->stats.end()) { n = it->second.n; b = it->second.bases; }
    if (n_records) *n_records = n;
    if (aligned_bases) *aligned_bases = b;
    return SC_OK;
}

int sc_aln_pileup_flags(sc_aln* a, const char* gene, int P, int Q, int mq, unsigned char* covered, unsigned char* has_ins,
                        unsigned char* has_del) {
    if (!a || !gene || Q < P || !covered || !has_ins || !has_del) return SC_ERR_ARG;
    const int n = Q - P + 1;
    std::vector<int> diff((size_t)n + 1, 0);
    memset(has_ins, 0, (size_t)n); memset(has_del, 0, (size_t)n);
    auto it = a->by_ref.find(gene);
    if (it != a->by_ref.end()) {
        auto mark = [&](unsigned char* arr, int p) { if (p >= P && p <= Q) arr[p - P] = 1; };
        std::vector<Op> ops;
        for (const Rec& r : it->second) {
            if ((r.flag & 1796) || r.mapq < mq) continue;
            if (r.ref_end < P || r.pos > Q) continue;
            // samtools' own CIGAR reading here (= X distinct from M does not matter: all three align bases)
            ops.clear();
            long v = 0;
            for (int i = 0; i < r.clen; i++) {
                const char ch = r.cigar[i];
                if (ch >= '0' && ch <= '9') { v = v * 10 + (ch - '0'); continue; }
                if (ch != 'H' && ch != 'P') ops.push_back({ch, (int)v});
                v = 0;
            }
            const int mq_char = 33 + std::min(r.mapq, 93);
            int p = r.pos;
            bool first = true;
            for (size_t k = 0; k < ops.size(); k++) {
                const char op = ops[k].op;
                const int ln = ops[k].len;
                if (op == 'M' || op == '=' || op == 'X') {
                    const int lo = std::max(p, P), hi = std::min(p + ln - 1, Q);
                    if (lo <= hi) { diff[(size_t)(lo - P)]++; diff[(size_t)(hi - P + 1)]--; }
                    if (first) {
                        // '^' + the mapping quality character in front of the first base: '+', '-' and '*' read as indel marks
                        if (mq_char == '+') mark(has_ins, p);
                        else if (mq_char == '-' || mq_char == '*') mark(has_del, p);
                        first = false;
                    }
                    if (k + 1 < ops.size()) {
                        if (ops[k + 1].op == 'I') mark(has_ins, p + ln - 1);
                        else if (ops[k + 1].op == 'D') mark(has_del, p + ln - 1);
                    }
                    p += ln;
                } else if (op == 'D' || op == 'N') {
                    // a deleted base prints '*' (read as a deletion mark, StrainCall.cpp:712-735); a reference skip (N)
                    // prints '>' or '<', which window_adjust does not look for: it only counts towards the coverage
                    const int lo = std::max(p, P), hi = std::min(p + ln - 1, Q);
                    if (lo <= hi) {
                        diff[(size_t)(lo - P)]++; diff[(size_t)(hi - P + 1)]--;
                        if (op == 'D') memset(has_del + (lo - P), 1, (size_t)(hi - lo + 1));
                    }
                    p += ln;
                }
            }
        }
    }
    int depth = 0;
    for (int i = 0; i < n; i++) {
        depth += diff[(size_t)i];
        covered[i] = depth > 0;
        if (!covered[i]) { has_ins[i] = 0; has_del[i] = 0; }
    }
    return SC_OK;
}

int sc_aln_load_reads(sc_aln* a, const char* gene, int p0, int p1, int mq, int rl, int max_ins, int max_depth, sc_reads** out) {
    if (!a || !gene || !out) return SC_ERR_ARG;
    *out = nullptr;
    tl_error.clear();
    std::unique_ptr<sc_reads> R(new sc_reads());
    // ---- samtools view -q mq -F 1804 gene:p0-p1
    std::vector<const Rec*> view;
    auto it = a->by_ref.find(gene);
    if (it != a->by_ref.end())
        for (const Rec& r : it->second) {
            if ((r.flag & 1804) || r.mapq < mq) continue;
            if (r.ref_end < p0 || r.pos > p1) continue;
            view.push_back(&r);
        }
    R->n_input = (long)view.size();
    // ---- depth over the window -> keep probability (StrainCall.cpp:503-529)
    std::vector<Op> ops;
    int depth = 0;
    for (const Rec* r : view) {
        if (!parse_cigar(r->cigar, r->clen, ops)) { tl_error = "malformed CIGAR"; return SC_ERR_ARG; }
        int len = 0;
        for (const Op& o : ops) if (o.op == 'M' || o.op == 'D') len += o.len;
        const int r0 = r->pos, r1 = r0 + len - 1;
        if (p0 <= r0 && p1 > r1) depth += r1 - r0 + 1;
        else if (p0 <= r0 && p1 <= r1) depth += p1 - r0 + 1;
        else if (p0 > r0 && p1 <= r1) depth += p1 - p0 + 1;
        else if (p0 > r0 && p1 > r1) depth += r1 - p0 + 1;
    }
    depth /= (p1 - p0 + 1);
    R->depth = depth;
    const double rho = std::min(1.0, (double)max_depth / ((double)depth + 0.0));
    Mt19937 gen(1234u);
    // ---- filters, crop, thinning
    std::vector<Kept> kept;
    std::string cig_pool;
    Cropped cr;
    char num[16];
    for (const Rec* r : view) {
        if ((size_t)r->slen < (size_t)(long)rl) continue;                      // f10.length() < rl (rl widened like the reference's)
        bool amb = false;
        for (int i = 0; i < r->slen; i++) if (r->seq[i] == 'N' || r->seq[i] == 'n') { amb = true; break; }
        if (amb) continue;
        char suffix = 0;
        if ((r->flag & 65) == 65) suffix = '1';
        else if ((r->flag & 129) == 129) suffix = '2';
        parse_cigar(r->cigar, r->clen, ops);
        const int rp0 = r->pos;
        int rp1 = rp0;
        for (const Op& o : ops) if (o.op == 'M' || o.op == 'D') rp1 += o.len;
        rp1 -= 1;
        int relpos = rp0 - p0;
        if (relpos < 0) relpos = 0;
        crop_to_window(p0, p1, ops, rp0, rp1, cr);
        if (!cr.ok || cr.lead > r->slen || cr.trail > r->quallen) {
            tl_error = "a read cannot be cropped to the window (soft clips / CIGAR longer than its bases)";
            return SC_ERR_ARG;
        }
        const int cnt = r->slen - cr.lead - cr.trail;
        const int seq_len = cnt < 0 ? r->slen - cr.lead : cnt;                  // substr(i, npos-like count)
        int maxins = 0;
        for (const Op& o : cr.ops) if (o.op == 'I' && o.len > maxins) maxins = o.len;
        if (!((size_t)seq_len > (size_t)(long)rl && maxins < max_ins)) continue;
        if (gen.canonical() > rho) continue;                                    // drawn only for reads that passed
        Kept kp;
        kp.relpos = relpos;
        kp.cig_off = (uint32_t)cig_pool.size();
        for (const Op& o : cr.ops) { const int m = snprintf(num, sizeof num, "%d", o.len); cig_pool.append(num, (size_t)m); cig_pool.push_back(o.op); }
        kp.cig_len = (uint32_t)cig_pool.size() - kp.cig_off;
        kp.seq = r->seq + cr.lead; kp.seq_len = seq_len;
        kp.name = r->qname; kp.name_len = r->qlen; kp.suffix = suffix;
        kept.push_back(kp);
    }
    // ---- exact duplicates collapse; unique reads in (position, CIGAR text, bases) order (std::map<AlignRead,...>, :594-627)
    const size_t nk = kept.size();
    std::vector<uint32_t> order(nk);
    std::iota(order.begin(), order.end(), 0u);
    auto cmp3 = [&](const Kept& x, const Kept& y) {
        if (x.relpos != y.relpos) return x.relpos < y.relpos ? -1 : 1;
        const size_t cl = std::min(x.cig_len, y.cig_len);
        int c = memcmp(cig_pool.data() + x.cig_off, cig_pool.data() + y.cig_off, cl);
        if (c == 0 && x.cig_len != y.cig_len) c = x.cig_len < y.cig_len ? -1 : 1;
        if (c != 0) return c;
        const int sl = std::min(x.seq_len, y.seq_len);
        c = memcmp(x.seq, y.seq, (size_t)sl);
        if (c == 0 && x.seq_len != y.seq_len) c = x.seq_len < y.seq_len ? -1 : 1;
        return c;
    };
    stable_sort_mt(order, [&](uint32_t i, uint32_t j) { return cmp3(kept[i], kept[j]) < 0; });
    std::vector<int> uid_of(nk);
    int n_uniq = 0;
    for (size_t k = 0; k < nk; k++) {
        const Kept& cur = kept[order[k]];
        if (k == 0 || cmp3(kept[order[k - 1]], cur) != 0) {
            R->pos.push_back(cur.relpos);
            R->cigar_off.push_back((int)R->cigar_text.size());
            R->cigar_text.append(cig_pool.data() + cur.cig_off, cur.cig_len);
            R->seq_off.push_back((int)R->seq_text.size());
            R->seq_text.append(cur.seq, (size_t)cur.seq_len);
            R->copies.push_back(0);
            n_uniq++;
        }
        R->copies.back()++;
        uid_of[order[k]] = n_uniq - 1;
    }
    R->cigar_off.push_back((int)R->cigar_text.size());
    R->seq_off.push_back((int)R->seq_text.size());
    // ---- mates (:630-665): names in byte order; a name met again keeps its last unique read; mate = the same name
    // with the other /1 /2 ending
    struct Named { uint32_t k; };           // k: index into `order` = assignment sequence (unique-read order, then input order)
    std::vector<uint32_t> by_name(nk);
    std::iota(by_name.begin(), by_name.end(), 0u);
    auto name_cmp = [&](const Kept& x, const Kept& y) {
        // compares name + ("/" suffix): the suffixed names as strings
        const int l = std::min(x.name_len, y.name_len);
        int c = memcmp(x.name, y.name, (size_t)l);
        if (c != 0) return c;
        // one name is a prefix of the other (or equal): compare the remainders, where a remainder is the rest of the
        // name followed by "/s" when the read is paired
        auto at = [](const Kept& z, int i) -> int {   // character i of the suffixed name, -1 past the end
            if (i < z.name_len) return (unsigned char)z.name[i];
            if (!z.suffix) return -1;
            if (i == z.name_len) return '/';
            if (i == z.name_len + 1) return (unsigned char)z.suffix;
            return -1;
        };
        for (int i = l;; i++) {
            const int cx = at(x, i), cy = at(y, i);
            if (cx != cy) return cx < cy ? -1 : 1;
            if (cx < 0) return 0;
        }
    };
    // by_name holds positions in `order`
    stable_sort_mt(by_name, [&](uint32_t i, uint32_t j) { return name_cmp(kept[order[i]], kept[order[j]]) < 0; });
    // distinct names, each with the uid assigned last
    std::vector<uint32_t> names;            // positions in `order`, one per distinct name (the last assignment)
    for (size_t k = 0; k < nk; k++) {
        if (k + 1 < nk && name_cmp(kept[order[by_name[k]]], kept[order[by_name[k + 1]]]) == 0) continue;
        names.push_back(by_name[k]);
    }
    std::vector<std::vector<int>> mates((size_t)n_uniq);
    for (uint32_t pos_in_order : names) {
        const Kept& kp = kept[order[pos_in_order]];
        const int uid = uid_of[order[pos_in_order]];
        // paired: the suffixed name ends in "/1" or "/2" -- by the flag, or because the read name itself does
        char ending = kp.suffix;
        int base_len = kp.name_len;
        if (!ending && kp.name_len >= 2 && kp.name[kp.name_len - 2] == '/' && (kp.name[kp.name_len - 1] == '1' || kp.name[kp.name_len - 1] == '2')) {
            ending = kp.name[kp.name_len - 1];
            base_len = kp.name_len - 2;
        }
        int mate = -1;
        if (ending == '1' || ending == '2') {
            // binary search for base + "/" + other among the distinct names
            Kept probe = kp;
            std::string pn;
            if (kp.suffix) { probe.suffix = ending == '1' ? '2' : '1'; }
            else { pn.assign(kp.name, (size_t)base_len); pn += '/'; pn += (ending == '1' ? '2' : '1'); probe.name = pn.data(); probe.name_len = (int)pn.size(); probe.suffix = 0; }
            size_t lo = 0, hi = names.size();
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if (name_cmp(kept[order[names[mid]]], probe) < 0) lo = mid + 1; else hi = mid;
            }
            if (lo < names.size() && name_cmp(kept[order[names[lo]]], probe) == 0) mate = uid_of[order[names[lo]]];
        }
        mates[(size_t)uid].push_back(mate);
    }
    R->mate_off.push_back(0);
    for (int u = 0; u < n_uniq; u++) {
        for (int m : mates[(size_t)u]) R->mate_idx.push_back(m);
        R->mate_off.push_back((int)R->mate_idx.size());
    }
    *out = R.release();
    return SC_OK;
}

int sc_reads_get(sc_reads* r, int* n_reads, const int** pos, const char** cigar_text, const int** cigar_off, const char** seq_text,
                 const int** seq_off, const int** copies, const int** mate_idx, const int** mate_off, long* n_input, int* depth) {
    if (!r) return SC_ERR_ARG;
    static const int none = 0;
    if (n_reads) *n_reads = (int)r->pos.size();
    if (pos) *pos = r->pos.empty() ? &none : r->pos.data();
    if (cigar_text) *cigar_text = r->cigar_text.c_str();
    if (cigar_off) *cigar_off = r->cigar_off.data();
    if (seq_text) *seq_text = r->seq_text.c_str();
    if (seq_off) *seq_off = r->seq_off.data();
    if (copies) *copies = r->copies.empty() ? &none : r->copies.data();
    if (mate_idx) *mate_idx = r->mate_idx.empty() ? &none : r->mate_idx.data();
    if (mate_off) *mate_off = r->mate_off.data();
    if (n_input) *n_input = r->n_input;
    if (depth) *depth = r->depth;
    return SC_OK;
}

void sc_reads_free(sc_reads* r) { delete r; }

}  // extern "C"

// host_mtx.cpp -- host sparse library of the engine (layer L2 of the reference: the behaviour of
// sparse_helper.h:14-259,475-509 and of the parts of mmio.h:254-367 it uses).
//
// Own implementation written from the behavioural spec.  The whole file is read into
// memory; the entry section is tokenised and converted by all host cores (chunks cut at whitespace,
// token g belongs to field g % 3 of entry g / 3, so the line layout is as irrelevant as it is to the
// reference's fscanf loop, sparse_helper.h:135-165); the CSR/CSC arrays are then built by a parallel
// stable bucket-by-major + per-segment stable sort instead of the reference's qsort + counting +
// CSC_2_CSR pipeline.  It yields the same arrays:
//   * CSC: entries ordered by (col, row), equal keys in file order;
//   * CSR: entries ordered by (row, col), equal keys in file order
// which is what qsort(cmp_by_column_row) (glibc merge sort, stable) followed by CSC_2_CSR gives.
// SEXTANS_LOADER_THREADS overrides the thread count (default: all cores for bodies >= 1 MiB).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cstdint>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "sextans_amd.h"

namespace {

struct Cursor {
    const char *p;
    const char *end;
    void skip_ws() { while (p < end && isspace((unsigned char)*p)) ++p; }
    bool at_end() { skip_ws(); return p >= end; }
};

// scanf("%d") semantics on a NUL-terminated buffer: skip whitespace, optional sign, digits.
bool next_int(Cursor &c, int &out) {
    c.skip_ws();
    if (c.p >= c.end) return false;
    char *e = nullptr;
    long v = strtol(c.p, &e, 10);
    if (e == c.p) return false;
    c.p = e;
    out = (int)v;
    return true;
}

// Returns [line_begin, line_end) and advances past the newline.
bool next_line(Cursor &c, const char *&b, const char *&e) {
    if (c.p >= c.end) return false;
    b = c.p;
    const char *nl = (const char *)memchr(c.p, '\n', (size_t)(c.end - c.p));
    e = nl ? nl : c.end;
    c.p = nl ? nl + 1 : c.end;
    return true;
}

std::string lower(std::string s) {
    for (auto &ch : s) ch = (char)tolower((unsigned char)ch);
    return s;
}

struct Banner { bool coordinate, pattern, complex_, symmetric; };

// Banner rules of mm_read_banner (mmio.h:254-337): five whitespace-separated tokens; the first
// must START with "%%MatrixMarket" (the reference uses strncmp), the other four are matched
// case-insensitively against the fixed vocabularies.
int parse_banner(const char *b, const char *e, Banner &out) {
    std::vector<std::string> tok;
    const char *p = b;
    while (p < e && tok.size() < 5) {
        while (p < e && isspace((unsigned char)*p)) ++p;
        const char *s = p;
        while (p < e && !isspace((unsigned char)*p)) ++p;
        if (p > s) tok.emplace_back(s, p);
    }
    if (tok.size() != 5) return SEXTANS_ERR_BANNER;
    if (tok[0].compare(0, 14, "%%MatrixMarket") != 0) return SEXTANS_ERR_BANNER;
    if (lower(tok[1]) != "matrix") return SEXTANS_ERR_BANNER;
    std::string crd = lower(tok[2]), dt = lower(tok[3]), st = lower(tok[4]);
    if (crd == "coordinate") out.coordinate = true;
    else if (crd == "array") out.coordinate = false;
    else return SEXTANS_ERR_BANNER;
    out.pattern = out.complex_ = false;
    if (dt == "real" || dt == "integer") {}
    else if (dt == "complex") out.complex_ = true;
    else if (dt == "pattern") out.pattern = true;
    else return SEXTANS_ERR_BANNER;
    // Only "symmetric" mirrors: mm_is_symmetric tests 'S' (mmio.h:48); hermitian and
    // skew-symmetric files are read as general, exactly like the reference.
    if (st == "symmetric") out.symmetric = true;
    else if (st == "general" || st == "hermitian" || st == "skew-symmetric") out.symmetric = false;
    else return SEXTANS_ERR_BANNER;
    return SEXTANS_OK;
}

// ---- parallel body parser --------------------------------------------------------------------------
// The entry section is a whitespace-separated token stream (fscanf semantics: line structure does not
// matter).  It is cut into chunks at whitespace; pass 1 counts the tokens of every chunk, a prefix sum
// gives each chunk the global index of its first token, and pass 2 parses token g into field g % tpe of
// entry g / tpe (tpe = 2 for pattern files, 3 otherwise).  No token straddles a chunk, so chunks are
// independent whatever the line layout is.

inline bool is_ws(unsigned char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); }

int loader_threads(size_t body_bytes) {
    if (const char *e = getenv("SEXTANS_LOADER_THREADS")) {
        int t = atoi(e);
        if (t >= 1) return t > 64 ? 64 : t;
    }
    if (body_bytes < (1u << 20)) return 1;
    unsigned hw = std::thread::hardware_concurrency();
    return hw < 1 ? 1 : (hw > 32 ? 32 : (int)hw);
}

template <class F> void parallel_for(int T, F &&fn) {
    if (T == 1) { fn(0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    for (int t = 0; t < T; ++t) th.emplace_back([&fn, t] { fn(t); });
    for (auto &x : th) x.join();
}

// Decimal integer token, fully consumed.  Long digit strings fall back to strtol (same wrap as "%d").
inline bool parse_int_token(const char *s, const char *e, int &out) {
    const char *p = s;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); ++p; }
    if (p >= e) return false;
    if (e - p > 9) {
        char *end = nullptr;
        long v = strtol(s, &end, 10);
        if (end != e) return false;
        out = (int)v;
        return true;
    }
    int v = 0;
    for (; p < e; ++p) {
        const unsigned d = (unsigned)(*p - '0');
        if (d > 9) return false;
        v = v * 10 + (int)d;
    }
    out = neg ? -v : v;
    return true;
}

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Single-precision token with strtof's result ("%f": the correctly rounded float of the decimal text).
// Fast path: <= 19 significant digits with the integer mantissa < 2^53 and |exponent| <= 22 give the
// correctly rounded DOUBLE with one multiply or divide (both operands exact); rounding that double to
// float equals rounding the text to float unless the double sits within one double-ulp of a float
// rounding boundary -- those cases, and everything unusual (inf, nan, hex, huge exponents, subnormal or
// overflowing results), go to strtof.
inline bool parse_float_token(const char *s, const char *e, float &out) {
    const char *p = s;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); ++p; }
    uint64_t m = 0;
    int digits = 0, exp10 = 0;
    bool any = false, fast = true;
    for (; p < e && (unsigned)(*p - '0') <= 9; ++p) {
        any = true;
        if (m == 0 && *p == '0') continue;                 // leading zeros
        if (digits < 19) { m = m * 10 + (unsigned)(*p - '0'); ++digits; } else fast = false;
    }
    if (p < e && *p == '.') {
        ++p;
        for (; p < e && (unsigned)(*p - '0') <= 9; ++p) {
            any = true;
            if (m == 0 && *p == '0') { --exp10; continue; }
            if (digits < 19) { m = m * 10 + (unsigned)(*p - '0'); ++digits; --exp10; } else fast = false;
        }
    }
    if (any && p < e && (*p == 'e' || *p == 'E')) {
        const char *q = p + 1;
        bool eneg = false;
        if (q < e && (*q == '+' || *q == '-')) { eneg = (*q == '-'); ++q; }
        if (q < e && (unsigned)(*q - '0') <= 9) {
            int ex = 0;
            for (; q < e && (unsigned)(*q - '0') <= 9; ++q)
                if (ex < 100000) ex = ex * 10 + (*q - '0');
            exp10 += eneg ? -ex : ex;
            p = q;
        }
    }
    if (any && fast && p == e) {
        if (m == 0) { out = neg ? -0.0f : 0.0f; return true; }
        if (m < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
            const double d = exp10 < 0 ? (double)m / kPow10[-exp10] : (double)m * kPow10[exp10];
            if (d >= 1.1754943508222875e-38 && d <= 3.4028234663852886e+38) {
                uint64_t bits;
                memcpy(&bits, &d, 8);
                const int64_t low = (int64_t)(bits & ((1ull << 29) - 1)) - (1ll << 28);
                if (low > 1 || low < -1) {
                    const float f = (float)d;
                    out = neg ? -f : f;
                    return true;
                }
            }
        }
    }
    char *end = nullptr;                                   // buffer is NUL-terminated; tokens end at whitespace
    const float v = strtof(s, &end);
    if (end != e) return false;
    out = v;
    return true;
}

// Whole file into a NUL-terminated buffer; large files are read by all cores (pread of disjoint
// ranges: the page-cache copy is the cost, and it parallelises).
int read_whole_file(const char *path, std::string &buf) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SEXTANS_ERR_OPEN;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {            // pipes etc.: stream
        char chunk[1 << 16];
        ssize_t n;
        while ((n = read(fd, chunk, sizeof chunk)) > 0) buf.append(chunk, (size_t)n);
        close(fd);
        return SEXTANS_OK;
    }
    const size_t size = (size_t)st.st_size;
    buf.resize(size);
    const int T = loader_threads(size);
    std::vector<int> bad((size_t)T, 0);
    parallel_for(T, [&](int t) {
        size_t o = size * (size_t)t / (size_t)T;
        const size_t end = size * (size_t)(t + 1) / (size_t)T;
        while (o < end) {
            const ssize_t n = pread(fd, &buf[o], end - o, (off_t)o);
            if (n <= 0) { bad[(size_t)t] = 1; break; }
            o += (size_t)n;
        }
    });
    close(fd);
    for (int x : bad)
        if (x) return SEXTANS_ERR_OPEN;
    return SEXTANS_OK;
}

struct BodyError { int64_t entry; int code; };

// Parses up to nnz_file entries from [b, e).  Fills r/c (as written in the file, 1-based) and v.
int parse_body(const char *b, const char *e, int tpe, int64_t nnz_file, int T, std::vector<int> &r,
               std::vector<int> &c, std::vector<float> &v, BodyError &err) {
    err = {INT64_MAX, SEXTANS_OK};
    std::vector<const char *> cut((size_t)T + 1);
    cut[0] = b;
    cut[(size_t)T] = e;
    for (int t = 1; t < T; ++t) {
        const char *p = b + (size_t)(e - b) * (size_t)t / (size_t)T;
        if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
        while (p < e && !is_ws((unsigned char)*p)) ++p;
        cut[(size_t)t] = p;
    }
    std::vector<int64_t> first((size_t)T + 1, 0);
    parallel_for(T, [&](int t) {
        int64_t n = 0;
        bool in = false;
        for (const char *p = cut[(size_t)t]; p < cut[(size_t)t + 1]; ++p) {
            const bool w = is_ws((unsigned char)*p);
            n += (!w && !in);
            in = !w;
        }
        first[(size_t)t + 1] = n;
    });
    for (int t = 0; t < T; ++t) first[(size_t)t + 1] += first[(size_t)t];
    const int64_t need = nnz_file * tpe;
    if (first[(size_t)T] < need) err = {first[(size_t)T] / tpe, SEXTANS_ERR_PARSE};   // file ends early
    r.resize((size_t)nnz_file);
    c.resize((size_t)nnz_file);
    if (tpe == 3) v.resize((size_t)nnz_file);
    std::vector<BodyError> terr((size_t)T, BodyError{INT64_MAX, SEXTANS_OK});
    parallel_for(T, [&](int t) {
        int64_t g = first[(size_t)t];
        const char *p = cut[(size_t)t], *end = cut[(size_t)t + 1];
        while (g < need) {
            while (p < end && is_ws((unsigned char)*p)) ++p;
            if (p >= end) break;
            const char *s = p;
            while (p < end && !is_ws((unsigned char)*p)) ++p;
            const int64_t ent = g / tpe;
            const int field = (int)(g % tpe);
            bool ok;
            if (field == 0) ok = parse_int_token(s, p, r[(size_t)ent]);
            else if (field == 1) ok = parse_int_token(s, p, c[(size_t)ent]);
            else ok = parse_float_token(s, p, v[(size_t)ent]);
            if (!ok) { terr[(size_t)t] = {ent, SEXTANS_ERR_PARSE}; break; }
            ++g;
        }
    });
    for (const BodyError &te : terr)
        if (te.entry < err.entry) err = te;
    return SEXTANS_OK;
}

// Stable sort of COO entries by (major, minor, input order) into ptr / idx / val, in parallel:
// histogram of the major key, nnz-balanced major ranges per thread, each thread places the entries of
// its range in input order (it alone advances those cursors), then orders every segment by the minor key
// with a stable sort.  Same result as two stable counting sorts (minor, then major).
void coo_to_compressed(int nmajor, const std::vector<int> &major, const std::vector<int> &minor,
                       const std::vector<float> &val, int T, std::vector<int> &ptr, std::vector<int> &idx,
                       std::vector<float> &out_val) {
    const size_t nnz = major.size();
    ptr.assign((size_t)nmajor + 1, 0);
    for (size_t i = 0; i < nnz; ++i) ptr[(size_t)major[i] + 1]++;
    for (int k = 0; k < nmajor; ++k) ptr[(size_t)k + 1] += ptr[(size_t)k];
    idx.resize(nnz);
    out_val.resize(nnz);
    if (T > nmajor) T = nmajor > 0 ? nmajor : 1;
    std::vector<int> lo((size_t)T + 1, nmajor);
    lo[0] = 0;
    for (int t = 1; t < T; ++t) {
        const int target = (int)((uint64_t)nnz * (uint64_t)t / (uint64_t)T);
        lo[(size_t)t] = (int)(std::lower_bound(ptr.begin(), ptr.end() - 1, target) - ptr.begin());
        if (lo[(size_t)t] < lo[(size_t)t - 1]) lo[(size_t)t] = lo[(size_t)t - 1];
    }
    std::vector<int> cursor(ptr.begin(), ptr.end() - 1);
    parallel_for(T, [&](int t) {
        const int k0 = lo[(size_t)t], k1 = lo[(size_t)t + 1];
        if (k0 >= k1) return;
        for (size_t i = 0; i < nnz; ++i) {
            const int k = major[i];
            if (k < k0 || k >= k1) continue;
            const int dst = cursor[(size_t)k]++;
            idx[(size_t)dst] = minor[i];
            out_val[(size_t)dst] = val[i];
        }
        std::vector<std::pair<int, float>> tmp;
        for (int k = k0; k < k1; ++k) {
            const int a = ptr[(size_t)k], bnd = ptr[(size_t)k + 1];
            bool sorted = true;
            for (int j = a + 1; j < bnd && sorted; ++j) sorted = idx[(size_t)j - 1] <= idx[(size_t)j];
            if (sorted) continue;
            tmp.resize((size_t)(bnd - a));
            for (int j = a; j < bnd; ++j) tmp[(size_t)(j - a)] = {idx[(size_t)j], out_val[(size_t)j]};
            std::stable_sort(tmp.begin(), tmp.end(),
                             [](const std::pair<int, float> &x, const std::pair<int, float> &y) { return x.first < y.first; });
            for (int j = a; j < bnd; ++j) { idx[(size_t)j] = tmp[(size_t)(j - a)].first; out_val[(size_t)j] = tmp[(size_t)(j - a)].second; }
        }
    });
}

template <class T> T *dup(const std::vector<T> &v) {
    T *p = (T *)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

}  // namespace

extern "C" {

int sextans_mtx_read(const char *path, int format, int *M_out, int *K_out, int *nnz_out, int **ptr_out,
                     int **idx_out, float **val_out) {
    if (!path || !M_out || !K_out || !nnz_out || !ptr_out || !idx_out || !val_out ||
        (format != SEXTANS_FMT_CSR && format != SEXTANS_FMT_CSC))
        return SEXTANS_ERR_INVALID;
    std::string buf;
    if (int rc = read_whole_file(path, buf)) return rc;
    Cursor cur{buf.c_str(), buf.c_str() + buf.size()};   // c_str() is NUL-terminated for strto*

    const char *lb, *le;
    if (!next_line(cur, lb, le)) return SEXTANS_ERR_BANNER;
    Banner bn{};
    if (int rc = parse_banner(lb, le, bn)) return rc;

    // Size line (mm_read_mtx_crd_size, mmio.h:339-367): skip lines whose first character is '%';
    // the first other line should hold "M K nnz"; if it does not (e.g. it is blank) the next three
    // integers of the stream are taken instead.
    int M = 0, K = 0, nnz_file = 0;
    do {
        if (!next_line(cur, lb, le)) return SEXTANS_ERR_SIZE;
    } while (lb < le && *lb == '%');
    {
        std::string line(lb, le);
        Cursor lc{line.c_str(), line.c_str() + line.size()};
        if (!(next_int(lc, M) && next_int(lc, K) && next_int(lc, nnz_file))) {
            if (!(next_int(cur, M) && next_int(cur, K) && next_int(cur, nnz_file)))
                return SEXTANS_ERR_SIZE;
        }
    }
    if (!bn.coordinate) return SEXTANS_ERR_NOT_COORD;
    if (bn.complex_) return SEXTANS_ERR_COMPLEX;
    if (M < 0 || K < 0 || nnz_file < 0) return SEXTANS_ERR_SIZE;

    // Entries (load_S_matrix, sparse_helper.h:112-167): parse in parallel, then in file order: drop
    // +0.0f (-0.0f is kept), range-check, 1 -> 0 based, mirror off-diagonal entries of symmetric files
    // right after the entry itself.
    const int tpe = bn.pattern ? 2 : 3;
    const int T = loader_threads((size_t)(cur.end - cur.p));
    std::vector<int> fr, fc;
    std::vector<float> fv;
    BodyError perr;
    parse_body(cur.p, cur.end, tpe, nnz_file, T, fr, fc, fv, perr);
    const int64_t n_ok = perr.entry < (int64_t)nnz_file ? perr.entry : (int64_t)nnz_file;   // entries before the first parse error
    std::vector<int64_t> base((size_t)T + 1, 0);
    std::vector<BodyError> rerr((size_t)T, BodyError{INT64_MAX, SEXTANS_OK});
    auto classify = [&](int64_t i, bool &keep, bool &mirror) {
        uint32_t bits = 0x3f800000u;                                  // pattern -> 1.0f
        if (tpe == 3) memcpy(&bits, &fv[(size_t)i], 4);
        keep = bits != 0;
        mirror = keep && bn.symmetric && fr[(size_t)i] != fc[(size_t)i];
    };
    parallel_for(T, [&](int t) {
        const int64_t i0 = n_ok * t / T, i1 = n_ok * (t + 1) / T;
        int64_t n = 0;
        for (int64_t i = i0; i < i1; ++i) {
            bool keep, mirror;
            classify(i, keep, mirror);
            // the mirrored entry (c, r) of a symmetric file must fit too: on a size line with M != K the reference
            // writes it past the end of its arrays (sparse_helper.h:155-161); here it is an index error
            if (keep && (fr[(size_t)i] < 1 || fc[(size_t)i] < 1 || fr[(size_t)i] > M || fc[(size_t)i] > K ||
                         (mirror && (fc[(size_t)i] > M || fr[(size_t)i] > K)))) {
                rerr[(size_t)t] = {i, SEXTANS_ERR_INDEX};
                break;
            }
            n += keep + mirror;
        }
        base[(size_t)t + 1] = n;
    });
    BodyError first_err = perr;
    for (const BodyError &re : rerr)
        if (re.entry < first_err.entry) first_err = re;                // an entry is parsed before it is checked
    if (first_err.code != SEXTANS_OK && first_err.entry < (int64_t)nnz_file) return first_err.code;
    for (int t = 0; t < T; ++t) base[(size_t)t + 1] += base[(size_t)t];
    if (base[(size_t)T] > 0x7fffffffLL) return SEXTANS_ERR_SIZE;
    const int nnz = (int)base[(size_t)T];
    std::vector<int> row((size_t)nnz), col((size_t)nnz);
    std::vector<float> cv((size_t)nnz);
    parallel_for(T, [&](int t) {
        const int64_t i0 = n_ok * t / T, i1 = n_ok * (t + 1) / T;
        size_t o = (size_t)base[(size_t)t];
        for (int64_t i = i0; i < i1; ++i) {
            bool keep, mirror;
            classify(i, keep, mirror);
            if (!keep) continue;
            const float x = tpe == 3 ? fv[(size_t)i] : 1.0f;
            row[o] = fr[(size_t)i] - 1; col[o] = fc[(size_t)i] - 1; cv[o] = x; ++o;
            if (mirror) { row[o] = fc[(size_t)i] - 1; col[o] = fr[(size_t)i] - 1; cv[o] = x; ++o; }
        }
    });
    { std::vector<int>().swap(fr); std::vector<int>().swap(fc); std::vector<float>().swap(fv); }

    const bool csr = (format == SEXTANS_FMT_CSR);
    std::vector<int> ptr, idx;
    std::vector<float> val;
    coo_to_compressed(csr ? M : K, csr ? row : col, csr ? col : row, cv, T, ptr, idx, val);
    *M_out = M; *K_out = K; *nnz_out = nnz;
    *ptr_out = dup(ptr); *idx_out = dup(idx); *val_out = dup(val);
    if (!*ptr_out || !*idx_out || !*val_out) return SEXTANS_ERR_ALLOC;
    return SEXTANS_OK;
}

void sextans_host_free(void *p) { free(p); }

// ---- binary matrix cache ------------------------------------------------------------------------------
// Header (little endian, 64 bytes): "SXTCSR01", int32 format, M, K, nnz, int64 src_size, src_mtime_ns,
// 24 reserved bytes; then ptr[(format == CSR ? M : K) + 1], idx[nnz], val[nnz].
namespace {
struct CacheHeader {
    char magic[8];
    int32_t format, M, K, nnz;
    int64_t src_size, src_mtime_ns;
    char reserved[24];
};
static_assert(sizeof(CacheHeader) == 64, "cache header layout");
const char kCacheMagic[8] = {'S', 'X', 'T', 'C', 'S', 'R', '0', '1'};

bool read_fully(int fd, void *dst, size_t bytes, off_t off) {
    const int T = loader_threads(bytes);
    std::vector<int> bad((size_t)T, 0);
    parallel_for(T, [&](int t) {
        size_t o = bytes * (size_t)t / (size_t)T;
        const size_t end = bytes * (size_t)(t + 1) / (size_t)T;
        while (o < end) {
            const ssize_t n = pread(fd, (char *)dst + o, end - o, off + (off_t)o);
            if (n <= 0) { bad[(size_t)t] = 1; break; }
            o += (size_t)n;
        }
    });
    for (int x : bad)
        if (x) return false;
    return true;
}

int save_cache(const char *path, int format, int M, int K, int nnz, const int *ptr, const int *idx,
               const float *val, int64_t src_size, int64_t src_mtime_ns) {
    if (!path || (format != SEXTANS_FMT_CSR && format != SEXTANS_FMT_CSC) || M < 0 || K < 0 || nnz < 0 || !ptr ||
        (nnz > 0 && (!idx || !val)))
        return SEXTANS_ERR_INVALID;
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return SEXTANS_ERR_OPEN;
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kCacheMagic, 8);
    h.format = format; h.M = M; h.K = K; h.nnz = nnz;
    h.src_size = src_size; h.src_mtime_ns = src_mtime_ns;
    const size_t np = (size_t)(format == SEXTANS_FMT_CSR ? M : K) + 1;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(ptr, sizeof(int), np, f) == np;
    if (nnz) ok = ok && fwrite(idx, sizeof(int), (size_t)nnz, f) == (size_t)nnz &&
                  fwrite(val, sizeof(float), (size_t)nnz, f) == (size_t)nnz;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return SEXTANS_ERR_OPEN; }
    return SEXTANS_OK;
}

int load_cache(const char *path, int *format, int *M, int *K, int *nnz, int **ptr, int **idx, float **val,
               int64_t *src_size, int64_t *src_mtime_ns) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return SEXTANS_ERR_OPEN;
    CacheHeader h;
    struct stat st;
    if (fstat(fd, &st) != 0 || pread(fd, &h, sizeof h, 0) != (ssize_t)sizeof h || memcmp(h.magic, kCacheMagic, 8) != 0 ||
        (h.format != SEXTANS_FMT_CSR && h.format != SEXTANS_FMT_CSC) || h.M < 0 || h.K < 0 || h.nnz < 0) {
        close(fd);
        return SEXTANS_ERR_PARSE;
    }
    const size_t np = (size_t)(h.format == SEXTANS_FMT_CSR ? h.M : h.K) + 1;
    const size_t want = sizeof h + 4 * np + 8 * (size_t)h.nnz;
    if ((size_t)st.st_size != want) { close(fd); return SEXTANS_ERR_PARSE; }
    int *p = (int *)malloc(4 * np), *i = (int *)malloc(4 * (size_t)(h.nnz ? h.nnz : 1));
    float *v = (float *)malloc(4 * (size_t)(h.nnz ? h.nnz : 1));
    bool ok = p && i && v;
    if (!ok) { free(p); free(i); free(v); close(fd); return SEXTANS_ERR_ALLOC; }
    ok = read_fully(fd, p, 4 * np, sizeof h);
    if (h.nnz) ok = ok && read_fully(fd, i, 4 * (size_t)h.nnz, (off_t)(sizeof h + 4 * np)) &&
                    read_fully(fd, v, 4 * (size_t)h.nnz, (off_t)(sizeof h + 4 * np + 4 * (size_t)h.nnz));
    close(fd);
    ok = ok && p[0] == 0 && p[np - 1] == h.nnz;
    if (!ok) { free(p); free(i); free(v); return SEXTANS_ERR_PARSE; }
    *format = h.format; *M = h.M; *K = h.K; *nnz = h.nnz;
    *ptr = p; *idx = i; *val = v;
    if (src_size) *src_size = h.src_size;
    if (src_mtime_ns) *src_mtime_ns = h.src_mtime_ns;
    return SEXTANS_OK;
}
}  // namespace

int sextans_matrix_save(const char *path, int format, int M, int K, int nnz, const int *ptr, const int *idx,
                        const float *val) {
    return save_cache(path, format, M, K, nnz, ptr, idx, val, 0, 0);
}

int sextans_matrix_load(const char *path, int *format, int *M, int *K, int *nnz, int **ptr, int **idx,
                        float **val) {
    if (!path || !format || !M || !K || !nnz || !ptr || !idx || !val) return SEXTANS_ERR_INVALID;
    return load_cache(path, format, M, K, nnz, ptr, idx, val, nullptr, nullptr);
}

// Matrix-Market writer for the tools (holdout matrices written to disk and run through the CLI): `real general`, one entry per
// line in CSR order, %.9g (every fp32 value round-trips).  Row ranges are formatted by all cores into per-thread buffers and
// written in order.  The reference's mmio.h writers (mm_write_mtx_crd) are never called by its host (out of scope).
int sextans_mtx_write(const char *path, int M, int K, const int *row_ptr, const int *col_idx, const float *val) {
    if (!path || M < 0 || K < 0 || !row_ptr || (row_ptr[M] > 0 && (!col_idx || !val))) return SEXTANS_ERR_INVALID;
    FILE *f = fopen(path, "w");
    if (!f) return SEXTANS_ERR_OPEN;
    fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n", M, K, row_ptr[M]);
    const int T = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    const int chunk_rows = 65536;
    bool ok = true;
    for (int base = 0; base < M && ok; base += chunk_rows * T) {
        std::vector<std::string> out((size_t)T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                const int r0 = std::min(M, base + t * chunk_rows), r1 = std::min(M, r0 + chunk_rows);
                std::string &o = out[(size_t)t];
                o.reserve((size_t)(row_ptr[r1] - row_ptr[r0]) * 28);
                char line[64];
                for (int r = r0; r < r1; ++r)
                    for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
                        o.append(line, (size_t)snprintf(line, sizeof line, "%d %d %.9g\n", r + 1, col_idx[j] + 1, (double)val[j]));
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < T && ok; ++t) ok = out[(size_t)t].empty() || fwrite(out[(size_t)t].data(), 1, out[(size_t)t].size(), f) == out[(size_t)t].size();
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? SEXTANS_OK : SEXTANS_ERR_OPEN;
}

int sextans_mtx_read_cached(const char *path, const char *cache_path, int format, int *M, int *K, int *nnz,
                            int **ptr, int **idx, float **val, int *cache_hit) {
    if (!path || (format != SEXTANS_FMT_CSR && format != SEXTANS_FMT_CSC)) return SEXTANS_ERR_INVALID;
    if (cache_hit) *cache_hit = 0;
    struct stat st;
    if (stat(path, &st) != 0) return SEXTANS_ERR_OPEN;
    const int64_t size = (int64_t)st.st_size;
    const int64_t mtime = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    const std::string cpath = cache_path ? std::string(cache_path)
                                         : std::string(path) + (format == SEXTANS_FMT_CSR ? ".csr.sxbin" : ".csc.sxbin");
    int f2 = -1;
    int64_t csize = -1, cmtime = -1;
    if (load_cache(cpath.c_str(), &f2, M, K, nnz, ptr, idx, val, &csize, &cmtime) == SEXTANS_OK) {
        if (f2 == format && csize == size && cmtime == mtime) {
            if (cache_hit) *cache_hit = 1;
            return SEXTANS_OK;
        }
        free(*ptr); free(*idx); free(*val);                          // stale or other format: rebuild
    }
    if (int rc = sextans_mtx_read(path, format, M, K, nnz, ptr, idx, val)) return rc;
    (void)save_cache(cpath.c_str(), format, *M, *K, *nnz, *ptr, *idx, *val, size, mtime);   // best effort
    return SEXTANS_OK;
}

int sextans_csc_to_csr(int M, int K, int nnz, const int *col_ptr, const int *row_idx,
                       const float *csc_val, int *row_ptr, int *col_idx, float *csr_val) {
    if (M < 0 || K < 0 || nnz < 0 || !col_ptr || !row_ptr) return SEXTANS_ERR_INVALID;
    // One stable counting sort by row over the CSC order: the write cursor of each row only moves
    // forward while columns are visited in ascending order.
    std::vector<int> cursor((size_t)M + 1, 0);
    for (int j = 0; j < nnz; ++j) {
        if (row_idx[j] < 0 || row_idx[j] >= M) return SEXTANS_ERR_INDEX;
        cursor[(size_t)row_idx[j] + 1]++;
    }
    for (int r = 0; r < M; ++r) cursor[(size_t)r + 1] += cursor[(size_t)r];
    for (int r = 0; r <= M; ++r) row_ptr[r] = cursor[(size_t)r];
    for (int c = 0; c < K; ++c) {
        for (int j = col_ptr[c]; j < col_ptr[c + 1]; ++j) {
            int dst = cursor[(size_t)row_idx[j]]++;
            col_idx[dst] = c;
            csr_val[dst] = csc_val[j];
        }
    }
    return SEXTANS_OK;
}

void sextans_init_dense_B(int K, int N, float *B) {
    const size_t n = (size_t)K * (size_t)N;
    for (size_t i = 0; i < n; ++i) B[i] = 1.0f;
}

void sextans_init_dense_C(int M, int N, float *C) {
    for (int n = 0; n < N; ++n) {
        float *col = C + (size_t)M * n;
        for (int m = 0; m < M; ++m) col[m] = (float)(1.0 * (m + 1) * (n + 1) / M / N);
    }
}

int sextans_round_up_n(int N) { return (N + 7) / 8 * 8; }

int sextans_verify(int M, int N, const float *c_cpu, const float *c_dev, float *percent) {
    int bad = 0;
    const size_t total = (size_t)M * (size_t)N;
    for (size_t i = 0; i < total; ++i) {
        const float a = c_cpu[i], b = c_dev[i];
        const float diff = std::fabs(a - b);
        const float floor_ = (float)((double)std::fmin(std::fabs(a), std::fabs(b)) + 1e-4);
        if ((double)(diff / floor_) > 1e-4) ++bad;
    }
    if (percent) *percent = (float)(100.0 * bad / M / N);
    return bad;
}

double sextans_gflops(int M, int N, int64_t nnz, double seconds) {
    return 2.0 * N * ((double)nnz + (double)M) / 1e9 / seconds;
}

// CLI self-check golden (see header).  Row-at-a-time with a small stack/heap accumulator; the
// arithmetic (fp32 product rounded, then added, CSR order; alpha*psum + beta*c) is the
// reference's (sparse_helper.h:279-289); this TU is compiled with -ffp-contract=off.
int sextans_selfcheck_golden(int M, int N, int K, float alpha, const int *row_ptr,
                             const int *col_idx, const float *val, const float *B, float beta,
                             float *C) {
    if (M < 0 || N <= 0 || K < 0) return SEXTANS_ERR_INVALID;
    std::vector<float> acc((size_t)N);
    for (int m = 0; m < M; ++m) {
        std::fill(acc.begin(), acc.end(), 0.0f);
        for (int j = row_ptr[m]; j < row_ptr[m + 1]; ++j) {
            const float a = val[j];
            const size_t k = (size_t)col_idx[j];
            for (int n = 0; n < N; ++n) {
                const float prod = a * B[k + (size_t)K * n];
                acc[(size_t)n] = acc[(size_t)n] + prod;
            }
        }
        for (int n = 0; n < N; ++n) {
            float &c = C[(size_t)m + (size_t)M * n];
            const float t0 = alpha * acc[(size_t)n];
            const float t1 = beta * c;
            c = t0 + t1;
        }
    }
    return SEXTANS_OK;
}

const char *sextans_error_string(int code) {
    switch (code) {
        case SEXTANS_OK: return "ok";
        case SEXTANS_ERR_OPEN: return "could not open matrix file";
        case SEXTANS_ERR_BANNER: return "could not process Matrix Market banner";
        case SEXTANS_ERR_SIZE: return "could not read Matrix Market size line";
        case SEXTANS_ERR_NOT_COORD: return "not a coordinate file";
        case SEXTANS_ERR_COMPLEX: return "complex matrices are not supported";
        case SEXTANS_ERR_INDEX: return "index out of range";
        case SEXTANS_ERR_ALLOC: return "out of memory";
        case SEXTANS_ERR_PARSE: return "malformed matrix entry";
        case SEXTANS_ERR_INVALID: return "invalid argument";
        case SEXTANS_ERR_NO_DEVICE: return "no usable gfx950 HIP device (there is no CPU fallback)";
        case SEXTANS_ERR_HIP: return "HIP runtime error";
        case SEXTANS_ERR_STATE: return "engine state error (matrix not set?)";
        case SEXTANS_ERR_PEER: return "a collective preparation failed on another rank";
        default: return "unknown error";
    }
}

}  // extern "C"

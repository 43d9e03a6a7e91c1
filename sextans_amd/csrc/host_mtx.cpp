// host_mtx.cpp -- host sparse library of the engine (layer L2 of the reference: the behaviour of
// sparse_helper.h:14-259,475-509 and of the parts of mmio.h:254-367 it uses).
//
// Own implementation written from the behavioural spec: the whole file is read
// into memory and tokenised in one pass; the CSR/CSC arrays are then built with two stable
// counting sorts (LSD radix over (row, col)) instead of the reference's qsort + counting +
// CSC_2_CSR pipeline -- O(nnz + M + K), no comparison sort, and it yields the same arrays:
//   * CSC: entries ordered by (col, row), equal keys in file order;
//   * CSR: entries ordered by (row, col), equal keys in file order
// which is what qsort(cmp_by_column_row) (glibc merge sort, stable) followed by CSC_2_CSR gives.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

// scanf("%f") semantics: correctly rounded single precision straight from the text (strtof), not
// a double parse followed by a second rounding.
bool next_float(Cursor &c, float &out) {
    c.skip_ws();
    if (c.p >= c.end) return false;
    char *e = nullptr;
    float v = strtof(c.p, &e);
    if (e == c.p) return false;
    c.p = e;
    out = v;
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

struct Coo { std::vector<int> r, c; std::vector<float> v; };

// Stable counting sort of the permutation `perm` by key[perm[i]] in [0, nkeys).
void stable_count_sort(const std::vector<int> &key, int nkeys, const std::vector<int> &perm,
                       std::vector<int> &out, std::vector<int> *ptr_out) {
    std::vector<int> cnt((size_t)nkeys + 1, 0);
    for (int i : perm) cnt[(size_t)key[(size_t)i] + 1]++;
    for (int k = 0; k < nkeys; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
    if (ptr_out) *ptr_out = cnt;
    out.resize(perm.size());
    std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    for (int i : perm) out[(size_t)pos[(size_t)key[(size_t)i]]++] = i;
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
    FILE *f = fopen(path, "rb");
    if (!f) return SEXTANS_ERR_OPEN;
    std::string buf;
    {
        char chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, n);
        fclose(f);
    }
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

    Coo coo;
    size_t cap = (size_t)nnz_file * (bn.symmetric ? 2 : 1);
    coo.r.reserve(cap); coo.c.reserve(cap); coo.v.reserve(cap);
    for (int i = 0; i < nnz_file; ++i) {
        int r = 0, c = 0;
        float v = 1.0f;                                   // pattern -> 1.0
        if (!next_int(cur, r) || !next_int(cur, c)) return SEXTANS_ERR_PARSE;
        if (!bn.pattern && !next_float(cur, v)) return SEXTANS_ERR_PARSE;
        uint32_t bits;
        memcpy(&bits, &v, 4);
        if (bits == 0) continue;                          // +0.0f dropped, -0.0f kept
        if (r < 1 || c < 1 || r > M || c > K) return SEXTANS_ERR_INDEX;
        coo.r.push_back(r - 1); coo.c.push_back(c - 1); coo.v.push_back(v);
        if (bn.symmetric && r != c) {
            coo.r.push_back(c - 1); coo.c.push_back(r - 1); coo.v.push_back(v);
        }
    }
    const int nnz = (int)coo.v.size();

    // LSD radix: minor key first, then major key; both passes stable.
    std::vector<int> ident((size_t)nnz), p1, p2, ptr;
    for (int i = 0; i < nnz; ++i) ident[(size_t)i] = i;
    const bool csr = (format == SEXTANS_FMT_CSR);
    const std::vector<int> &minor = csr ? coo.c : coo.r;
    const std::vector<int> &major = csr ? coo.r : coo.c;
    stable_count_sort(minor, csr ? K : M, ident, p1, nullptr);
    stable_count_sort(major, csr ? M : K, p1, p2, &ptr);

    std::vector<int> idx((size_t)nnz);
    std::vector<float> val((size_t)nnz);
    for (int i = 0; i < nnz; ++i) {
        idx[(size_t)i] = minor[(size_t)p2[(size_t)i]];
        val[(size_t)i] = coo.v[(size_t)p2[(size_t)i]];
    }
    *M_out = M; *K_out = K; *nnz_out = nnz;
    *ptr_out = dup(ptr); *idx_out = dup(idx); *val_out = dup(val);
    if (!*ptr_out || !*idx_out || !*val_out) return SEXTANS_ERR_ALLOC;
    return SEXTANS_OK;
}

void sextans_host_free(void *p) { free(p); }

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
        default: return "unknown error";
    }
}

}  // extern "C"

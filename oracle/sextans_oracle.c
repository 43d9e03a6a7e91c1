/*
 * sextans_oracle.c -- CPU restatement of the Sextans host path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the HIP engine in sextans_amd/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (sextans_amd/, include/) never links, imports or calls anything in oracle/.
 *
 * It restates, in plain C, the arithmetic and the loader semantics of the reference
 * (linghaosong/Sextans, branch tapa).  Each function cites the reference file:line it
 * follows.  Parity pinning: tests/test_oracle_vs_reference.py checks every function here
 * bit-for-bit against the reference's own header-only host library built by
 * oracle/Makefile into oracle/_ref/libsextans_ref.so, and tests/test_oracle_golden.py
 * checks it against the committed golden vectors in tests/golden/ (which were produced
 * by that reference build; SURVEY.md section 8c lists the known answers).
 *
 * Differences from the reference that are deliberate:
 *   - errors are returned as codes (the reference prints and exit(1)s,
 *     sparse_helper.h:100-109,120-123,146-149,181-191);
 *   - nothing is printed.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).  -ffp-contract=off
 * keeps `a*b + c` as a rounded multiply followed by a rounded add, which is what the
 * reference's `g++ -O2` build (README.md:28) produces on baseline x86-64.
 */
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_OK 0
#define ORC_ERR_OPEN 1        /* sparse_helper.h:181-184 */
#define ORC_ERR_BANNER 2      /* sparse_helper.h:100-103 */
#define ORC_ERR_SIZE 3        /* sparse_helper.h:105-109 */
#define ORC_ERR_NOT_COORD 4   /* sparse_helper.h:188-191 */
#define ORC_ERR_COMPLEX 5     /* sparse_helper.h:120-123 */
#define ORC_ERR_INDEX 6       /* sparse_helper.h:146-149 */
#define ORC_ERR_ALLOC 7

#define ORC_CSR 0             /* enum MATRIX_FORMAT, sparse_helper.h:20 */
#define ORC_CSC 1

#define ORC_LINE 1025         /* MM_MAX_LINE_LENGTH, mmio.h:17 */
#define ORC_TOKEN 64          /* MM_MAX_TOKEN_LENGTH, mmio.h:19 */

/* ---- mm_read_banner, mmio.h:254-337: 4-char typecode {M, C|A, R|C|P|I, G|S|H|K}. ---- */
static void lower(char *p) { for (; *p; ++p) *p = (char)tolower((unsigned char)*p); }

int orc_mm_read_banner(FILE *f, char tc[4]) {
    char line[ORC_LINE], banner[ORC_TOKEN], mtx[ORC_TOKEN], crd[ORC_TOKEN], dt[ORC_TOKEN],
        st[ORC_TOKEN];
    tc[0] = tc[1] = tc[2] = ' '; tc[3] = 'G';               /* mm_clear_typecode, mmio.h:73-74 */
    if (!fgets(line, ORC_LINE, f)) return 12;               /* MM_PREMATURE_EOF, mmio.h:266-267 */
    if (sscanf(line, "%s %s %s %s %s", banner, mtx, crd, dt, st) != 5) return 12; /* :269-271 */
    lower(mtx); lower(crd); lower(dt); lower(st);           /* :273-276 */
    if (strncmp(banner, "%%MatrixMarket", 14) != 0) return 14;   /* MM_NO_HEADER :279-280 */
    if (strcmp(mtx, "matrix") != 0) return 15;              /* MM_UNSUPPORTED_TYPE :283-284 */
    tc[0] = 'M';
    if (strcmp(crd, "coordinate") == 0) tc[1] = 'C';        /* :292-298 */
    else if (strcmp(crd, "array") == 0) tc[1] = 'A';
    else return 15;
    if (strcmp(dt, "real") == 0) tc[2] = 'R';               /* :303-315 */
    else if (strcmp(dt, "complex") == 0) tc[2] = 'C';
    else if (strcmp(dt, "pattern") == 0) tc[2] = 'P';
    else if (strcmp(dt, "integer") == 0) tc[2] = 'I';
    else return 15;
    if (strcmp(st, "general") == 0) tc[3] = 'G';            /* :320-333 */
    else if (strcmp(st, "symmetric") == 0) tc[3] = 'S';
    else if (strcmp(st, "hermitian") == 0) tc[3] = 'H';
    else if (strcmp(st, "skew-symmetric") == 0) tc[3] = 'K';
    else return 15;
    return 0;
}

/* ---- mm_read_mtx_crd_size, mmio.h:339-367: first non-'%' line; blank lines tolerated. ---- */
int orc_mm_read_crd_size(FILE *f, int *M, int *N, int *nz) {
    char line[ORC_LINE];
    int got;
    *M = *N = *nz = 0;
    do {
        if (!fgets(line, ORC_LINE, f)) return 12;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", M, N, nz) == 3) return 0;
    do {
        got = fscanf(f, "%d %d %d", M, N, nz);
        if (got == EOF) return 12;
    } while (got != 3);
    return 0;
}

/* rcv record + comparators, sparse_helper.h:14-18, :37-62. */
typedef struct { int r, c; float v; } orc_rcv;

static int cmp_row_col(const void *aa, const void *bb) {
    const orc_rcv *a = (const orc_rcv *)aa, *b = (const orc_rcv *)bb;
    if (a->r > b->r) return +1;
    if (a->r < b->r) return -1;
    if (a->c > b->c) return +1;
    if (a->c < b->c) return -1;
    return 0;
}
static int cmp_col_row(const void *aa, const void *bb) {
    const orc_rcv *a = (const orc_rcv *)aa, *b = (const orc_rcv *)bb;
    if (a->c > b->c) return +1;
    if (a->c < b->c) return -1;
    if (a->r > b->r) return +1;
    if (a->r < b->r) return -1;
    return 0;
}

/*
 * read_suitsparse_matrix, sparse_helper.h:169-259 (with mm_init_read :89-110,
 * load_S_matrix :112-167, sort_by_fn :65-87).
 *
 * Output arrays are malloc'ed; the caller frees them with orc_free().
 *   fmt = ORC_CSR: ptr has M+1 entries, idx = column indices
 *   fmt = ORC_CSC: ptr has K+1 entries, idx = row indices
 */
int orc_read_suitsparse_matrix(const char *path, int fmt, int *M_out, int *K_out, int *nnz_out,
                               int **ptr_out, int **idx_out, float **val_out) {
    FILE *f = fopen(path, "r");
    char tc[4];
    int M, K, nnz_mmio;
    if (!f) return ORC_ERR_OPEN;                                        /* :181-184 */
    if (orc_mm_read_banner(f, tc) != 0) { fclose(f); return ORC_ERR_BANNER; }      /* :100-103 */
    if (orc_mm_read_crd_size(f, &M, &K, &nnz_mmio) != 0) { fclose(f); return ORC_ERR_SIZE; }
    if (tc[1] != 'C') { fclose(f); return ORC_ERR_NOT_COORD; }           /* :188-191 */
    if (tc[2] == 'C') { fclose(f); return ORC_ERR_COMPLEX; }             /* :120-123 */

    int symmetric = (tc[3] == 'S');                /* mm_is_symmetric tests 'S' only, mmio.h:48 */
    int pattern = (tc[2] == 'P');
    long nalloc = symmetric ? 2L * nnz_mmio : nnz_mmio;                  /* :193 */
    orc_rcv *e = (orc_rcv *)malloc(sizeof(orc_rcv) * (size_t)(nalloc > 0 ? nalloc : 1));
    if (!e) { fclose(f); return ORC_ERR_ALLOC; }

    int idx = 0, r_idx = 0, c_idx = 0;
    float value = 0.0f;
    for (int i = 0; i < nnz_mmio; ++i) {                                 /* :135-165 */
        if (pattern) {
            if (fscanf(f, "%d %d\n", &r_idx, &c_idx) < 0) { /* ignored like the reference */ }
            value = 1.0f;                                                /* :136-138 */
        } else {
            if (fscanf(f, "%d %d %f\n", &r_idx, &c_idx, &value) < 0) { }  /* :140 */
        }
        uint32_t bits;
        memcpy(&bits, &value, 4);
        if (bits != 0) {                           /* drops +0.0 only; -0.0 is kept, :143-145 */
            if (r_idx < 1 || c_idx < 1) { free(e); fclose(f); return ORC_ERR_INDEX; } /* :146 */
            e[idx].r = r_idx - 1; e[idx].c = c_idx - 1; e[idx].v = value; idx++;
            if (symmetric && r_idx != c_idx) {                           /* :156-163 */
                e[idx].r = c_idx - 1; e[idx].c = r_idx - 1; e[idx].v = value; idx++;
            }
        }
    }
    fclose(f);
    int nnz = idx;                                                       /* :166 */

    /* sort_by_fn, :65-87: libc qsort over {r,c,v} records. */
    qsort(e, (size_t)nnz, sizeof(orc_rcv), fmt == ORC_CSR ? cmp_row_col : cmp_col_row);

    int MK = (fmt == ORC_CSR) ? M : K;                                   /* :217 */
    int *ptr = (int *)calloc((size_t)MK + 1, sizeof(int));
    int *ind = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    float *val = (float *)malloc(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1));
    if (!ptr || !ind || !val) { free(e); free(ptr); free(ind); free(val); return ORC_ERR_ALLOC; }
    for (int i = 0; i < nnz; ++i) {                                      /* :221-242 */
        int key = (fmt == ORC_CSR) ? e[i].r : e[i].c;
        ptr[key + 1]++;
    }
    for (int i = 1; i <= MK; ++i) ptr[i] += ptr[i - 1];
    for (int i = 0; i < nnz; ++i) {                                      /* :244-253 */
        ind[i] = (fmt == ORC_CSR) ? e[i].c : e[i].r;
        val[i] = e[i].v;
    }
    free(e);
    *M_out = M; *K_out = K; *nnz_out = nnz;
    *ptr_out = ptr; *idx_out = ind; *val_out = val;
    return ORC_OK;
}

void orc_free(void *p) { free(p); }

/* ---- CSC_2_CSR, sparse_helper.h:475-509: histogram, prefix sum, column-ordered scatter. ---- */
void orc_csc_to_csr(int M, int K, int nnz, const int *col_ptr, const int *row_idx,
                    const float *csc_val, int *row_ptr /* M+1 */, int *col_idx, float *csr_val) {
    for (int i = 0; i <= M; ++i) row_ptr[i] = 0;
    for (int i = 0; i < nnz; ++i) row_ptr[row_idx[i] + 1]++;             /* :488-490 */
    for (int i = 0; i < M; ++i) row_ptr[i + 1] += row_ptr[i];            /* :492-494 */
    int *row_nz = (int *)calloc((size_t)(M > 0 ? M : 1), sizeof(int));   /* :496 */
    for (int c = 0; c < K; ++c) {                                        /* :497-508 */
        for (int j = col_ptr[c]; j < col_ptr[c + 1]; ++j) {
            int r = row_idx[j];
            int pos = row_ptr[r] + row_nz[r];
            csr_val[pos] = csc_val[j];
            col_idx[pos] = c;
            row_nz[r]++;
        }
    }
    free(row_nz);
}

/*
 * ---- cpu_spmm_CSR, sparse_helper.h:262-290: THE ORACLE. ----
 * C = ALPHA * A * B + BETA * C, B (K x N) and C (M x N) column major, fp32 throughout,
 * per-row psum[N] accumulated in CSR order, product rounded then added (no FMA),
 * epilogue ALPHA*psum + BETA*C.
 */
void orc_cpu_spmm_csr(int M, int N, int K, int nnz, float alpha, const int *row_ptr,
                      const int *col_idx, const float *val, const float *B, float beta, float *C) {
    (void)nnz;
    float *psum = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < M; ++i) {
        for (int nn = 0; nn < N; ++nn) psum[nn] = 0.0f;                  /* :280 */
        for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j) {              /* :281 */
            const float a = val[j];
            const float *bk = B + col_idx[j];
            for (int nn = 0; nn < N; ++nn) {                             /* :282-284 */
                psum[nn] += a * bk[(size_t)K * nn];
            }
        }
        for (int nn = 0; nn < N; ++nn) {                                 /* :286-288 */
            size_t o = (size_t)i + (size_t)M * nn;
            C[o] = alpha * psum[nn] + beta * C[o];
        }
    }
    free(psum);
}

/* Row-range variant of the same loop nest (rows [r0, r1) only): used for the bounded-sample
 * CPU baseline and for the world_size>1 gloo tests.  Same arithmetic as above. */
void orc_cpu_spmm_csr_rows(int r0, int r1, int M, int N, int K, float alpha, const int *row_ptr,
                           const int *col_idx, const float *val, const float *B, float beta,
                           float *C) {
    float *psum = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int i = r0; i < r1; ++i) {
        for (int nn = 0; nn < N; ++nn) psum[nn] = 0.0f;
        for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j) {
            const float a = val[j];
            const float *bk = B + col_idx[j];
            for (int nn = 0; nn < N; ++nn) psum[nn] += a * bk[(size_t)K * nn];
        }
        for (int nn = 0; nn < N; ++nn) {
            size_t o = (size_t)i + (size_t)M * nn;
            C[o] = alpha * psum[nn] + beta * C[o];
        }
    }
    free(psum);
}

/* The SAME loop nest with fused multiply-adds: what the engine's opt-in "exact" = 0 kernels and its fp32 matrix-core path
 * ("mfma_dense_tiles" = 2, v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, one rounding per step) compute -- psum = fmaf(a, b, psum)
 * in CSR order, epilogue fmaf(alpha, psum, beta * c).  NOT the reference's arithmetic (cpu_spmm_CSR rounds every product,
 * sparse_helper.h:283): this variant exists so that the tests can demand BIT EQUALITY between the engine's two in-tolerance paths and
 * pin both to one CPU statement; the reference bound for them stays |d| <= 1e-4 * (|alpha| sum|a b| + |beta c|) against
 * orc_cpu_spmm_csr.  fmaf() is correctly rounded by C99 whether or not the host has an FMA unit.  Row-parallel
 * (rows are independent).  TEST INFRASTRUCTURE. */
void orc_cpu_spmm_csr_fma(int M, int N, int K, float alpha, const int *row_ptr, const int *col_idx, const float *val, const float *B,
                          float beta, float *C) {
#pragma omp parallel
    {
        float psum[512];
        float *ps = N <= 512 ? psum : (float *)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < M; ++i) {
            for (int nn = 0; nn < N; ++nn) ps[nn] = 0.0f;
            for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j) {
                const float a = val[j];
                const float *bk = B + col_idx[j];
                for (int nn = 0; nn < N; ++nn) ps[nn] = fmaf(a, bk[(size_t)K * nn], ps[nn]);
            }
            for (int nn = 0; nn < N; ++nn) {
                size_t o = (size_t)i + (size_t)M * nn;
                const float bc = beta * C[o];
                C[o] = fmaf(alpha, ps[nn], bc);
            }
        }
        if (ps != psum) free(ps);
    }
}

/* Same loop nest, row-parallel on all host cores (SURVEY.md 8d baseline 2: "the same loop nest row-parallel with
 * OpenMP on all cores").  Rows are independent, every row is still summed sequentially in CSR order, so the
 * result is bit-identical to orc_cpu_spmm_csr.  Returns the wall seconds (steady clock) and writes the number
 * of threads actually used to *threads.  dynamic,256: rows of power-law matrices differ in length. */
#include <omp.h>
double orc_time_spmm_csr_omp(int M, int N, int K, float alpha, const int *row_ptr, const int *col_idx,
                             const float *val, const float *B, float beta, float *C, int *threads) {
    struct timespec t0, t1;
    int used = 1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel
    {
        float psum[512];
        float *ps = N <= 512 ? psum : (float *)malloc(sizeof(float) * (size_t)N);
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < M; ++i) {
            for (int nn = 0; nn < N; ++nn) ps[nn] = 0.0f;
            for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j) {
                const float a = val[j];
                const float *bk = B + col_idx[j];
                for (int nn = 0; nn < N; ++nn) ps[nn] += a * bk[(size_t)K * nn];
            }
            for (int nn = 0; nn < N; ++nn) {
                size_t o = (size_t)i + (size_t)M * nn;
                C[o] = alpha * ps[nn] + beta * C[o];
            }
        }
        if (ps != psum) free(ps);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (threads) *threads = used;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- dense operand init, sextans-host.cpp:94-111. ---- */
void orc_init_B(int K, int N, float *B) {
    for (int nn = 0; nn < N; ++nn)
        for (int kk = 0; kk < K; ++kk) B[(size_t)kk + (size_t)K * nn] = 1.0;      /* :102 */
}
void orc_init_C(int M, int N, float *C) {
    for (int nn = 0; nn < N; ++nn)
        for (int mm = 0; mm < M; ++mm)
            C[(size_t)mm + (size_t)M * nn] = 1.0 * (mm + 1) * (nn + 1) / M / N;   /* :109 */
}

/* ---- verification, sextans-host.cpp:262-289: returns mismatch count; pass iff pct < 2. ---- */
int orc_verify(int M, int N, const float *c_cpu, const float *c_dev, float *pct_out) {
    int mismatch = 0;
    for (int nn = 0; nn < N; ++nn) {
        for (int mm = 0; mm < M; ++mm) {
            float v_cpu = c_cpu[(size_t)mm + (size_t)nn * M];
            float v_dev = c_dev[(size_t)mm + (size_t)nn * M];
            float dff = fabsf(v_cpu - v_dev);                            /* :272 */
            float a = fabsf(v_cpu), b = fabsf(v_dev);
            float x = (a < b ? a : b) + 1e-4;                            /* :273 (double add, float store) */
            if (dff / x > 1e-4) mismatch++;                              /* :274 (compared as double) */
        }
    }
    if (pct_out) *pct_out = 100.0 * mismatch / M / N;                    /* :281 */
    return mismatch;
}

/* ---- throughput formula, sextans-host.cpp:219,255-260: 2*N*(nnz+M)/1e9/t. ---- */
double orc_gflops(int M, int N, long nnz, double seconds) {
    return 2.0 * N * ((double)nnz + M) / 1e9 / seconds;
}

/* Timed wrapper used by bench.py's cpu_baseline leg (steady clock like :207-217). */
double orc_time_spmm_rows(int r0, int r1, int M, int N, int K, float alpha, const int *row_ptr,
                          const int *col_idx, const float *val, const float *B, float beta,
                          float *C) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    orc_cpu_spmm_csr_rows(r0, r1, M, N, K, alpha, row_ptr, col_idx, val, B, beta, C);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- The accelerator's scheduled non-zero stream.
 * generate_edge_list_for_one_PE / _all_PEs, sparse_helper.h:292-403: per 4096-column window, the CSC entries
 * of the window are dealt to PE = row % 64 in CSC order; inside a PE every entry takes the first free slot
 * at or after (slot of the previous entry of the same row in this window) + 10, probing linearly (:315-327);
 * after each window all 64 lists are padded with bubbles to the longest (:390-397) and
 * edge_list_ptr[w + 1] = that length (:400).
 * edge_list_64bit, sparse_helper.h:406-473 (8 channels): word = (col & 0x3FFF) << 50 | (row & 0x3FFFF) << 32 |
 * fp32 bits, bubble = 0x3FFFF << 32 (:427-443); PE p = j + 8*cc is stored in channel j = p % 8 at slot
 * ((p/8 & 1) * 4 + (p/8 & 2) + (p/8 & 4) / 4) of 8-word group i (:458-464); every channel is
 * round_up(8 * length, 512) words, zero filled (:412-417).
 * Returns 0 and malloc'ed *ptr_out (num_windows + 1 ints) and *chan_out (8 * *chan_len words, channel c at
 * c * *chan_len).  Parity: tests/test_edge_stream.py checks the slot assignment against the reference's own
 * generate_edge_list_for_all_PEs (oracle/_ref and tests/golden/edges/); the word layout has no executable
 * reference here (edge_list_64bit needs TAPA's allocator type) and is pinned by hand-computed words. */
int orc_edge_stream(int M, int K, const int *col_ptr, const int *row_idx, const float *val, int **ptr_out,
                    uint64_t **chan_out, long *chan_len) {
    enum { NPE = 64, WIN = 4096, DIST = 10 };
    const int nwin = (K + WIN - 1) / WIN;
    int *ptr = (int *)calloc((size_t)nwin + 1, sizeof(int));
    uint64_t *pe[NPE];
    size_t cap[NPE], len[NPE];
    int *last = (int *)malloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    if (!ptr || !last) return ORC_ERR_ALLOC;
    for (int p = 0; p < NPE; ++p) { pe[p] = NULL; cap[p] = len[p] = 0; }
    const uint64_t bubble = (uint64_t)0x3FFFF << 32;
    for (int w = 0; w < nwin; ++w) {
        const int c0 = w * WIN, c1 = (c0 + WIN < K) ? c0 + WIN : K;
        size_t start[NPE], used[NPE];
        for (int p = 0; p < NPE; ++p) { start[p] = len[p]; used[p] = 0; }
        for (int r = 0; r < M; ++r) last[r] = -DIST;                       /* cycles_rows, :308 */
        for (int c = c0; c < c1; ++c)
            for (int j = col_ptr[c]; j < col_ptr[c + 1]; ++j) {
                const int r = row_idx[j], p = r % NPE;
                size_t cyc = (size_t)(last[r] + DIST);
                for (;;) {                                                  /* :318-327 */
                    if (cyc >= used[p]) {
                        const size_t need = start[p] + cyc + 1;
                        if (need > cap[p]) {
                            cap[p] = need * 2 + 64;
                            pe[p] = (uint64_t *)realloc(pe[p], cap[p] * sizeof(uint64_t));
                            if (!pe[p]) return ORC_ERR_ALLOC;
                        }
                        for (size_t t = used[p]; t <= cyc; ++t) pe[p][start[p] + t] = bubble;
                        used[p] = cyc + 1;
                    }
                    if (pe[p][start[p] + cyc] != bubble) ++cyc; else break;
                }
                uint32_t bits;
                memcpy(&bits, &val[j], 4);
                pe[p][start[p] + cyc] = ((uint64_t)((c - c0) & 0x3FFF) << 50) |
                                        ((uint64_t)((r / NPE) & 0x3FFFF) << 32) | bits;
                last[r] = (int)cyc;
            }
        size_t longest = 0;
        for (int p = 0; p < NPE; ++p) { len[p] = start[p] + used[p]; if (len[p] > longest) longest = len[p]; }
        for (int p = 0; p < NPE; ++p) {
            if (longest > cap[p]) {
                cap[p] = longest + 64;
                pe[p] = (uint64_t *)realloc(pe[p], cap[p] * sizeof(uint64_t));
                if (!pe[p]) return ORC_ERR_ALLOC;
            }
            for (size_t t = len[p]; t < longest; ++t) pe[p][t] = bubble;
            len[p] = longest;
        }
        ptr[w + 1] = (int)longest;
    }
    const long L = nwin ? ptr[nwin] : 0;
    const long clen = (8 * L + 511) / 512 * 512;
    uint64_t *ch = (uint64_t *)calloc((size_t)(clen > 0 ? clen * 8 : 1), sizeof(uint64_t));
    if (!ch) return ORC_ERR_ALLOC;
    for (int p = 0; p < NPE; ++p) {
        const int g = p / 8, slot = ((g & 1) ? 4 : 0) + ((g & 2) ? 2 : 0) + ((g & 4) ? 1 : 0);
        for (long i = 0; i < L; ++i) ch[(long)(p % 8) * clen + i * 8 + slot] = pe[p][i];
        free(pe[p]);
    }
    free(last);
    *ptr_out = ptr; *chan_out = ch; *chan_len = clen;
    return ORC_OK;
}

/* ---- Blocked-ELL bf16 SpMM (BASELINE config 5).  NO reference analogue: parity for this path is
 * UNPINNED by the reference; this is our own fp32 CPU restatement on bf16 inputs:
 * every bf16 x bf16 product is exact in fp32; products are accumulated in fp32 in slot order, k
 * ascending, then C = alpha*acc + beta*C as in cpu_spmm_CSR (sparse_helper.h:286-288).  The GPU's
 * MFMA sums the 16 products of a k-step in hardware order, so tests compare within a stated
 * tolerance relative to sum|a*b|, and also against a float64 evaluation. */
static float orc_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void orc_bell_spmm(int M, int K, int N, int ell_width, const int *block_col, const uint16_t *block_val,
                   const uint16_t *B /* bf16 column-major, ld = K */, float alpha, float beta,
                   float *C /* fp32 column-major, ld = M */, double *abs_sum /* optional M*N: sum|a*b| */) {
    const int mb = M / 32;
    for (int br = 0; br < mb; ++br) {
        for (int i = 0; i < 32; ++i) {
            const int m = br * 32 + i;
            for (int n = 0; n < N; ++n) {
                float acc = 0.0f;
                double asum = 0.0;
                for (int s = 0; s < ell_width; ++s) {
                    const int bc = block_col[(size_t)br * ell_width + s];
                    if (bc < 0) continue;
                    const uint16_t *a = block_val + ((size_t)br * ell_width + s) * 1024 + (size_t)i * 32;
                    const uint16_t *b = B + (size_t)n * K + (size_t)bc * 32;
                    for (int k = 0; k < 32; ++k) {
                        const float p = orc_bf16_to_f32(a[k]) * orc_bf16_to_f32(b[k]);
                        acc += p;
                        asum += fabs((double)p);
                    }
                }
                const size_t o = (size_t)m + (size_t)M * n;
                if (abs_sum) abs_sum[o] = asum;
                C[o] = alpha * acc + beta * C[o];
            }
        }
    }
    (void)K;
}

/*
 * snb_oracle.c -- CPU ORACLE (test infrastructure only; see snb_oracle.h for the rules).
 *
 * Plain-C restatement of the Reference-platform arithmetic of craabreu/openmm-nonbonded-slicing.
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Build: gcc -O2 -fopenmp -shared -fPIC snb_oracle.c -o libsnb_oracle.so -lm   (oracle/Makefile)
 * The reference's code is single-threaded; here the pair-list build, the pair loops and the PME passes run on OpenMP threads
 * (round 4: the full-size parity cases must fit the driver's time limit).  Only the order of summation differs between thread counts.
 */
#include "snb_oracle.h"
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* OpenMM 8.3 SimTKOpenMMRealType.h constants (third-party, absent from /root/reference; values
 * asserted here and recorded in DESIGN.md). */
#define ORC_ONE_4PI_EPS0 138.93545764438198
#define ORC_PI 3.14159265358979323846
#define ORC_EPSILON0 (1.0 / (4.0 * ORC_PI * ORC_ONE_4PI_EPS0))

#define COUL 0
#define VDW 1

typedef double complex cplx;

static long long g_last_pairs = 0;
long long orc_last_pair_count(void) { return g_last_pairs; }

/* SlicedNonbondedForce.h:22 */
static inline int slice_index(int i, int j) { return i > j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

/* ------------------------------------------------------------------------------------------------
 * Displacements.  OpenMM ReferenceForce::getDeltaR / getDeltaRPeriodic (third-party, a13):
 * delta = xj_arg2 - xi_arg1, triclinic wrap subtracts box[2], box[1], box[0] multiples in turn.
 * The reference calls them as getDeltaR*(x[jj], x[ii], ...) so delta = x[ii] - x[jj]
 * (ReferenceSlicedLJCoulombIxn.cpp:376,462,585).
 * ---------------------------------------------------------------------------------------------- */
static inline void delta_plain(const double* xjj, const double* xii, double* d) {
    d[0] = xii[0] - xjj[0]; d[1] = xii[1] - xjj[1]; d[2] = xii[2] - xjj[2];
}
static inline void delta_periodic(const double* xjj, const double* xii, const double* box, double* d) {
    d[0] = xii[0] - xjj[0]; d[1] = xii[1] - xjj[1]; d[2] = xii[2] - xjj[2];
    double s2 = floor(d[2] / box[8] + 0.5);
    d[0] -= s2 * box[6]; d[1] -= s2 * box[7]; d[2] -= s2 * box[8];
    double s1 = floor(d[1] / box[4] + 0.5);
    d[0] -= s1 * box[3]; d[1] -= s1 * box[4];
    double s0 = floor(d[0] / box[0] + 0.5);
    d[0] -= s0 * box[0];
}

/* ------------------------------------------------------------------------------------------------
 * Exclusion sets: sorted CSR (the reference uses vector<set<int>>,
 * ReferenceNonbondedSlicingKernels.cpp:99-112 -- every exception is an exclusion, Quirk Q6).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int* start; int* list; } excl_t;

static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

static void build_exclusions(int n, int n_exc, const int* pairs, excl_t* ex) {
    ex->start = (int*)calloc((size_t)n + 1, sizeof(int));
    for (int k = 0; k < n_exc; k++) { ex->start[pairs[2 * k] + 1]++; ex->start[pairs[2 * k + 1] + 1]++; }
    for (int i = 0; i < n; i++) ex->start[i + 1] += ex->start[i];
    ex->list = (int*)malloc(sizeof(int) * (size_t)(ex->start[n] > 0 ? ex->start[n] : 1));
    int* fill = (int*)calloc((size_t)n, sizeof(int));
    for (int k = 0; k < n_exc; k++) {
        int a = pairs[2 * k], b = pairs[2 * k + 1];
        ex->list[ex->start[a] + fill[a]++] = b;
        ex->list[ex->start[b] + fill[b]++] = a;
    }
    /* sort + unique each row (std::set semantics) */
    for (int i = 0; i < n; i++) {
        int len = fill[i];
        int* row = ex->list + ex->start[i];
        qsort(row, (size_t)len, sizeof(int), cmp_int);
        int m = 0;
        for (int k = 0; k < len; k++) if (k == 0 || row[k] != row[k - 1]) row[m++] = row[k];
        for (int k = m; k < len; k++) row[k] = -1; /* tombstones; never match */
    }
    free(fill);
}
static inline int is_excluded(const excl_t* ex, int i, int j) {
    for (int k = ex->start[i]; k < ex->start[i + 1]; k++) if (ex->list[k] == j) return 1;
    return 0;
}
static void free_exclusions(excl_t* ex) { free(ex->start); free(ex->list); }

/* ------------------------------------------------------------------------------------------------
 * Neighbour list: all non-excluded pairs (i<j) with minimum-image distance < cutoff.
 * Stands in for OpenMM's computeNeighborListVoxelHash (third-party; only the summation order of the
 * pair loop depends on it).  Cell grid for rectangular periodic boxes, brute force otherwise.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pairlist_s { int* ij; long long n, cap; struct pairlist_s* parts; int nparts; } pairlist_t;      /* large lists stay in the chunks they were built in (parts; ij == NULL) */
static void free_pairlist(pairlist_t* pl) { free(pl->ij); for (int k = 0; k < pl->nparts; k++) free(pl->parts[k].ij); free(pl->parts); pl->ij = NULL; pl->parts = NULL; pl->nparts = 0; pl->n = 0; }

static void pl_push(pairlist_t* pl, int i, int j) {
    if (pl->n == pl->cap) { pl->cap = pl->cap ? pl->cap * 2 : 1 << 12; pl->ij = (int*)realloc(pl->ij, sizeof(int) * 2 * (size_t)pl->cap); }
    pl->ij[2 * pl->n] = i; pl->ij[2 * pl->n + 1] = j; pl->n++;
}

/* pairs with r2lo <= r^2 < r2hi (r2lo < 0: everything below r2hi); `cutoff` bounds the cell size and must be >= sqrt(r2hi).
 * The rows i are cut into chunks that OpenMP threads take in any order; every chunk fills a list of its own and the lists are
 * concatenated in chunk order, so the result is the list the serial double loop would produce, pair for pair. */
#define PL_CHUNK 64
static void build_pairlist_band(int n, const double* pos, const double* box, int periodic, double cutoff, double r2lo, double rc2, const excl_t* ex, pairlist_t* pl) {
    pl->ij = NULL; pl->n = 0; pl->cap = 0; pl->parts = NULL; pl->nparts = 0;
    int rect = periodic && box[3] == 0 && box[6] == 0 && box[7] == 0;
    int nc[3] = {0, 0, 0};
    if (rect) for (int d = 0; d < 3; d++) { nc[d] = (int)floor(box[4 * d] / cutoff); }
    const int brute = !rect || n < 2000 || nc[0] < 3 || nc[1] < 3 || nc[2] < 3;
    int *cstart = NULL, *order = NULL, *cell = NULL; double* spos = NULL;
    if (!brute) {      /* cell grid: atoms counting-sorted by cell (ascending index inside a cell), positions gathered in that order */
        long long ncell = (long long)nc[0] * nc[1] * nc[2];
        cstart = (int*)calloc((size_t)ncell + 1, sizeof(int));
        order = (int*)malloc(sizeof(int) * (size_t)n);
        cell = (int*)malloc(sizeof(int) * (size_t)n);
        spos = (double*)malloc(sizeof(double) * 3 * (size_t)n);
        for (int i = 0; i < n; i++) {
            int c[3];
            for (int d = 0; d < 3; d++) {
                double f = pos[3 * i + d] / box[4 * d]; f -= floor(f);
                c[d] = (int)(f * nc[d]); if (c[d] >= nc[d]) c[d] = nc[d] - 1;
            }
            cell[i] = (c[0] * nc[1] + c[1]) * nc[2] + c[2];
            cstart[cell[i] + 1]++;
        }
        for (long long c = 0; c < ncell; c++) cstart[c + 1] += cstart[c];
        int* fill = (int*)malloc(sizeof(int) * (size_t)ncell);
        memcpy(fill, cstart, sizeof(int) * (size_t)ncell);
        for (int i = 0; i < n; i++) { int k = fill[cell[i]]++; order[k] = i; spos[3 * k] = pos[3 * i]; spos[3 * k + 1] = pos[3 * i + 1]; spos[3 * k + 2] = pos[3 * i + 2]; }
        free(fill);
    }
    const int nchunk = (n + PL_CHUNK - 1) / PL_CHUNK;
    const double tp0 = omp_get_wtime();
    pairlist_t* part = (pairlist_t*)calloc((size_t)(nchunk > 0 ? nchunk : 1), sizeof(pairlist_t));
#pragma omp parallel for schedule(dynamic, 4)
    for (int ch = 0; ch < nchunk; ch++) {
        pairlist_t* my = &part[ch];
        const int i1 = (ch + 1) * PL_CHUNK < n ? (ch + 1) * PL_CHUNK : n;
        for (int i = ch * PL_CHUNK; i < i1; i++) {
            if (brute) {
                for (int j = i + 1; j < n; j++) {
                    double d[3];
                    if (periodic) delta_periodic(pos + 3 * j, pos + 3 * i, box, d); else delta_plain(pos + 3 * j, pos + 3 * i, d);
                    double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                    if (r2 < rc2 && r2 >= r2lo && !is_excluded(ex, i, j)) pl_push(my, i, j);
                }
                continue;
            }
            int ci = cell[i];
            int cx = ci / (nc[1] * nc[2]), cy = (ci / nc[2]) % nc[1], cz = ci % nc[2];
            for (int dx = -1; dx <= 1; dx++) for (int dy = -1; dy <= 1; dy++) for (int dz = -1; dz <= 1; dz++) {
                int c2 = (((cx + dx + nc[0]) % nc[0]) * nc[1] + (cy + dy + nc[1]) % nc[1]) * nc[2] + (cz + dz + nc[2]) % nc[2];
                for (int k = cstart[c2]; k < cstart[c2 + 1]; k++) {
                    const int j = order[k];
                    if (j <= i) continue;
                    double d[3];
                    delta_periodic(spos + 3 * k, pos + 3 * i, box, d);
                    double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                    if (r2 < rc2 && r2 >= r2lo && !is_excluded(ex, i, j)) pl_push(my, i, j);
                }
            }
        }
    }
    const double tp1 = omp_get_wtime();
    long long total = 0;
    for (int ch = 0; ch < nchunk; ch++) total += part[ch].n;
    if (total > (4LL << 20)) {      /* (copying half a gigabyte of pairs into one array costs more than finding them) */
        pl->parts = part; pl->nparts = nchunk; pl->n = total; part = NULL;
    } else {
        pl->cap = total > 0 ? total : 1; pl->ij = (int*)malloc(sizeof(int) * 2 * (size_t)pl->cap);
        for (int ch = 0; ch < nchunk; ch++) {
            if (part[ch].n) memcpy(pl->ij + 2 * pl->n, part[ch].ij, sizeof(int) * 2 * (size_t)part[ch].n);
            pl->n += part[ch].n; free(part[ch].ij);
        }
    }
    free(part); free(cstart); free(order); free(cell); free(spos);
    if (getenv("ORC_TRACE")) fprintf(stderr, "[orc] pair list: search %.2f s on %d threads, concatenation %.2f s\n", tp1 - tp0, omp_get_max_threads(), omp_get_wtime() - tp1);
}
static void build_pairlist(int n, const double* pos, const double* box, int periodic, double cutoff, const excl_t* ex, pairlist_t* pl) {
    build_pairlist_band(n, pos, box, periodic, cutoff, -1.0, cutoff * cutoff, ex, pl);
}

/* ------------------------------------------------------------------------------------------------
 * FFT (stands in for the vendored pocketfft used at ReferencePME.cpp:793-805): unnormalised
 * complex mixed-radix Stockham, any length (O(n * sum of prime factors)).
 * ---------------------------------------------------------------------------------------------- */
static void fft1d(int n, cplx* x, cplx* y, const cplx* w /* w[k] = exp(sign*2*pi*i*k/n) */) {
    /* decimation-in-frequency Stockham autosort */
    int len = n, s = 1;
    cplx* a = x; cplx* b = y;
    while (len > 1) {
        int p = 2;
        while (len % p) p++;
        int m = len / p;
        for (int q = 0; q < m; q++)
            for (int t = 0; t < s; t++) {
                cplx in[64];
                cplx* inp = in;
                cplx* heap = NULL;
                if (p > 64) inp = heap = (cplx*)malloc(sizeof(cplx) * (size_t)p);
                for (int j = 0; j < p; j++) inp[j] = a[t + s * (q + m * j)];
                for (int k = 0; k < p; k++) {
                    cplx acc = 0;
                    for (int j = 0; j < p; j++) acc += inp[j] * w[(int)(((long long)j * k * (n / p)) % n)];
                    b[t + s * (p * q + k)] = acc * w[(int)(((long long)q * k * s) % n)];
                }
                if (heap) free(heap);
            }
        cplx* tmp = a; a = b; b = tmp;
        len = m; s *= p;
    }
    if (a != x) memcpy(x, a, sizeof(cplx) * (size_t)n);
}

static void fft_axis(cplx* data, int nx, int ny, int nz, int axis, int sign) {
    int n = axis == 0 ? nx : axis == 1 ? ny : nz;
    cplx* w = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    for (int k = 0; k < n; k++) w[k] = cexp(sign * 2.0 * ORC_PI * I * (double)k / (double)n);
    long long nlines = (long long)nx * ny * nz / n;
#pragma omp parallel
    {
        cplx* buf = (cplx*)malloc(sizeof(cplx) * 2 * (size_t)n);
#pragma omp for schedule(static)
        for (long long l = 0; l < nlines; l++) {
            long long base, stride;
            if (axis == 2) { base = l * nz; stride = 1; }
            else if (axis == 1) { long long ix = l / nz, iz = l % nz; base = ix * ny * nz + iz; stride = nz; }
            else { base = l; stride = (long long)ny * nz; }
            for (int k = 0; k < n; k++) buf[k] = data[base + k * stride];
            fft1d(n, buf, buf + n, w);
            for (int k = 0; k < n; k++) data[base + k * stride] = buf[k];
        }
        free(buf);
    }
    free(w);
}

void orc_fft3d(double* data, int nx, int ny, int nz, int sign) {
    cplx* d = (cplx*)data;
    fft_axis(d, nx, ny, nz, 2, sign);
    fft_axis(d, nx, ny, nz, 1, sign);
    fft_axis(d, nx, ny, nz, 0, sign);
}

/* ------------------------------------------------------------------------------------------------
 * PME (ReferencePME.cpp).
 * ---------------------------------------------------------------------------------------------- */

/* ReferencePME.cpp:88-183 */
void orc_bspline_moduli(int n, int order, double* out) {
    double* data = (double*)calloc((size_t)order, sizeof(double));
    double* bsp = (double*)calloc((size_t)(n > order + 1 ? n : order + 1), sizeof(double));
    data[order - 1] = 0; data[1] = 0; data[0] = 1;
    for (int k = 3; k < order; k++) {
        double div = 1.0 / (k - 1.0);
        data[k - 1] = 0;
        for (int l = 1; l < (k - 1); l++) data[k - l - 1] = div * (l * data[k - l - 2] + (k - l) * data[k - l - 1]);
        data[0] = div * data[0];
    }
    double div = 1.0 / (order - 1);
    data[order - 1] = 0;
    for (int l = 1; l < (order - 1); l++) data[order - l - 1] = div * (l * data[order - l - 2] + (order - l) * data[order - l - 1]);
    data[0] = div * data[0];
    for (int i = 1; i <= order; i++) bsp[i] = data[i - 1];
    for (int i = 0; i < n; i++) {
        double sc = 0, ss = 0;
        for (int j = 0; j < n && j <= order; j++) { /* bsp[j] = 0 beyond order */
            double arg = (2.0 * ORC_PI * i * j) / n;
            sc += bsp[j] * cos(arg); ss += bsp[j] * sin(arg);
        }
        out[i] = sc * sc + ss * ss;
    }
    double* tmp = (double*)malloc(sizeof(double) * (size_t)n);
    memcpy(tmp, out, sizeof(double) * (size_t)n);
    /* the reference patches in place, left to right (ReferencePME.cpp:170-176) */
    for (int i = 0; i < n; i++)
        if (out[i] < 1.0e-7) out[i] = (out[(i - 1 + n) % n] + out[(i + 1) % n]) / 2;
    free(tmp); free(data); free(bsp);
}

typedef struct {
    int natoms, nsub, order; int ngrid[3]; double alpha;
    cplx* grid; double* moduli[3]; double* theta[3]; double* dtheta[3]; int* pidx; double* pfrac;
} pme_t;

/* ReferencePME.cpp:186-194 */
static void invert_box(const double* box, double* r /* r[3][3] */) {
    double det = box[0] * box[4] * box[8];
    double sc = 1.0 / det;
    r[0] = box[4] * box[8] * sc; r[1] = 0; r[2] = 0;
    r[3] = -box[3] * box[8] * sc; r[4] = box[0] * box[8] * sc; r[5] = 0;
    r[6] = (box[3] * box[7] - box[4] * box[6]) * sc; r[7] = -box[0] * box[7] * sc; r[8] = box[0] * box[4] * sc;
}

/* ReferencePME.cpp:708-748 */
static void pme_init(pme_t* p, double alpha, int natoms, int nsub, const int ngrid[3], int order) {
    p->natoms = natoms; p->nsub = nsub; p->order = order; p->alpha = alpha;
    for (int d = 0; d < 3; d++) {
        p->ngrid[d] = ngrid[d];
        p->theta[d] = (double*)malloc(sizeof(double) * (size_t)order * natoms);
        p->dtheta[d] = (double*)malloc(sizeof(double) * (size_t)order * natoms);
        p->moduli[d] = (double*)malloc(sizeof(double) * (size_t)ngrid[d]);
        orc_bspline_moduli(ngrid[d], order, p->moduli[d]);
    }
    p->pidx = (int*)malloc(sizeof(int) * 3 * (size_t)natoms);
    p->pfrac = (double*)malloc(sizeof(double) * 3 * (size_t)natoms);
    p->grid = (cplx*)malloc(sizeof(cplx) * (size_t)ngrid[0] * ngrid[1] * ngrid[2] * nsub);
}
static void pme_destroy(pme_t* p) {
    for (int d = 0; d < 3; d++) { free(p->theta[d]); free(p->dtheta[d]); free(p->moduli[d]); }
    free(p->pidx); free(p->pfrac); free(p->grid);
}

/* ReferencePME.cpp:196-256 and 264-317 */
static void pme_index_and_splines(pme_t* p, const double* pos, const double* recip) {
    int order = p->order;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->natoms; i++) {
        for (int d = 0; d < 3; d++) {
            double t = pos[3 * i] * recip[0 * 3 + d] + pos[3 * i + 1] * recip[1 * 3 + d] + pos[3 * i + 2] * recip[2 * 3 + d];
            t = (t - floor(t)) * p->ngrid[d];
            int ti = (int)t;
            p->pfrac[3 * i + d] = t - ti;
            p->pidx[3 * i + d] = ti % p->ngrid[d];
        }
        for (int j = 0; j < 3; j++) {
            double dr = p->pfrac[3 * i + j];
            double* data = p->theta[j] + (size_t)i * order;
            double* ddata = p->dtheta[j] + (size_t)i * order;
            data[order - 1] = 0; data[1] = dr; data[0] = 1 - dr;
            for (int k = 3; k < order; k++) {
                double div = 1.0 / (k - 1.0);
                data[k - 1] = div * dr * data[k - 2];
                for (int l = 1; l < (k - 1); l++) data[k - l - 1] = div * ((dr + l) * data[k - l - 2] + (k - l - dr) * data[k - l - 1]);
                data[0] = div * (1 - dr) * data[0];
            }
            ddata[0] = -data[0];
            for (int k = 1; k < order; k++) ddata[k] = data[k - 1] - data[k];
            double div = 1.0 / (order - 1);
            data[order - 1] = div * dr * data[order - 2];
            for (int l = 1; l < (order - 1); l++) data[order - l - 1] = div * ((dr + l) * data[order - l - 2] + (order - l - dr) * data[order - l - 1]);
            data[0] = div * (1 - dr) * data[0];
        }
    }
}

/* ReferencePME.cpp:320-396 (serial, atom order, like the reference) */
static void pme_spread(pme_t* p, const double* charges, const int* subsets) {
    int order = p->order, nx = p->ngrid[0], ny = p->ngrid[1], nz = p->ngrid[2];
    size_t total = (size_t)nx * ny * nz * p->nsub;
    for (size_t i = 0; i < total; i++) p->grid[i] = 0;
    for (int i = 0; i < p->natoms; i++) {
        double q = charges[i]; int subset = subsets[i];
        int x0 = p->pidx[3 * i], y0 = p->pidx[3 * i + 1], z0 = p->pidx[3 * i + 2];
        const double* tx = p->theta[0] + (size_t)i * order; const double* ty = p->theta[1] + (size_t)i * order; const double* tz = p->theta[2] + (size_t)i * order;
        for (int ix = 0; ix < order; ix++) {
            int xi = (x0 + ix) % nx;
            for (int iy = 0; iy < order; iy++) {
                int yi = (y0 + iy) % ny;
                for (int iz = 0; iz < order; iz++) {
                    int zi = (z0 + iz) % nz;
                    size_t index = (((size_t)subset * nx + xi) * ny + yi) * nz + zi;
                    p->grid[index] += q * tx[ix] * ty[iy] * tz[iz];
                }
            }
        }
    }
}

/* ReferencePME.cpp:400-496 (term=COUL) and 499-595 (term=VDW, dispersion) */
static void pme_convolution(pme_t* p, const double* box, const double* recip, double* sliceE /* [S][2] */, int term) {
    int nx = p->ngrid[0], ny = p->ngrid[1], nz = p->ngrid[2], ns = p->nsub;
    int S = ns * (ns + 1) / 2;
    double volume = box[0] * box[4] * box[8];
    double factor = ORC_PI * ORC_PI / (p->alpha * p->alpha);
    double boxfactorC = ORC_PI * volume;
    double boxfactorD = -2 * ORC_PI * sqrt(ORC_PI) / (6.0 * volume);
    int maxkx = (nx + 1) / 2, maxky = (ny + 1) / 2, maxkz = (nz + 1) / 2;
    double bfac = ORC_PI / p->alpha;
    double fac1 = 2.0 * ORC_PI * ORC_PI * ORC_PI * sqrt(ORC_PI);
    double fac2 = p->alpha * p->alpha * p->alpha;
    double fac3 = -2.0 * p->alpha * ORC_PI * ORC_PI;
    double* Eacc = (double*)calloc((size_t)S, sizeof(double));
#pragma omp parallel
    {
        double* El = (double*)calloc((size_t)S, sizeof(double));
#pragma omp for schedule(static)
        for (int kx = 0; kx < nx; kx++) {
            double mx = (kx < maxkx) ? kx : (kx - nx);
            double mhx = mx * recip[0];
            double bx = (term == COUL ? boxfactorC : 1.0) * p->moduli[0][kx];
            for (int ky = 0; ky < ny; ky++) {
                double my = (ky < maxky) ? ky : (ky - ny);
                double mhy = mx * recip[3] + my * recip[4];
                double by = p->moduli[1][ky];
                for (int kz = 0; kz < nz; kz++) {
                    if (term == COUL && kx == 0 && ky == 0 && kz == 0) continue;
                    double mz = (kz < maxkz) ? kz : (kz - nz);
                    double mhz = mx * recip[6] + my * recip[7] + mz * recip[8];
                    double m2 = mhx * mhx + mhy * mhy + mhz * mhz;
                    double bz = p->moduli[2][kz];
                    double eterm;
                    if (term == COUL) {
                        double denom = m2 * bx * by * bz;
                        eterm = ORC_ONE_4PI_EPS0 * exp(-factor * m2) / denom;
                    } else {
                        double denom = boxfactorD / (bx * by * bz);
                        double m = sqrt(m2), m3 = m * m2, b = bfac * m;
                        eterm = (fac1 * erfc(b) * m3 + exp(-b * b) * (fac2 + fac3 * m2)) * denom;
                    }
                    for (int j = 0; j < ns; j++) {
                        cplx* ptr = p->grid + (((size_t)j * nx + kx) * ny + ky) * nz + kz;
                        double d1 = creal(*ptr), d2 = cimag(*ptr);
                        *ptr = d1 * eterm + d2 * eterm * I;
                        El[j * (j + 3) / 2] += 0.5 * eterm * (d1 * d1 + d2 * d2);
                        for (int i = 0; i < j; i++) {
                            cplx* pi = p->grid + (((size_t)i * nx + kx) * ny + ky) * nz + kz; /* already convolved */
                            El[j * (j + 1) / 2 + i] += d1 * creal(*pi) + d2 * cimag(*pi);
                        }
                    }
                }
            }
        }
#pragma omp critical
        for (int s = 0; s < S; s++) Eacc[s] += El[s];
        free(El);
    }
    for (int s = 0; s < S; s++) sliceE[2 * s + term] += Eacc[s];
    free(Eacc);
}

/* ReferencePME.cpp:598-702 */
static void pme_interpolate(pme_t* p, const double* recip, const int* subsets, const double* lambdas, const double* charges,
                            double* forces, int term, int correct_q1) {
    int order = p->order, nx = p->ngrid[0], ny = p->ngrid[1], nz = p->ngrid[2];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->natoms; i++) {
        double fx = 0, fy = 0, fz = 0, q = charges[i];
        int si = subsets[i];
        int x0 = p->pidx[3 * i], y0 = p->pidx[3 * i + 1], z0 = p->pidx[3 * i + 2];
        const double* thx = p->theta[0] + (size_t)i * order; const double* thy = p->theta[1] + (size_t)i * order; const double* thz = p->theta[2] + (size_t)i * order;
        const double* dthx = p->dtheta[0] + (size_t)i * order; const double* dthy = p->dtheta[1] + (size_t)i * order; const double* dthz = p->dtheta[2] + (size_t)i * order;
        for (int ix = 0; ix < order; ix++) {
            int xi = (x0 + ix) % nx; double tx = thx[ix], dtx = dthx[ix];
            for (int iy = 0; iy < order; iy++) {
                int yi = (y0 + iy) % ny; double ty = thy[iy], dty = dthy[iy];
                for (int iz = 0; iz < order; iz++) {
                    int zi = (z0 + iz) % nz; double tz = thz[iz], dtz = dthz[iz];
                    for (int sj = 0; sj < p->nsub; sj++) {
                        /* Quirk Q1: the reference strides subsets by ngrid[2] (ReferencePME.cpp:682) */
                        size_t index = (((size_t)sj * (correct_q1 ? nx : nz) + xi) * ny + yi) * nz + zi;
                        double gv = lambdas[2 * slice_index(si, sj) + term] * creal(p->grid[index]);
                        fx += dtx * ty * tz * gv; fy += tx * dty * tz * gv; fz += tx * ty * dtz * gv;
                    }
                }
            }
        }
        forces[3 * i] -= q * (fx * nx * recip[0]);
        forces[3 * i + 1] -= q * (fx * nx * recip[3] + fy * ny * recip[4]);
        forces[3 * i + 2] -= q * (fx * nx * recip[6] + fy * ny * recip[7] + fz * nz * recip[8]);
    }
}

/* ReferencePME.cpp:754-811 (term=COUL) / 814-871 (term=VDW) */
static void pme_exec(pme_t* p, const double* pos, const int* subsets, const double* lambdas, double* forces, const double* charges,
                     const double* box, double* sliceE, int term, int correct_q1) {
    double recip[9];
    invert_box(box, recip);
    pme_index_and_splines(p, pos, recip);
    pme_spread(p, charges, subsets);
    int nx = p->ngrid[0], ny = p->ngrid[1], nz = p->ngrid[2];
    for (int s = 0; s < p->nsub; s++) orc_fft3d((double*)(p->grid + (size_t)s * nx * ny * nz), nx, ny, nz, -1);
    pme_convolution(p, box, recip, sliceE, term);
    for (int s = 0; s < p->nsub; s++) orc_fft3d((double*)(p->grid + (size_t)s * nx * ny * nz), nx, ny, nz, +1);
    pme_interpolate(p, recip, subsets, lambdas, charges, forces, term, correct_q1);
}

/* ------------------------------------------------------------------------------------------------
 * Dispersion correction coefficients (SlicedNonbondedForceImpl.cpp:150-185, 263-354).
 * NB: the reference forms class-pair counts in `int` (overflow for classes > ~46k);
 * this restatement uses doubles and agrees wherever the reference does not overflow.
 * ---------------------------------------------------------------------------------------------- */
static double eval_integral(double r, double rs, double rc, double sigma) {
    double A = 1 / (rc - rs), A2 = A * A, A3 = A2 * A;
    double sig2 = sigma * sigma, sig6 = sig2 * sig2 * sig2;
    double rs2 = rs * rs, rs3 = rs * rs2;
    double r2 = r * r, r3 = r * r2, r4 = r * r3, r5 = r * r4, r6 = r * r5, r9 = r3 * r6;
    return sig6 * A3 * ((sig6 * (+rs3 * 28 * (6 * rs2 * A2 + 15 * rs * A + 10) - r * rs2 * 945 * (rs2 * A2 + 2 * rs * A + 1) +
                                 r2 * rs * 1080 * (2 * rs2 * A2 + 3 * rs * A + 1) - r3 * 420 * (6 * rs2 * A2 + 6 * rs * A + 1) +
                                 r4 * 756 * (2 * rs * A2 + A) - r5 * 378 * A2) -
                         r6 * (+rs3 * 84 * (6 * rs2 * A2 + 15 * rs * A + 10) - r * rs2 * 3780 * (rs2 * A2 + 2 * rs * A + 1) +
                               r2 * rs * 7560 * (2 * rs2 * A2 + 3 * rs * A + 1))) /
                            (252 * r9) -
                        log(r) * 10 * (6 * rs2 * A2 + 6 * rs * A + 1) + r * 15 * (2 * rs * A2 + A) - r2 * 3 * A2);
}

typedef struct { double sigma, eps; int subset; double count; } pclass_t;
static int cmp_class(const void* a, const void* b) {
    const pclass_t* x = (const pclass_t*)a; const pclass_t* y = (const pclass_t*)b;
    if (x->sigma != y->sigma) return x->sigma < y->sigma ? -1 : 1;
    if (x->eps != y->eps) return x->eps < y->eps ? -1 : 1;
    return (x->subset > y->subset) - (x->subset < y->subset);
}

int orc_dispersion_coefficients(int n, int nsub, const double* sigma, const double* epsilon, const int* subset,
                                double cutoff, int use_switch, double switch_distance, double* out) {
    int S = nsub * (nsub + 1) / 2;
    pclass_t* all = (pclass_t*)malloc(sizeof(pclass_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { all[i].sigma = sigma[i]; all[i].eps = epsilon[i]; all[i].subset = subset[i]; all[i].count = 1; }
    qsort(all, (size_t)n, sizeof(pclass_t), cmp_class);
    int nc = 0;
    for (int i = 0; i < n; i++) {
        if (nc > 0 && cmp_class(&all[nc - 1], &all[i]) == 0) all[nc - 1].count += 1;
        else all[nc++] = all[i];
    }
    double* sum1 = (double*)calloc((size_t)S, sizeof(double));
    double* sum2 = (double*)calloc((size_t)S, sizeof(double));
    double* sum3 = (double*)calloc((size_t)S, sizeof(double));
    for (int a = 0; a < nc; a++) {
        double sg = all[a].sigma, ep = all[a].eps; int s = all[a].subset;
        double count = all[a].count * (all[a].count + 1) / 2;
        double s2 = sg * sg, s6 = s2 * s2 * s2;
        int slice = s * (s + 3) / 2;
        sum1[slice] += count * ep * s6 * s6; sum2[slice] += count * ep * s6;
        if (use_switch) sum3[slice] += count * ep * (eval_integral(cutoff, switch_distance, cutoff, sg) - eval_integral(switch_distance, switch_distance, cutoff, sg));
    }
    for (int a = 0; a < nc; a++)
        for (int b = 0; b < a; b++) {
            double sg = 0.5 * (all[a].sigma + all[b].sigma), ep = sqrt(all[a].eps * all[b].eps);
            int slice = slice_index(all[a].subset, all[b].subset);
            double count = all[a].count * all[b].count;
            double s2 = sg * sg, s6 = s2 * s2 * s2;
            sum1[slice] += count * ep * s6 * s6; sum2[slice] += count * ep * s6;
            if (use_switch) sum3[slice] += count * ep * (eval_integral(cutoff, switch_distance, cutoff, sg) - eval_integral(switch_distance, switch_distance, cutoff, sg));
        }
    double N = (double)n;
    double numInteractions = (N * (N + 1)) / 2;
    for (int s = 0; s < S; s++)
        out[s] = 8 * N * N * ORC_PI * ((sum1[s] / numInteractions) / (9 * pow(cutoff, 9)) - (sum2[s] / numInteractions) / (3 * pow(cutoff, 3)) + sum3[s] / numInteractions);
    free(all); free(sum1); free(sum2); free(sum3);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Pair interactions.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_config* cfg; const double* pos; const double* box; const int* subset;
    const double* sig; const double* eps; const double* q; /* sig = sigma/2, eps = 2 sqrt(epsilon) */
    const double* lam; double krf, crf; int periodic;
} ctx_t;

/* ReferenceSlicedLJCoulombIxn.cpp:571-631 */
static void one_pair(const ctx_t* c, int ii, int jj, double* fi, double* fj, double* sliceE) {
    const orc_config* cfg = c->cfg;
    int slice = slice_index(c->subset[ii], c->subset[jj]);
    double d[3];
    if (c->periodic) delta_periodic(c->pos + 3 * jj, c->pos + 3 * ii, c->box, d); else delta_plain(c->pos + 3 * jj, c->pos + 3 * ii, d);
    double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double r = sqrt(r2), inverseR = 1.0 / r;
    double switchValue = 1, switchDeriv = 0;
    int cutoff = cfg->method != ORC_NoCutoff;
    if (cfg->use_switch && r > cfg->switch_distance) {
        double t = (r - cfg->switch_distance) / (cfg->cutoff - cfg->switch_distance);
        switchValue = 1 + t * t * t * (-10 + t * (15 - t * 6));
        switchDeriv = t * t * (-30 + t * (60 - t * 30)) / (cfg->cutoff - cfg->switch_distance);
    }
    double sig = c->sig[ii] + c->sig[jj];
    double sig2 = inverseR * sig; sig2 *= sig2;
    double sig6 = sig2 * sig2 * sig2;
    double eps = c->eps[ii] * c->eps[jj];
    double dEdRvdW = switchValue * eps * (12.0 * sig6 - 6.0) * sig6 * inverseR * inverseR;
    double dEdRCoul = inverseR * inverseR;
    double qq = ORC_ONE_4PI_EPS0 * c->q[ii] * c->q[jj];
    if (cutoff) dEdRCoul *= qq * (inverseR - 2.0 * c->krf * r2); else dEdRCoul *= qq * inverseR;
    double energy = eps * (sig6 - 1.0) * sig6;
    if (cfg->use_switch) { dEdRvdW -= energy * switchDeriv * inverseR; energy *= switchValue; }
    sliceE[2 * slice + VDW] += energy;
    if (cutoff) sliceE[2 * slice + COUL] += qq * (inverseR + c->krf * r2 - c->crf); else sliceE[2 * slice + COUL] += qq * inverseR;
    double factor = c->lam[2 * slice + VDW] * dEdRvdW + c->lam[2 * slice + COUL] * dEdRCoul;
    for (int k = 0; k < 3; k++) { double f = factor * d[k]; fi[k] += f; fj[k] -= f; }
}
static void one_ixn(const ctx_t* c, int ii, int jj, double* forces, double* sliceE) { one_pair(c, ii, jj, forces + 3 * ii, forces + 3 * jj, sliceE); }

/* one real-space Ewald / PME / LJPME pair (ReferenceSlicedLJCoulombIxn.cpp:367-445) */
static void ewald_pair(const ctx_t* c, int ii, int jj, double* fi, double* fj, double* sliceE) {
    const orc_config* cfg = c->cfg;
    const int ljpme = cfg->method == ORC_LJPME;
    const double alpha = cfg->alpha, alphaD = cfg->alpha_d, SQRT_PI = sqrt(ORC_PI);
    const double invCut2 = 1.0 / (cfg->cutoff * cfg->cutoff), invCut6 = invCut2 * invCut2 * invCut2;
    int slice = slice_index(c->subset[ii], c->subset[jj]);
    double d[3];
    delta_periodic(c->pos + 3 * jj, c->pos + 3 * ii, c->box, d);
    double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), inverseR = 1.0 / r;
    double switchValue = 1, switchDeriv = 0;
    if (cfg->use_switch && r > cfg->switch_distance) {
        double t = (r - cfg->switch_distance) / (cfg->cutoff - cfg->switch_distance);
        switchValue = 1 + t * t * t * (-10 + t * (15 - t * 6));
        switchDeriv = t * t * (-30 + t * (60 - t * 30)) / (cfg->cutoff - cfg->switch_distance);
    }
    double alphaR = alpha * r;
    double qq = ORC_ONE_4PI_EPS0 * c->q[ii] * c->q[jj];
    double dEdRCoul = qq * inverseR * inverseR * inverseR;
    dEdRCoul *= erfc(alphaR) + 2 * alphaR * exp(-alphaR * alphaR) / SQRT_PI;
    double sig = c->sig[ii] + c->sig[jj];
    double sig2 = inverseR * sig; sig2 *= sig2;
    double sig6 = sig2 * sig2 * sig2;
    double eps = c->eps[ii] * c->eps[jj];
    double dEdRvdW = switchValue * eps * (12.0 * sig6 - 6.0) * sig6 * inverseR * inverseR;
    double vdwEnergy = eps * (sig6 - 1.0) * sig6;
    if (ljpme) {
        double dalphaR = alphaD * r, dar2 = dalphaR * dalphaR, dar4 = dar2 * dar2, dar6 = dar4 * dar2;
        double inverseR2 = inverseR * inverseR;
        double c6i = 8.0 * pow(c->sig[ii], 3.0) * c->eps[ii], c6j = 8.0 * pow(c->sig[jj], 3.0) * c->eps[jj];
        double emult = c6i * c6j * inverseR2 * inverseR2 * inverseR2 * (1.0 - exp(-dar2) * (1.0 + dar2 + 0.5 * dar4));
        dEdRvdW += 6.0 * c6i * c6j * inverseR2 * inverseR2 * inverseR2 * inverseR2 * (1.0 - exp(-dar2) * (1.0 + dar2 + 0.5 * dar4 + dar6 / 6.0));
        sig2 = c->sig[ii] + c->sig[jj]; sig2 *= sig2; sig6 = sig2 * sig2 * sig2;
        double potentialshift = eps * (1.0 - sig6 * invCut6) * sig6 * invCut6;
        dalphaR = alphaD * cfg->cutoff; dar2 = dalphaR * dalphaR; dar4 = dar2 * dar2;
        potentialshift -= c6i * c6j * invCut6 * (1.0 - exp(-dar2) * (1.0 + dar2 + 0.5 * dar4));
        vdwEnergy += emult + potentialshift;
    }
    if (cfg->use_switch) { dEdRvdW -= vdwEnergy * switchDeriv * inverseR; vdwEnergy *= switchValue; }
    double factor = c->lam[2 * slice + VDW] * dEdRvdW + c->lam[2 * slice + COUL] * dEdRCoul;
    for (int k = 0; k < 3; k++) { double f = factor * d[k]; fi[k] += f; fj[k] -= f; }
    sliceE[2 * slice + VDW] += vdwEnergy;
    sliceE[2 * slice + COUL] += qq * inverseR * erfc(alphaR);
}

/* The pair loops below visit a list of independent pairs; only the ORDER in which their contributions are added to the force and
 * slice-energy sums depends on how the list is walked (the reference walks OpenMM's neighbour list, a third-party order in any case).
 * They are therefore split over OpenMP threads, each with force and energy sums of its own, added up afterwards: the full-size parity
 * cases spend their time here (c5: 1.4e8 pairs), and the arithmetic of every pair is untouched. */
typedef void (*pair_fn)(const ctx_t* c, int ii, int jj, double* fi, double* fj, double* sliceE);
static void pairs_threaded(const ctx_t* c, const pairlist_t* pl, pair_fn fn, double* forces, double* sliceE) {
    const int n = c->cfg->n_atoms, ns = c->cfg->n_subsets, S2 = ns * (ns + 1);
    if (pl->ij && (pl->n < 200000 || omp_get_max_threads() == 1)) {
        for (long long p = 0; p < pl->n; p++) { int ii = pl->ij[2 * p], jj = pl->ij[2 * p + 1]; fn(c, ii, jj, forces + 3 * ii, forces + 3 * jj, sliceE); }
        return;
    }
    const int nt = omp_get_max_threads();
    double* f = (double*)calloc((size_t)nt * 3 * (size_t)n, sizeof(double));
    double* e = (double*)calloc((size_t)nt * (size_t)S2, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        double* mf = f + (size_t)t * 3 * (size_t)n; double* me = e + (size_t)t * (size_t)S2;
        if (pl->ij) {
#pragma omp for schedule(static)
            for (long long p = 0; p < pl->n; p++) { int ii = pl->ij[2 * p], jj = pl->ij[2 * p + 1]; fn(c, ii, jj, mf + 3 * ii, mf + 3 * jj, me); }
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int k = 0; k < pl->nparts; k++) {
                const int* ij = pl->parts[k].ij;
                for (long long p = 0; p < pl->parts[k].n; p++) { int ii = ij[2 * p], jj = ij[2 * p + 1]; fn(c, ii, jj, mf + 3 * ii, mf + 3 * jj, me); }
            }
        }
#pragma omp for schedule(static)
        for (long long k = 0; k < 3LL * n; k++) { double a = 0; for (int u = 0; u < nt; u++) a += f[(size_t)u * 3 * (size_t)n + (size_t)k]; forces[k] += a; }
    }
    for (int k = 0; k < S2; k++) { double a = 0; for (int u = 0; u < nt; u++) a += e[(size_t)u * (size_t)S2 + k]; sliceE[k] += a; }
    free(f); free(e);
}

/* ReferenceSlicedLJCoulombIxn.cpp:179-507 */
static void ewald_ixn(const ctx_t* c, const excl_t* ex, const pairlist_t* pl, double* forces, double* sliceE) {
    const orc_config* cfg = c->cfg;
    int n = cfg->n_atoms, ns = cfg->n_subsets;
    int pme = cfg->method == ORC_PME || cfg->method == ORC_LJPME, ljpme = cfg->method == ORC_LJPME, ewald = cfg->method == ORC_Ewald;
    double alpha = cfg->alpha, alphaD = cfg->alpha_d;
    double factorEwald = -1 / (4 * alpha * alpha);
    double SQRT_PI = sqrt(ORC_PI), TWO_PI = 2.0 * ORC_PI;
    double volume = c->box[0] * c->box[4] * c->box[8];
    double recipCoeff = ORC_ONE_4PI_EPS0 * 4 * ORC_PI / volume;

    /* self energy + neutralising background (:203-222) */
    if (cfg->include_reciprocal) {
        double* subsetCharges = (double*)calloc((size_t)ns, sizeof(double));
        for (int a = 0; a < n; a++) {
            int s = c->subset[a]; double ch = c->q[a];
            subsetCharges[s] += ch;
            int slice = s * (s + 3) / 2;
            sliceE[2 * slice + COUL] -= ORC_ONE_4PI_EPS0 * ch * ch * alpha / SQRT_PI;
            if (ljpme) sliceE[2 * slice + VDW] += pow(alphaD, 6.0) * 64.0 * pow(c->sig[a], 6.0) * pow(c->eps[a], 2.0) / 12.0;
        }
        if (cfg->background_term) {
            double factor = factorEwald / (2 * ORC_EPSILON0 * volume);
            for (int i = 0; i < ns; i++)
                for (int j = i; j < ns; j++)
                    sliceE[2 * (j * (j + 1) / 2 + i) + COUL] += (i == j ? 1 : 2) * subsetCharges[i] * subsetCharges[j] * factor;
        }
        free(subsetCharges);
    }

    /* reciprocal space (:229-358) */
    if (pme && cfg->include_reciprocal) {
        pme_t p;
        double* charges = (double*)malloc(sizeof(double) * (size_t)n);
        pme_init(&p, alpha, n, ns, cfg->grid, 5);
        for (int i = 0; i < n; i++) charges[i] = c->q[i];
        pme_exec(&p, c->pos, c->subset, c->lam, forces, charges, c->box, sliceE, COUL, cfg->correct_q1);
        pme_destroy(&p);
        if (ljpme) {
            pme_init(&p, alphaD, n, ns, cfg->dgrid, 5);
            for (int i = 0; i < n; i++) charges[i] = 8.0 * pow(c->sig[i], 3.0) * c->eps[i];
            pme_exec(&p, c->pos, c->subset, c->lam, forces, charges, c->box, sliceE, VDW, cfg->correct_q1);
            pme_destroy(&p);
        }
        free(charges);
    } else if (ewald && cfg->include_reciprocal) {
        int numRx = cfg->kmax[0], numRy = cfg->kmax[1], numRz = cfg->kmax[2];
        int kmax = numRx > numRy ? (numRx > numRz ? numRx : numRz) : (numRy > numRz ? numRy : numRz);
        double rb[3] = {TWO_PI / c->box[0], TWO_PI / c->box[4], TWO_PI / c->box[8]};
        cplx* eir = (cplx*)malloc(sizeof(cplx) * (size_t)kmax * n * 3);
#define EIR(x, y, z) eir[((size_t)(x) * n + (y)) * 3 + (z)]
        cplx* tab_xy = (cplx*)malloc(sizeof(cplx) * (size_t)n);
        cplx* tab_qxyz = (cplx*)malloc(sizeof(cplx) * (size_t)n);
        double* cs = (double*)malloc(sizeof(double) * (size_t)ns);
        double* ss = (double*)malloc(sizeof(double) * (size_t)ns);
        for (int i = 0; i < n; i++) {
            for (int m = 0; m < 3; m++) EIR(0, i, m) = 1;
            if (kmax > 1) for (int m = 0; m < 3; m++) EIR(1, i, m) = cos(c->pos[3 * i + m] * rb[m]) + I * sin(c->pos[3 * i + m] * rb[m]);
            for (int j = 2; j < kmax; j++) for (int m = 0; m < 3; m++) EIR(j, i, m) = EIR(j - 1, i, m) * EIR(1, i, m);
        }
        int lowry = 0, lowrz = 1;
        for (int rx = 0; rx < numRx; rx++) {
            double kx = rx * rb[0];
            for (int ry = lowry; ry < numRy; ry++) {
                double ky = ry * rb[1];
                if (ry >= 0) for (int a = 0; a < n; a++) tab_xy[a] = EIR(rx, a, 0) * EIR(ry, a, 1);
                else for (int a = 0; a < n; a++) tab_xy[a] = EIR(rx, a, 0) * conj(EIR(-ry, a, 1));
                for (int rz = lowrz; rz < numRz; rz++) {
                    if (rz >= 0) for (int a = 0; a < n; a++) tab_qxyz[a] = c->q[a] * (tab_xy[a] * EIR(rz, a, 2));
                    else for (int a = 0; a < n; a++) tab_qxyz[a] = c->q[a] * (tab_xy[a] * conj(EIR(-rz, a, 2)));
                    for (int s = 0; s < ns; s++) cs[s] = ss[s] = 0;
                    for (int a = 0; a < n; a++) { cs[c->subset[a]] += creal(tab_qxyz[a]); ss[c->subset[a]] += cimag(tab_qxyz[a]); }
                    double kz = rz * rb[2];
                    double k2 = kx * kx + ky * ky + kz * kz;
                    double ak = exp(k2 * factorEwald) / k2;
                    for (int a = 0; a < n; a++) {
                        int i = c->subset[a];
                        for (int j = 0; j < ns; j++) {
                            int slice = slice_index(i, j);
                            double f = 2 * recipCoeff * c->lam[2 * slice + COUL] * ak * (cs[j] * cimag(tab_qxyz[a]) - ss[j] * creal(tab_qxyz[a]));
                            forces[3 * a] += f * kx; forces[3 * a + 1] += f * ky; forces[3 * a + 2] += f * kz;
                        }
                    }
                    for (int j = 0; j < ns; j++) {
                        for (int i = 0; i < j; i++) sliceE[2 * (j * (j + 1) / 2 + i) + COUL] += 2 * recipCoeff * ak * (cs[i] * cs[j] + ss[i] * ss[j]);
                        sliceE[2 * (j * (j + 3) / 2) + COUL] += recipCoeff * ak * (cs[j] * cs[j] + ss[j] * ss[j]);
                    }
                    lowrz = 1 - numRz;
                }
                lowry = 1 - numRy;
            }
        }
#undef EIR
        free(eir); free(tab_xy); free(tab_qxyz); free(cs); free(ss);
    }

    if (!cfg->include_direct) return;

    /* real space (:367-445); the list already holds exactly the non-excluded pairs within the cutoff (Q5) */
    { const double t0 = omp_get_wtime(); pairs_threaded(c, pl, ewald_pair, forces, sliceE); if (getenv("ORC_TRACE")) fprintf(stderr, "[orc] real-space pairs %.2f s\n", omp_get_wtime() - t0); }

    /* exclusion correction (:449-506) */
    double TWO_OVER_SQRT_PI = 2 / sqrt(ORC_PI);
    for (int i = 0; i < n; i++)
        for (int k = ex->start[i]; k < ex->start[i + 1]; k++) {
            int jj = ex->list[k];
            if (jj <= i) continue;
            int ii = i;
            int slice = slice_index(c->subset[ii], c->subset[jj]);
            double d[3];
            if (cfg->exceptions_periodic) delta_periodic(c->pos + 3 * jj, c->pos + 3 * ii, c->box, d); else delta_plain(c->pos + 3 * jj, c->pos + 3 * ii, d);
            double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), inverseR = 1.0 / r, alphaR = alpha * r;
            double qq = ORC_ONE_4PI_EPS0 * c->q[ii] * c->q[jj];
            if (erf(alphaR) > 1e-6) {
                double dEdR = qq * inverseR * inverseR * inverseR;
                dEdR = dEdR * (erf(alphaR) - 2 * alphaR * exp(-alphaR * alphaR) / SQRT_PI);
                double factor = c->lam[2 * slice + COUL] * dEdR;
                for (int kk = 0; kk < 3; kk++) { double f = factor * d[kk]; forces[3 * ii + kk] -= f; forces[3 * jj + kk] += f; }
                sliceE[2 * slice + COUL] -= qq * inverseR * erf(alphaR);
            } else
                sliceE[2 * slice + COUL] -= alpha * TWO_OVER_SQRT_PI * qq;
            if (ljpme) {
                double dalphaR = alphaD * r, inverseR2 = inverseR * inverseR;
                double dar2 = dalphaR * dalphaR, dar4 = dar2 * dar2, dar6 = dar4 * dar2;
                double c6i = 8.0 * pow(c->sig[ii], 3.0) * c->eps[ii], c6j = 8.0 * pow(c->sig[jj], 3.0) * c->eps[jj];
                sliceE[2 * slice + VDW] += c6i * c6j * inverseR2 * inverseR2 * inverseR2 * (1.0 - exp(-dar2) * (1.0 + dar2 + 0.5 * dar4));
                double dEdR = -6.0 * c6i * c6j * inverseR2 * inverseR2 * inverseR2 * inverseR2 * (1.0 - exp(-dar2) * (1.0 + dar2 + 0.5 * dar4 + dar6 / 6.0));
                double factor = c->lam[2 * slice + VDW] * dEdR;
                for (int kk = 0; kk < 3; kk++) { double f = factor * d[kk]; forces[3 * ii + kk] -= f; forces[3 * jj + kk] += f; }
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * Orchestration: ReferenceCalcSlicedNonbondedForceKernel::execute + computeParameters
 * (ReferenceNonbondedSlicingKernels.cpp:187-268, 339-391).  Parameter offsets are applied by the
 * caller (the host layer), so charge/sigma/epsilon here are the effective values.
 * ---------------------------------------------------------------------------------------------- */
int orc_evaluate(const orc_config* cfg, const double* pos, const double* box,
                 const double* charge, const double* sigma, const double* epsilon, const int* subset,
                 int n_exc, const int* exc_pairs, const double* exc_chargeprod, const double* exc_sigma, const double* exc_epsilon,
                 const double* lambdas, const double* disp_coef, double* forces, double* sliceE) {
    int n = cfg->n_atoms, ns = cfg->n_subsets, S = ns * (ns + 1) / 2;
    if (n < 0 || ns < 1) return -2;
    int method = cfg->method;
    int periodic = method == ORC_CutoffPeriodic, ewald = method == ORC_Ewald, pme = method == ORC_PME, ljpme = method == ORC_LJPME;
    int anyPeriodic = periodic || ewald || pme || ljpme;
    for (int s = 0; s < 2 * S; s++) sliceE[s] = 0;

    /* computeParameters (:351-368) */
    double* sig = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* eps = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) { sig[i] = 0.5 * sigma[i]; eps[i] = 2.0 * sqrt(epsilon[i]); }

    const int trace = getenv("ORC_TRACE") != NULL;      /* diagnostic: where the time of a large evaluation goes */
    double tt0 = omp_get_wtime();
    excl_t ex; build_exclusions(n, n_exc, exc_pairs, &ex);
    if (trace) fprintf(stderr, "[orc] exclusions %.2f s\n", omp_get_wtime() - tt0);
    pairlist_t pl = {NULL, 0, 0, NULL, 0};
    if (anyPeriodic) {
        double minAllowed = 1.999999 * cfg->cutoff; /* :201-204 */
        if (box[0] < minAllowed || box[4] < minAllowed || box[8] < minAllowed) { free(sig); free(eps); free_exclusions(&ex); return -1; }
    }
    tt0 = omp_get_wtime();
    if (method != ORC_NoCutoff) build_pairlist(n, pos, box, anyPeriodic, cfg->cutoff, &ex, &pl);
    if (trace) fprintf(stderr, "[orc] pair list (%lld pairs) %.2f s\n", pl.n, omp_get_wtime() - tt0);
    g_last_pairs = pl.n;

    ctx_t c;
    c.cfg = cfg; c.pos = pos; c.box = box; c.subset = subset; c.sig = sig; c.eps = eps; c.q = charge; c.lam = lambdas;
    c.periodic = periodic; /* setPeriodic is also called for Ewald/PME, but those never reach one_ixn */
    /* setUseCutoff (ReferenceSlicedLJCoulombIxn.cpp:60-68) */
    c.krf = pow(cfg->cutoff, -3.0) * (cfg->rf_dielectric - 1.0) / (2.0 * cfg->rf_dielectric + 1.0);
    c.crf = (1.0 / cfg->cutoff) * (3.0 * cfg->rf_dielectric) / (2.0 * cfg->rf_dielectric + 1.0);

    orc_config local = *cfg;
    if (ljpme) local.use_switch = 0; /* Q2: LJPME disables the switch on Reference (:174) */
    if (method == ORC_NoCutoff) local.use_switch = 0; /* :149-152 */
    c.cfg = &local;

    /* calculatePairIxn (ReferenceSlicedLJCoulombIxn.cpp:528-554) */
    if (ewald || pme || ljpme) {
        ewald_ixn(&c, &ex, &pl, forces, sliceE);
    } else if (local.include_direct) {
        if (method != ORC_NoCutoff) {
            pairs_threaded(&c, &pl, one_pair, forces, sliceE);
        } else {
            for (int ii = 0; ii < n; ii++)
                for (int jj = ii + 1; jj < n; jj++)
                    if (!is_excluded(&ex, jj, ii)) one_ixn(&c, ii, jj, forces, sliceE);
            g_last_pairs = (long long)n * (n - 1) / 2;
        }
    }

    if (local.include_direct) {
        /* 1-4 exceptions (ReferenceNonbondedSlicingKernels.cpp:233-243; ReferenceSlicedLJCoulomb14.cpp:61-95) */
        int exPeriodic = (method == ORC_NoCutoff || method == ORC_CutoffNonPeriodic) ? 0 : cfg->exceptions_periodic;
        for (int k = 0; k < n_exc; k++) {
            if (!(exc_chargeprod[k] != 0.0 || exc_epsilon[k] != 0.0)) continue; /* Q6 */
            int i = exc_pairs[2 * k], j = exc_pairs[2 * k + 1];
            int slice = slice_index(subset[i], subset[j]);
            double p0 = exc_sigma[k], p1 = 4.0 * exc_epsilon[k], p2 = exc_chargeprod[k];
            double d[3];
            if (exPeriodic) delta_periodic(pos + 3 * j, pos + 3 * i, box, d); else delta_plain(pos + 3 * j, pos + 3 * i, d);
            double inverseR = 1.0 / sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double sig2 = inverseR * p0; sig2 *= sig2;
            double sig6 = sig2 * sig2 * sig2;
            double dEdR = lambdas[2 * slice + VDW] * p1 * (12.0 * sig6 - 6.0) * sig6;
            dEdR += lambdas[2 * slice + COUL] * ORC_ONE_4PI_EPS0 * p2 * inverseR;
            dEdR *= inverseR * inverseR;
            for (int kk = 0; kk < 3; kk++) { double f = dEdR * d[kk]; forces[3 * i + kk] += f; forces[3 * j + kk] -= f; }
            sliceE[2 * slice + VDW] += p1 * (sig6 - 1.0) * sig6;
            sliceE[2 * slice + COUL] += ORC_ONE_4PI_EPS0 * p2 * inverseR;
        }
        /* long-range dispersion correction (:244-249): not for LJPME (Q2) */
        if ((periodic || ewald || pme) && cfg->use_dispersion_correction) {
            /* the reference computes the coefficients once, at default global-parameter values
             * (SlicedNonbondedForceImpl.cpp:281-291); the caller passes them in disp_coef (NULL = derive here) */
            double* coef = (double*)malloc(sizeof(double) * (size_t)S);
            if (disp_coef) memcpy(coef, disp_coef, sizeof(double) * (size_t)S);
            else orc_dispersion_coefficients(n, ns, sigma, epsilon, subset, cfg->cutoff, cfg->use_switch, cfg->switch_distance, coef);
            double volume = box[0] * box[4] * box[8];
            for (int s = 0; s < S; s++) sliceE[2 * s + VDW] += coef[s] / volume;
            free(coef);
        }
    }
    free(sig); free(eps); free_exclusions(&ex); free_pairlist(&pl);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * DIAGNOSTIC for the parity tests -- not part of the restated path.  The truncated pair potential is discontinuous at the cutoff
 * (ReferenceSlicedLJCoulombIxn.cpp:367 takes every pair of the neighbour list, i.e. r < cutoff, at full strength), so an
 * implementation that evaluates r^2 in single precision may put a pair whose r^2 lies within rounding of cutoff^2 on the other side.
 * This lists those pairs:  | r^2 / cutoff^2 - 1 | < rel_eps,  non-excluded, with what each would contribute:
 *   out_ij[2k], out_ij[2k+1] = the atoms;  out_vals[4k..4k+3] = r, |F| on either atom (lambda-scaled, as in the forces),
 *   raw Coulomb energy, raw vdW energy of the pair (as added to the slice energies).
 * Returns the number of band pairs (only the first max_out are stored), or a negative error code.
 * ---------------------------------------------------------------------------------------------- */
long long orc_cutoff_band_pairs(const orc_config* cfg, const double* pos, const double* box,
                                const double* charge, const double* sigma, const double* epsilon, const int* subset,
                                int n_exc, const int* exc_pairs, const double* lambdas, double rel_eps,
                                long long max_out, int* out_ij, double* out_vals) {
    int n = cfg->n_atoms, ns = cfg->n_subsets, S = ns * (ns + 1) / 2;
    int method = cfg->method;
    if (method == ORC_NoCutoff || n < 0 || rel_eps <= 0 || rel_eps > 0.1) return -2;
    int anyPeriodic = method >= ORC_CutoffPeriodic;
    double* sig = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* eps = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) { sig[i] = 0.5 * sigma[i]; eps[i] = 2.0 * sqrt(epsilon[i]); }
    const int trace = getenv("ORC_TRACE") != NULL;      /* diagnostic: where the time of a large evaluation goes */
    double tt0 = omp_get_wtime();
    excl_t ex; build_exclusions(n, n_exc, exc_pairs, &ex);
    if (trace) fprintf(stderr, "[orc] exclusions %.2f s\n", omp_get_wtime() - tt0);
    pairlist_t pl = {NULL, 0, 0, NULL, 0};
    const double rc2 = cfg->cutoff * cfg->cutoff;
    build_pairlist_band(n, pos, box, anyPeriodic, cfg->cutoff * sqrt(1.0 + rel_eps) * (1.0 + 1e-12), rc2 * (1.0 - rel_eps), rc2 * (1.0 + rel_eps), &ex, &pl);
    ctx_t c;
    orc_config local = *cfg;
    if (method == ORC_LJPME || method == ORC_NoCutoff) local.use_switch = 0;
    c.cfg = &local; c.pos = pos; c.box = box; c.subset = subset; c.sig = sig; c.eps = eps; c.q = charge; c.lam = lambdas;
    c.periodic = method == ORC_CutoffPeriodic;
    c.krf = pow(cfg->cutoff, -3.0) * (cfg->rf_dielectric - 1.0) / (2.0 * cfg->rf_dielectric + 1.0);
    c.crf = (1.0 / cfg->cutoff) * (3.0 * cfg->rf_dielectric) / (2.0 * cfg->rf_dielectric + 1.0);
    double* sliceE = (double*)malloc(sizeof(double) * 2 * (size_t)S);
    for (long long p = 0; p < pl.n && p < max_out; p++) {
        int ii = pl.ij[2 * p], jj = pl.ij[2 * p + 1];
        double fi[3] = {0, 0, 0}, fj[3] = {0, 0, 0}, d[3];
        for (int s = 0; s < 2 * S; s++) sliceE[s] = 0;
        if (method >= ORC_Ewald) ewald_pair(&c, ii, jj, fi, fj, sliceE); else one_pair(&c, ii, jj, fi, fj, sliceE);
        if (anyPeriodic) delta_periodic(pos + 3 * jj, pos + 3 * ii, box, d); else delta_plain(pos + 3 * jj, pos + 3 * ii, d);
        int slice = slice_index(subset[ii], subset[jj]);
        out_ij[2 * p] = ii; out_ij[2 * p + 1] = jj;
        out_vals[4 * p] = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        out_vals[4 * p + 1] = sqrt(fi[0] * fi[0] + fi[1] * fi[1] + fi[2] * fi[2]);
        out_vals[4 * p + 2] = sliceE[2 * slice + COUL]; out_vals[4 * p + 3] = sliceE[2 * slice + VDW];
    }
    long long count = pl.n;
    free(sliceE); free(sig); free(eps); free_exclusions(&ex); free_pairlist(&pl);
    return count;
}

/* number of OpenMP threads the PME sections of the oracle use (the pair loop is serial, like the reference's) */
int orc_num_threads(void) { return omp_get_max_threads(); }

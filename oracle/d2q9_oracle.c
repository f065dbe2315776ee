/*
 * d2q9_oracle.c — CPU restatement of the reference's D2Q9-BGK timestep path (see d2q9_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — the checker and the reported CPU baseline, never the product.
 * Parity status: PINNED against the compiled, unmodified reference (oracle/_ref, `make ref`):
 * byte-identical final_state.dat and av_vels.dat, see tests/golden/digests.json.
 *
 * Build with `gcc -std=c99 -O3` (ISO mode: no FMA contraction, which is what makes the per-cell
 * arithmetic reproduce the reference bit for bit).  The structure is this repo's
 * own (one scalar cell kernel + row drivers), the ARITHMETIC ORDER is the reference's and each
 * block says which lines it follows.  Paths are relative to /root/reference/.
 */
#include "d2q9_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define Q ORACLE_NSPEEDS

/* ------------------------------------------------------------------------------------------ */
/* input files                                                                                */
/* ------------------------------------------------------------------------------------------ */

static int fail(char* err, size_t errlen, const char* msg)
{
  if (err && errlen) {
    strncpy(err, msg, errlen - 1);
    err[errlen - 1] = '\0';
  }
  return -1;
}

/* d2q9-bgk.c:772-803: seven fscanf's in fixed order, each with its own die() message. */
int oracle_read_params(const char* path, oracle_params* p, char* err, size_t errlen)
{
  char msg[1200];
  FILE* fp = fopen(path, "r");
  if (!fp) {
    snprintf(msg, sizeof msg, "could not open input parameter file: %s", path);   /* :776 */
    return fail(err, errlen, msg);
  }
  static const char* const names[7] = {"nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega"};
  int* ints[4] = {&p->nx, &p->ny, &p->max_iters, &p->reynolds_dim};
  float* flts[3] = {&p->density, &p->accel, &p->omega};
  for (int i = 0; i < 7; i++) {
    int got = (i < 4) ? fscanf(fp, "%d\n", ints[i]) : fscanf(fp, "%f\n", flts[i - 4]);   /* :781-800 */
    if (got != 1) {
      fclose(fp);
      snprintf(msg, sizeof msg, "could not read param file: %s", names[i]);
      return fail(err, errlen, msg);
    }
  }
  fclose(fp);
  return 0;
}

/* d2q9-bgk.c:917-953: "x y 1" lines, duplicates allowed, free-cell count drops only on first mark. */
int oracle_read_obstacles(const char* path, int nx, int ny, int* obstacles, int* free_cells,
                          char* err, size_t errlen)
{
  char msg[1200];
  memset(obstacles, 0, sizeof(int) * (size_t)nx * (size_t)ny);          /* :918-922 */
  int nfree = nx * ny;                                                   /* :805 */
  FILE* fp = fopen(path, "r");
  if (!fp) {
    snprintf(msg, sizeof msg, "could not open input obstacles file: %s", path);   /* :928 */
    return fail(err, errlen, msg);
  }
  int xx, yy, blocked, got;
  while ((got = fscanf(fp, "%d %d %d\n", &xx, &yy, &blocked)) != EOF) {  /* :933 */
    const char* bad = NULL;
    if (got != 3) bad = "expected 3 values per line in obstacle file";  /* :936 */
    else if (xx < 0 || xx > nx - 1) bad = "obstacle x-coord out of range";   /* :938 */
    else if (yy < 0 || yy > ny - 1) bad = "obstacle y-coord out of range";   /* :940 */
    else if (blocked != 1) bad = "obstacle blocked value should be 1";  /* :942 */
    if (bad) {
      fclose(fp);
      return fail(err, errlen, bad);
    }
    if (obstacles[(size_t)yy * nx + xx] == 0) nfree--;                   /* :945-946 */
    obstacles[(size_t)yy * nx + xx] = blocked;                           /* :947 */
  }
  fclose(fp);
  *free_cells = nfree;
  return 0;
}

/* d2q9-bgk.c:834-862. */
void oracle_decompose(int ny, int size, int* ny_local, int* displs)
{
  int base = ny / size;
  int left = ny % size;
  int extra_last = 0, less_second_last = 0;
  if (base < 3 && left) {            /* :840-843 */
    left--;
    extra_last = 1;
  } else if (base < 3 && !left) {    /* :844-847 */
    extra_last = 1;
    less_second_last = 1;
  }
  for (int r = 0; r < size; r++) {   /* :850-862 */
    if (r < size - 2) ny_local[r] = base;
    else if (r == size - 2) ny_local[r] = base - less_second_last;
    else ny_local[r] = base + extra_last;   /* r == size-1 */
    if (r < left) ny_local[r]++;
    displs[r] = (r == 0) ? 0 : displs[r - 1] + ny_local[r - 1];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* state                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* d2q9-bgk.c:880-902: every cell (obstacles too) gets the rest equilibrium. */
void oracle_init_cells(const oracle_params* p, float* cells, int rows)
{
  const float w0 = p->density * 4.0f / 9.0f;   /* :880 */
  const float w1 = p->density / 9.0f;          /* :881 */
  const float w2 = p->density / 36.0f;         /* :882 */
  const size_t n = (size_t)rows * (size_t)p->nx;
  for (size_t c = 0; c < n; c++) {
    float* f = cells + c * Q;
    f[0] = w0;
    f[1] = f[2] = f[3] = f[4] = w1;
    f[5] = f[6] = f[7] = f[8] = w2;
  }
}

/* d2q9-bgk.c:442-478, one row. */
void oracle_accelerate_row(const oracle_params* p, float* row_cells, const int* row_obstacles)
{
  const float w1 = p->density * p->accel * 0.111111111111111111111111f;    /* :445 */
  const float w2 = p->density * p->accel * 0.0277777777777777777777778f;   /* :446 */
  for (int x = 0; x < p->nx; x++) {
    float* f = row_cells + (size_t)x * Q;
    if (!row_obstacles[x] && f[3] - w1 > 0.0f && f[6] - w2 > 0.0f && f[7] - w2 > 0.0f) {   /* :457-460 */
      f[1] += w1; f[5] += w2; f[8] += w2;     /* :463-465 */
      f[3] -= w1; f[6] -= w2; f[7] -= w2;     /* :467-469 */
    }
  }
}

/*
 * d2q9-bgk.c:493-704, organised by this restatement as three passes per lattice row so that the
 * arithmetic pass is a plain unit-stride loop over struct-of-array scratch rows (gcc vectorises it;
 * IEEE semantics make that invisible in the results):
 *   pass 1  pull: the nine streamed-in populations of every cell of the row   (:526-538)
 *   pass 2  moments, equilibrium, relaxation; per-cell tot_u term in double   (:546-666)
 *   pass 3  fluid/obstacle select, AoS store, in-order float accumulation     (:649-698)
 */
typedef struct row_scratch {
  int nx;
  float* t;      /* 9 rows of nx: streamed-in populations  */
  float* o;      /* 9 rows of nx: relaxed populations      */
  float* aux;    /* 2 rows of nx: m^2 and 1/rho            */
  double* term;  /* nx: sqrt(m^2)*rinv in double (:667)    */
} row_scratch;

static __thread row_scratch tls_scratch;

static row_scratch* scratch_for(int nx)
{
  row_scratch* s = &tls_scratch;
  if (s->nx < nx) {
    free(s->t); free(s->o); free(s->aux); free(s->term);
    s->t = (float*)malloc(sizeof(float) * Q * (size_t)nx);
    s->o = (float*)malloc(sizeof(float) * Q * (size_t)nx);
    s->aux = (float*)malloc(sizeof(float) * 2 * (size_t)nx);
    s->term = (double*)malloc(sizeof(double) * (size_t)nx);
    s->nx = (s->t && s->o && s->aux && s->term) ? nx : 0;
    if (!s->nx) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  }
  return s;
}

static void pull_row(int nx, const float* south, const float* here, const float* north, float* t)
{
  float *t0 = t, *t1 = t + nx, *t2 = t + 2 * (size_t)nx, *t3 = t + 3 * (size_t)nx, *t4 = t + 4 * (size_t)nx,
        *t5 = t + 5 * (size_t)nx, *t6 = t + 6 * (size_t)nx, *t7 = t + 7 * (size_t)nx, *t8 = t + 8 * (size_t)nx;
  for (int x = 0; x < nx; x++) {
    const int xe = (x + 1 >= nx) ? x + 1 - nx : x + 1;                    /* :527-528 */
    const int xw = (x == 0) ? nx - 1 : x - 1;                             /* :529 */
    t0[x] = here[(size_t)x * Q + 0];                                      /* :530 */
    t1[x] = here[(size_t)xw * Q + 1];                                     /* :531 */
    t2[x] = south[(size_t)x * Q + 2];                                     /* :532 */
    t3[x] = here[(size_t)xe * Q + 3];                                     /* :533 */
    t4[x] = north[(size_t)x * Q + 4];                                     /* :534 */
    t5[x] = south[(size_t)xw * Q + 5];                                    /* :535 */
    t6[x] = south[(size_t)xe * Q + 6];                                    /* :536 */
    t7[x] = north[(size_t)xe * Q + 7];                                    /* :537 */
    t8[x] = north[(size_t)xw * Q + 8];                                    /* :538 */
  }
}

static void relax_row(int nx, float omega, const float* restrict t, float* restrict o,
                      float* restrict msq_row, float* restrict rinv_row, double* restrict term)
{
  const float csq_inv = 3.0f;                                             /* :497 */
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;      /* :499-501 */
#define ROW(base, k) ((base) + (size_t)(k) * (size_t)nx)
  const float* restrict t0 = ROW(t, 0); const float* restrict t1 = ROW(t, 1); const float* restrict t2 = ROW(t, 2);
  const float* restrict t3 = ROW(t, 3); const float* restrict t4 = ROW(t, 4); const float* restrict t5 = ROW(t, 5);
  const float* restrict t6 = ROW(t, 6); const float* restrict t7 = ROW(t, 7); const float* restrict t8 = ROW(t, 8);
  float* restrict o0 = ROW(o, 0); float* restrict o1 = ROW(o, 1); float* restrict o2 = ROW(o, 2);
  float* restrict o3 = ROW(o, 3); float* restrict o4 = ROW(o, 4); float* restrict o5 = ROW(o, 5);
  float* restrict o6 = ROW(o, 6); float* restrict o7 = ROW(o, 7); float* restrict o8 = ROW(o, 8);
#undef ROW
#pragma GCC ivdep   /* the 21 scratch rows never overlap */
  for (int x = 0; x < nx; x++) {
    float rho = t0[x];                                                    /* :546-554, left to right */
    rho += t1[x]; rho += t2[x]; rho += t3[x]; rho += t4[x];
    rho += t5[x]; rho += t6[x]; rho += t7[x]; rho += t8[x];
    const float rinv = 1.0f / rho;                                        /* :561 */
    float mx = t1[x] + t5[x];                                             /* :570-574 (momentum, NOT divided by rho: :575) */
    mx += t8[x]; mx -= t3[x]; mx -= t6[x]; mx -= t7[x];
    float my = t2[x] + t5[x];                                             /* :576-580 */
    my += t6[x]; my -= t4[x]; my -= t7[x]; my -= t8[x];
    const float msq = mx * mx + my * my;                                  /* :589 */
    const float e5 = mx + my, e6 = -mx + my, e7 = -mx - my, e8 = mx - my; /* :600-603 */
    const float e3 = -mx, e4 = -my;                                       /* :598-599 */
    const float a1 = mx * csq_inv, a2 = my * csq_inv, a3 = e3 * csq_inv, a4 = e4 * csq_inv,   /* :610-617 */
                a5 = e5 * csq_inv, a6 = e6 * csq_inv, a7 = e7 * csq_inv, a8 = e8 * csq_inv;
    const float b1 = a1 * mx, b2 = a2 * my, b3 = a3 * e3, b4 = a4 * e4,   /* :624-631 */
                b5 = a5 * e5, b6 = a6 * e6, b7 = a7 * e7, b8 = a8 * e8;
    const float h = 0.5f * rinv * csq_inv;                                /* prefix of :638-646: (0.5f*densinv)*ic_sq */
    const float q0 = w0 * (rho - h * msq);                                /* :638 */
    const float q1 = w1 * (rho + a1 + h * (b1 - msq));                    /* :639 */
    const float q2 = w1 * (rho + a2 + h * (b2 - msq));
    const float q3 = w1 * (rho + a3 + h * (b3 - msq));
    const float q4 = w1 * (rho + a4 + h * (b4 - msq));
    const float q5 = w2 * (rho + a5 + h * (b5 - msq));
    const float q6 = w2 * (rho + a6 + h * (b6 - msq));
    const float q7 = w2 * (rho + a7 + h * (b7 - msq));
    const float q8 = w2 * (rho + a8 + h * (b8 - msq));                    /* :646 */
    o0[x] = t0[x] + omega * (q0 - t0[x]);                                 /* :658-666 */
    o1[x] = t1[x] + omega * (q1 - t1[x]);
    o2[x] = t2[x] + omega * (q2 - t2[x]);
    o3[x] = t3[x] + omega * (q3 - t3[x]);
    o4[x] = t4[x] + omega * (q4 - t4[x]);
    o5[x] = t5[x] + omega * (q5 - t5[x]);
    o6[x] = t6[x] + omega * (q6 - t6[x]);
    o7[x] = t7[x] + omega * (q7 - t7[x]);
    o8[x] = t8[x] + omega * (q8 - t8[x]);
    msq_row[x] = msq;
    rinv_row[x] = rinv;
  }
  for (int x = 0; x < nx; x++)
    term[x] = sqrt((double)msq_row[x]) * (double)rinv_row[x];            /* :667 — sqrt and product in double */
}

float oracle_timestep_rows(const oracle_params* p, const float* cells, float* tmp_cells,
                           const int* obstacles, int start, int end, double* terms)
{
  static const int opposite[Q] = {0, 3, 4, 1, 2, 7, 8, 5, 6};             /* :687-695 */
  const int nx = p->nx;
  row_scratch* s = scratch_for(nx);
  float tot_u = 0.0f;                                                     /* :502 */
  for (int y = start; y < end; y++) {
    const float* south = cells + (size_t)(y - 1) * nx * Q;                /* y_s :512 */
    const float* here  = cells + (size_t)y * nx * Q;
    const float* north = cells + (size_t)(y + 1) * nx * Q;                /* y_n :511 */
    pull_row(nx, south, here, north, s->t);
    relax_row(nx, p->omega, s->t, s->o, s->aux, s->aux + nx, s->term);
    const int* orow = obstacles + (size_t)y * nx;
    float* dst = tmp_cells + (size_t)y * nx * Q;
    for (int x = 0; x < nx; x++) {
      if (!orow[x]) {                                                     /* :655 / :674 */
        for (int k = 0; k < Q; k++) dst[(size_t)x * Q + k] = s->o[(size_t)k * nx + x];
        tot_u += s->term[x];                                              /* float += double, :667/:684 */
        if (terms) terms[(size_t)y * nx + x] = s->term[x];
      } else {
        for (int k = 0; k < Q; k++) dst[(size_t)x * Q + opposite[k]] = s->t[(size_t)k * nx + x];
        if (terms) terms[(size_t)y * nx + x] = 0.0;
      }
    }
  }
  return tot_u;
}

/* d2q9-bgk.c:716-751. */
float oracle_av_velocity_sum(const oracle_params* p, const float* cells, const int* obstacles, int rows)
{
  float tot_u = 0.0f;
  const size_t n = (size_t)rows * (size_t)p->nx;
  for (size_t c = 0; c < n; c++) {
    if (obstacles[c]) continue;                                           /* :721 */
    const float* f = cells + c * Q;
    float rho = 0.0f;                                                     /* :724-729 */
    for (int k = 0; k < Q; k++) rho += f[k];
    const float ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;   /* :732-738 */
    const float uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;   /* :740-746 */
    tot_u += sqrt((ux * ux) + (uy * uy));                                 /* :748 */
  }
  return tot_u;
}

/* d2q9-bgk.c:1005-1007. */
float oracle_reynolds(const oracle_params* p, float av_velocity)
{
  const float viscosity = 1.0f / 6.0f * (2.0f / p->omega - 1.0f);
  return av_velocity * p->reynolds_dim / viscosity;
}

/* d2q9-bgk.c:1076-1112. */
void oracle_cell_observables(const oracle_params* p, const float* f, int obstacle,
                             float* u_x, float* u_y, float* u, float* pressure)
{
  const float c_sq = 1.0f / 3.0f;                                         /* :1040 */
  if (obstacle) {                                                         /* :1076-1080 */
    *u_x = *u_y = *u = 0.0f;
    *pressure = p->density * c_sq;
    return;
  }
  float rho = 0.0f;                                                       /* :1084-1090 */
  for (int k = 0; k < Q; k++) rho += f[k];
  *u_x = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;               /* :1093-1099 */
  *u_y = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;               /* :1101-1107 */
  *u = sqrt((*u_x * *u_x) + (*u_y * *u_y));                               /* :1109 */
  *pressure = rho * c_sq;                                                 /* :1111 */
}

/* ------------------------------------------------------------------------------------------ */
/* whole-run drivers (d2q9-bgk.c:315-396 for a single rank)                                   */
/* ------------------------------------------------------------------------------------------ */

typedef struct run_state {
  float* cells;      /* (ny+2)*nx*Q */
  float* tmp;
  int* obst;         /* (ny+2)*nx, rows 1..ny used (:875) */
  float free_cells_inv;
} run_state;

static int run_alloc(const oracle_params* p, const int* obstacles, int free_cells, run_state* s)
{
  const size_t row = (size_t)p->nx;
  const size_t n = (size_t)(p->ny + 2) * row;
  s->cells = (float*)calloc(n * Q, sizeof(float));
  s->tmp = (float*)calloc(n * Q, sizeof(float));
  s->obst = (int*)calloc(n, sizeof(int));
  if (!s->cells || !s->tmp || !s->obst) return -1;
  oracle_init_cells(p, s->cells + row * Q, p->ny);
  memcpy(s->obst + row, obstacles, sizeof(int) * row * (size_t)p->ny);
  s->free_cells_inv = 1.0f / free_cells;                                  /* :950 */
  return 0;
}

static void run_free(run_state* s)
{
  free(s->cells);
  free(s->tmp);
  free(s->obst);
}

/* The single-rank exchange of :295-303,326-327: top = bottom = self, so halo row ny+1 receives
 * owned row 1 and halo row 0 receives owned row ny. */
static void self_exchange(const oracle_params* p, float* cells)
{
  const size_t rowf = (size_t)p->nx * Q;
  memcpy(cells + (size_t)(p->ny + 1) * rowf, cells + rowf, rowf * sizeof(float));
  memcpy(cells, cells + (size_t)p->ny * rowf, rowf * sizeof(float));
}

static void split_rows(int first, int last_excl, int parts, int part, int* a, int* b)
{
  const int n = last_excl - first;
  const int base = n / parts, rem = n % parts;
  *a = first + part * base + (part < rem ? part : rem);
  *b = *a + base + (part < rem ? 1 : 0);
}

int oracle_run(const oracle_params* p, const int* obstacles, int free_cells, int n_steps,
               int nthreads, float* cells_out, float* av_vels, double* av_exact)
{
  run_state s;
  if (run_alloc(p, obstacles, free_cells, &s)) return -1;
  const int nx = p->nx, ny = p->ny;
  double* terms = (double*)malloc(sizeof(double) * (size_t)(ny + 2) * nx);
  if (!terms) { run_free(&s); return -1; }
  if (nthreads < 1) nthreads = 1;
  if (nthreads > ny) nthreads = ny;

  for (int tt = 0; tt < n_steps; tt++) {                                  /* :315 */
    self_exchange(p, s.cells);                                            /* :326-327 (+ :364) */
    oracle_accelerate_row(p, s.cells + (size_t)(ny - 1) * nx * Q, s.obst + (size_t)(ny - 1) * nx);   /* :345-348, ii = ny_local-1 (:449) */

#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int part = 0; part < nthreads; part++) {
      int a, b;
      split_rows(1, ny + 1, nthreads, part, &a, &b);
      (void)oracle_timestep_rows(p, s.cells, s.tmp, s.obst, a, b, terms);
    }

    /* reference summation order: interior rows [2,ny) (:350), then row 1 (:365), then row ny (:366) */
    float part_sum[3];
    const int ranges[3][2] = {{2, ny}, {1, 2}, {ny, ny + 1}};
    for (int r = 0; r < 3; r++) {
      float acc = 0.0f;
      for (int y = ranges[r][0]; y < ranges[r][1]; y++) {
        const double* trow = terms + (size_t)y * nx;
        const int* orow = s.obst + (size_t)y * nx;
        for (int x = 0; x < nx; x++)
          if (!orow[x]) acc += trow[x];
      }
      part_sum[r] = acc;
    }
    float local = part_sum[0];
    local += part_sum[1];
    local += part_sum[2];
    av_vels[tt] = local * s.free_cells_inv;                               /* :367 */

    if (av_exact) {
      double tot = 0.0;
      for (int y = 1; y <= ny; y++) {
        double rowsum = 0.0;
        const double* trow = terms + (size_t)y * nx;
        for (int x = 0; x < nx; x++) rowsum += trow[x];
        tot += rowsum;
      }
      av_exact[tt] = tot * (double)s.free_cells_inv;
    }

    float* sw = s.cells;                                                  /* :376-378 */
    s.cells = s.tmp;
    s.tmp = sw;
  }
  memcpy(cells_out, s.cells + (size_t)nx * Q, sizeof(float) * (size_t)ny * nx * Q);
  free(terms);
  run_free(&s);
  return 0;
}

int oracle_run_fast(const oracle_params* p, const int* obstacles, int free_cells, int n_steps,
                    int nthreads, float* cells_out, float* av_vels)
{
  run_state s;
  if (run_alloc(p, obstacles, free_cells, &s)) return -1;
  const int nx = p->nx, ny = p->ny;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > ny) nthreads = ny;
  float* partial = (float*)malloc(sizeof(float) * (size_t)nthreads);
  if (!partial) { run_free(&s); return -1; }

  for (int tt = 0; tt < n_steps; tt++) {
    self_exchange(p, s.cells);
    oracle_accelerate_row(p, s.cells + (size_t)(ny - 1) * nx * Q, s.obst + (size_t)(ny - 1) * nx);
    float local;
    if (nthreads == 1) {
      local = oracle_timestep_rows(p, s.cells, s.tmp, s.obst, 2, ny, NULL);          /* :350 */
      local += oracle_timestep_rows(p, s.cells, s.tmp, s.obst, 1, 2, NULL);          /* :365 */
      local += oracle_timestep_rows(p, s.cells, s.tmp, s.obst, ny, ny + 1, NULL);    /* :366 */
      local = local * s.free_cells_inv;                                              /* :367 */
    } else {
#pragma omp parallel for schedule(static) num_threads(nthreads)
      for (int part = 0; part < nthreads; part++) {
        int a, b;
        split_rows(1, ny + 1, nthreads, part, &a, &b);
        partial[part] = oracle_timestep_rows(p, s.cells, s.tmp, s.obst, a, b, NULL) * s.free_cells_inv;
      }
      local = 0.0f;                                                                  /* rank-order sum, as :396 */
      for (int part = 0; part < nthreads; part++) local += partial[part];
    }
    av_vels[tt] = local;
    float* sw = s.cells;
    s.cells = s.tmp;
    s.tmp = sw;
  }
  memcpy(cells_out, s.cells + (size_t)nx * Q, sizeof(float) * (size_t)ny * nx * Q);
  free(partial);
  run_free(&s);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* output files                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* d2q9-bgk.c:1054-1120. */
int oracle_write_final_state(const char* path, const oracle_params* p, const float* cells,
                             const int* obstacles, int rows, int displ, int append)
{
  FILE* fp = fopen(path, append ? "a" : "w");
  if (!fp) return -1;
  static char buf[1 << 20];
  setvbuf(fp, buf, _IOFBF, sizeof buf);
  for (int y = 0; y < rows; y++) {
    for (int x = 0; x < p->nx; x++) {
      const size_t c = (size_t)y * p->nx + x;
      float ux, uy, u, pr;
      oracle_cell_observables(p, cells + c * Q, obstacles[c], &ux, &uy, &u, &pr);
      fprintf(fp, "%d %d %.12E %.12E %.12E %.12E %d\n", x, y + displ, ux, uy, u, pr, obstacles[c]);   /* :1115 */
    }
  }
  fclose(fp);
  return 0;
}

/* d2q9-bgk.c:1127-1139. */
int oracle_write_av_vels(const char* path, const float* av_vels, int n)
{
  FILE* fp = fopen(path, "w");
  if (!fp) return -1;
  for (int i = 0; i < n; i++) fprintf(fp, "%d:\t%.12E\n", i, av_vels[i]);   /* :1136 */
  fclose(fp);
  return 0;
}

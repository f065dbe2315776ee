// lbm_host.cpp — host-only half of the C ABI (include/lbm_d2q9.h): the reference's input parsers,
// row decomposition, end-of-run reductions and output writers.  No HIP calls in this file; it is
// what the CLI shim and the Python host use around the device path, and it is exercised on
// GPU-less machines by the CPU test suite.  Reference lines are relative to the reference tree.

#include <algorithm>
#include <charconv>
#include <cmath>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lbm_d2q9.h"
#include "lbm_internal.h"

namespace {
thread_local std::string g_error;
}

namespace lbm_internal {
void set_error(const std::string& msg) { g_error = msg; }
}  // namespace lbm_internal

using lbm_internal::set_error;

extern "C" {

int lbm_abi_version(void) { return LBM_ABI_VERSION; }
const char* lbm_last_error(void) { return g_error.c_str(); }

// d2q9-bgk.c:772-803 — same token order, same conversion (fscanf %d / %f), same messages.
int lbm_read_params(const char* paramfile, lbm_params* out)
{
  if (!paramfile || !out) { set_error("lbm_read_params: null argument"); return 1; }
  std::FILE* fp = std::fopen(paramfile, "r");
  if (!fp) { set_error(std::string("could not open input parameter file: ") + paramfile); return 1; }   // :776
  struct Field { const char* name; const char* fmt; void* dst; };
  const Field fields[7] = {
      {"nx", "%d\n", &out->nx},           {"ny", "%d\n", &out->ny},
      {"maxIters", "%d\n", &out->max_iters}, {"reynolds_dim", "%d\n", &out->reynolds_dim},
      {"density", "%f\n", &out->density}, {"accel", "%f\n", &out->accel},
      {"omega", "%f\n", &out->omega}};
  for (const Field& f : fields) {
    if (std::fscanf(fp, f.fmt, f.dst) != 1) {                                                             // :781-800
      std::fclose(fp);
      set_error(std::string("could not read param file: ") + f.name);
      return 1;
    }
  }
  std::fclose(fp);
  return 0;
}

// d2q9-bgk.c:917-953.
int lbm_read_obstacles(const char* obstaclefile, int nx, int ny, int* obstacles, int* free_cells)
{
  if (!obstaclefile || !obstacles || !free_cells || nx <= 0 || ny <= 0) { set_error("lbm_read_obstacles: bad argument"); return 1; }
  const size_t n = static_cast<size_t>(nx) * static_cast<size_t>(ny);
  std::memset(obstacles, 0, sizeof(int) * n);                                                             // :918-922
  int nfree = nx * ny;                                                                                    // :805
  std::FILE* fp = std::fopen(obstaclefile, "r");
  if (!fp) { set_error(std::string("could not open input obstacles file: ") + obstaclefile); return 1; }  // :928
  int xx = 0, yy = 0, blocked = 0, got = 0;
  while ((got = std::fscanf(fp, "%d %d %d\n", &xx, &yy, &blocked)) != EOF) {                              // :933
    const char* bad = nullptr;
    if (got != 3) bad = "expected 3 values per line in obstacle file";                                    // :936
    else if (xx < 0 || xx > nx - 1) bad = "obstacle x-coord out of range";                                // :938
    else if (yy < 0 || yy > ny - 1) bad = "obstacle y-coord out of range";                                // :940
    else if (blocked != 1) bad = "obstacle blocked value should be 1";                                    // :942
    if (bad) { std::fclose(fp); set_error(bad); return 1; }
    int& cell = obstacles[static_cast<size_t>(yy) * nx + xx];
    if (cell == 0) --nfree;                                                                               // :945-946
    cell = blocked;                                                                                       // :947
  }
  std::fclose(fp);
  *free_cells = nfree;
  return 0;
}

// d2q9-bgk.c:834-862.
int lbm_decompose(int ny, int size, int* ny_local, int* displs)
{
  if (ny <= 0 || size <= 0 || !ny_local || !displs) { set_error("lbm_decompose: bad argument"); return 1; }
  int rows = ny / size, spare = ny % size;
  int last_gets_one = 0, second_last_gives_one = 0;
  if (rows < 3) {                 // the last rank must own >= 3 rows (:848-849)
    last_gets_one = 1;
    if (spare) --spare; else second_last_gives_one = 1;   // :840-847
  }
  int at = 0;
  for (int r = 0; r < size; ++r) {
    int n = rows;
    if (r == size - 2) n -= second_last_gives_one;
    if (r == size - 1) n += last_gets_one;
    if (r < spare) ++n;
    ny_local[r] = n;
    displs[r] = at;
    at += n;
  }
  return 0;
}

// Column blocks of the tile (2-D) decomposition — the reference never built one (report.odt "MPI Design" discusses it), so there is no
// rule to follow: whole x-PAIRS (the K-step kernels work on pairs of cells), nx / 2 pairs dealt as evenly as they go, the first ranks
// taking the spare ones.
int lbm_decompose_columns(int nx, int px, int* nx_local, int* displs)
{
  if (nx <= 0 || px <= 0 || !nx_local || !displs) { set_error("lbm_decompose_columns: bad argument"); return 1; }
  if (nx % 2 != 0 || nx / 2 < px) { set_error("lbm_decompose_columns: the tile decomposition needs an even nx and at least one x-pair per rank"); return 1; }
  const int pairs = nx / 2, each = pairs / px, spare = pairs % px;
  int at = 0;
  for (int r = 0; r < px; ++r) {
    nx_local[r] = 2 * (each + (r < spare ? 1 : 0));
    displs[r] = at;
    at += nx_local[r];
  }
  return 0;
}

int lbm_plan_next(int K, int four_rows, int tail4, int left)
{
  int k = left < K ? left : K;
  if (!four_rows || !tail4) return k;
  if (K == 3 && ((left % 3 == 1 && left >= 4) || (left % 3 == 2 && left >= 8))) k = 4;
  if (K == 4 && ((left % 4 == 3) || (left % 4 == 2 && left >= 6) || (left % 4 == 1 && left >= 9))) k = 3;
  return k;
}

// The launches between two halo exchanges of a partitioned run ("a group"): the next launches by lbm_plan_next for as long as their
// steps add up to at most `ghost` (the first launch of a group advances the ghost rows the later ones read), `group_max` launches at most.
int lbm_plan_group(int K, int ghost, int group_max, int left, int* steps, int cap)
{
  if (K < 1 || K > 4 || ghost < K || group_max < 1 || left < 0 || cap < 0 || (cap > 0 && !steps)) { lbm_internal::set_error("lbm_plan_group: bad argument"); return -1; }
  int n = 0, used = 0;
  while (left > 0 && n < group_max) {
    const int k = lbm_plan_next(K, ghost >= 4 ? 1 : 0, 1, left);
    if (used + k > ghost) break;
    if (n < cap) steps[n] = k;
    ++n; used += k; left -= k;
  }
  return n;
}

int lbm_plan_steps(int K, int four_rows, int n_steps, int* steps, int cap)
{
  if (K < 1 || K > 4 || n_steps < 0 || cap < 0 || (cap > 0 && !steps)) { lbm_internal::set_error("lbm_plan_steps: bad argument"); return -1; }
  int n = 0;
  for (int left = n_steps; left > 0; ++n) {
    const int k = lbm_plan_next(K, four_rows, 1, left);
    if (n < cap) steps[n] = k;
    left -= k;
  }
  return n;
}

// d2q9-bgk.c:716-751 (without the MPI_Reduce): float accumulator, double sqrt.
float lbm_av_velocity_host(const lbm_params* p, const float* cells, const int* obstacles, int rows)
{
  float tot_u = 0.0f;
  const size_t n = static_cast<size_t>(rows) * static_cast<size_t>(p->nx);
  for (size_t c = 0; c < n; ++c) {
    if (obstacles[c]) continue;                                                 // :721
    const float* f = cells + c * LBM_NSPEEDS;
    float rho = 0.0f;
    for (int k = 0; k < LBM_NSPEEDS; ++k) rho += f[k];                          // :724-729
    const float ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;         // :732-738
    const float uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;         // :740-746
    tot_u += std::sqrt(static_cast<double>((ux * ux) + (uy * uy)));            // :748
  }
  return tot_u;
}

// The same sum from device-computed u_x, u_y (lbm_get_observables): :748 in the reference's cell order.  The float
// accumulator makes the ORDER of the additions part of the result, so they stay serial; the double square roots
// (67 M of them at 8192x8192) do not depend on one another and are taken by several threads, a block of cells at a time.
float lbm_av_velocity_obs(const lbm_params* p, const float* obs, const int* obstacles, int rows)
{
  float tot_u = 0.0f;
  const size_t n = static_cast<size_t>(rows) * static_cast<size_t>(p->nx);
  const size_t block = size_t(1) << 20;
  unsigned workers = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
  if (n < block) workers = 1;
  std::vector<double> term(std::min(n, block));
  for (size_t c0 = 0; c0 < n; c0 += block) {
    const size_t m = std::min(block, n - c0);
    auto roots = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        const float ux = obs[4 * (c0 + i)], uy = obs[4 * (c0 + i) + 1];
        term[i] = std::sqrt(static_cast<double>((ux * ux) + (uy * uy)));       // :748
      }
    };
    if (workers == 1) {
      roots(0, m);
    } else {
      std::vector<std::thread> pool;
      const size_t per = (m + workers - 1) / workers;
      for (unsigned w = 0; w < workers; ++w) {
        const size_t lo = std::min(m, w * per), hi = std::min(m, lo + per);
        if (lo < hi) pool.emplace_back(roots, lo, hi);
      }
      for (std::thread& t : pool) t.join();
    }
    for (size_t i = 0; i < m; ++i)
      if (!obstacles[c0 + i]) tot_u += term[i];                                // :721, :748 — float += double, as the reference
  }
  return tot_u;
}

// d2q9-bgk.c:1005-1007.
float lbm_reynolds(const lbm_params* p, float av_velocity)
{
  const float viscosity = 1.0f / 6.0f * (2.0f / p->omega - 1.0f);
  return av_velocity * p->reynolds_dim / viscosity;
}

// d2q9-bgk.c:1054-1120.  Same bytes as the reference's fprintf("%d %d %.12E %.12E %.12E %.12E %d\n")
// (:1115), produced by std::to_chars (shortest-correct scientific formatting at precision 12 is what
// glibc prints too) over row blocks formatted in parallel and written in order: at 8192x8192 the file
// is 5.8 GB of text and formatting it dominates the run if done through stdio.
// `fluid(c, u_x, u_y, u, pressure)` supplies the four values of a non-obstacle cell (:1084-1111).
}  // extern "C"

namespace {
template <typename Fluid>
int write_final_state_impl(const char* path, const lbm_params* p, const int* obstacles, int rows, int displ, int append, Fluid fluid)
{
  std::FILE* fp = std::fopen(path, append ? "a" : "w");                        // :1054-1057
  if (!fp) { set_error("could not open file output file"); return 1; }         // :1061
  const int nx = p->nx;
  const float c_sq = 1.0f / 3.0f;                                              // :1040
  const float obstacle_pressure = p->density * c_sq;                           // :1079

  auto put_float = [](char* out, float v) -> char* {                          // "%.12E" of a float promoted to double
    const double d = static_cast<double>(v);
    if (std::isnan(d)) { const char* t = std::signbit(d) ? "-NAN" : "NAN"; while (*t) *out++ = *t++; return out; }
    if (std::isinf(d)) { const char* t = d < 0 ? "-INF" : "INF"; while (*t) *out++ = *t++; return out; }
    auto r = std::to_chars(out, out + 40, d, std::chars_format::scientific, 12);
    for (char* q = out; q < r.ptr; ++q)
      if (*q == 'e') { *q = 'E'; break; }
    return r.ptr;
  };
  auto put_int = [](char* out, int v) -> char* { return std::to_chars(out, out + 16, v).ptr; };

  auto format_rows = [&](int y0, int y1, std::string& text) {
    text.clear();
    text.reserve(static_cast<size_t>(y1 - y0) * nx * 96);
    char line[160];
    for (int y = y0; y < y1; ++y) {
      for (int x = 0; x < nx; ++x) {
        const size_t c = static_cast<size_t>(y) * nx + x;
        float u_x, u_y, u, pressure;
        if (obstacles[c]) {                                                    // :1076-1080
          u_x = u_y = u = 0.0f;
          pressure = obstacle_pressure;
        } else {
          fluid(c, u_x, u_y, u, pressure);
        }
        char* q = line;                                                        // :1115
        q = put_int(q, x); *q++ = ' ';
        q = put_int(q, y + displ); *q++ = ' ';
        q = put_float(q, u_x); *q++ = ' ';
        q = put_float(q, u_y); *q++ = ' ';
        q = put_float(q, u); *q++ = ' ';
        q = put_float(q, pressure); *q++ = ' ';
        q = put_int(q, obstacles[c]); *q++ = '\n';
        text.append(line, static_cast<size_t>(q - line));
      }
    }
  };

  // blocks of rows (~4 MB of text each), `workers` of them formatted concurrently, written in order
  const int rows_per_block = std::max(1, static_cast<int>((4u << 20) / (static_cast<size_t>(nx) * 90 + 1)));
  unsigned workers = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("LBM_WRITE_THREADS")) workers = static_cast<unsigned>(std::atoi(e));
  workers = std::max(1u, std::min(workers, 16u));
  std::vector<std::string> text(workers);
  bool ok = true;
  for (int y = 0; y < rows && ok; y += rows_per_block * static_cast<int>(workers)) {
    std::vector<std::thread> pool;
    int used = 0;
    for (unsigned w = 0; w < workers; ++w) {
      const int y0 = y + static_cast<int>(w) * rows_per_block;
      if (y0 >= rows) break;
      const int y1 = std::min(rows, y0 + rows_per_block);
      ++used;
      if (workers == 1) format_rows(y0, y1, text[w]);
      else pool.emplace_back(format_rows, y0, y1, std::ref(text[w]));
    }
    for (std::thread& t : pool) t.join();
    for (int w = 0; w < used; ++w)
      if (std::fwrite(text[w].data(), 1, text[w].size(), fp) != text[w].size()) { ok = false; break; }
  }
  if (std::fclose(fp) != 0) ok = false;
  if (!ok) { set_error("could not write file output file"); return 1; }
  return 0;
}
}  // namespace

extern "C" {

int lbm_write_final_state(const char* path, const lbm_params* p, const float* cells, const int* obstacles,
                          int rows, int displ, int append)
{
  if (!path || !p || !cells || !obstacles) { set_error("lbm_write_final_state: null argument"); return 1; }
  const float c_sq = 1.0f / 3.0f;
  return write_final_state_impl(path, p, obstacles, rows, displ, append, [=](size_t c, float& u_x, float& u_y, float& u, float& pressure) {
    const float* f = cells + c * LBM_NSPEEDS;
    float rho = 0.0f;
    for (int k = 0; k < LBM_NSPEEDS; ++k) rho += f[k];                       // :1084-1090
    u_x = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / rho;                 // :1093-1099
    u_y = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / rho;                 // :1101-1107
    u = static_cast<float>(std::sqrt(static_cast<double>((u_x * u_x) + (u_y * u_y))));   // :1109
    pressure = rho * c_sq;                                                   // :1111
  });
}

// The same file from lbm_get_observables() output: the four values were computed on the device.
int lbm_write_final_state_obs(const char* path, const lbm_params* p, const float* obs, const int* obstacles,
                              int rows, int displ, int append)
{
  if (!path || !p || !obs || !obstacles) { set_error("lbm_write_final_state_obs: null argument"); return 1; }
  return write_final_state_impl(path, p, obstacles, rows, displ, append, [=](size_t c, float& u_x, float& u_y, float& u, float& pressure) {
    const float* o = obs + 4 * c;
    u_x = o[0]; u_y = o[1]; u = o[2]; pressure = o[3];
  });
}

// d2q9-bgk.c:1127-1139.
int lbm_write_av_vels(const char* path, const float* av_vels, int n)
{
  std::FILE* fp = std::fopen(path, "w");
  if (!fp) { set_error("could not open file output file"); return 1; }         // :1131
  for (int i = 0; i < n; ++i) std::fprintf(fp, "%d:\t%.12E\n", i, av_vels[i]); // :1136
  std::fclose(fp);
  return 0;
}

}  // extern "C"

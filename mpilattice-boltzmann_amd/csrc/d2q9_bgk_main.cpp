// d2q9_bgk_main.cpp — the thin host shim: the reference's command-line contract
// (d2q9-bgk.c:153-440) in front of liblbm_d2q9.so.
//
//   d2q9-bgk <paramfile> <obstaclefile>
//
// writes final_state.dat and av_vels.dat into the cwd (d2q9-bgk.c:63-64) and prints the five
// stdout lines of d2q9-bgk.c:411-415 byte-compatibly, followed by extra lines (MLUPS, roofline).
// Everything between tic and toc is the device path; this file only parses, times and writes.
// Environment (optional): LBM_DEVICE=<hip ordinal>, LBM_NO_OUTPUT=1 (like the reference's
// -DPROFILE build, :419-421), LBM_FLAGS=<lbm_create flags>.
//
// LBM_GPUS=N (N > 1) plays the role of `mpirun -np N` (mpi_submit:63) inside ONE process: the rows are
// partitioned by the reference's rule (d2q9-bgk.c:834-862), rank r lives on device LBM_DEVICES[r] (a comma
// list; default r modulo the device count), one host thread per rank drives its device, and the halos travel
// as direct peer-to-peer stores over xGMI (include/lbm_d2q9_p2p.h).  The ranks' observables are gathered in
// rank order — the order in which the reference's ranks append to final_state.dat (:1049-1057).

#include <sys/resource.h>
#include <sys/time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "lbm_d2q9.h"
#include "lbm_d2q9_p2p.h"

namespace {

[[noreturn]] void die(const char* message, int line, const char* file)   // d2q9-bgk.c:1145-1151
{
  std::fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  std::fprintf(stderr, "%s\n", message);
  std::fflush(stderr);
  std::exit(EXIT_FAILURE);
}

[[noreturn]] void usage(const char* exe)                                  // d2q9-bgk.c:1153-1157
{
  std::fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", exe);
  std::exit(EXIT_FAILURE);
}

double wall_seconds()
{
  timeval t;
  gettimeofday(&t, nullptr);
  return t.tv_sec + t.tv_usec / 1000000.0;
}

int env_int(const char* name, int dflt)
{
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

}  // namespace

int main(int argc, char* argv[])
{
  if (argc != 3) usage(argv[0]);                                           // :197-200

  lbm_params params;
  if (lbm_read_params(argv[1], &params)) die(lbm_last_error(), __LINE__, __FILE__);
  if (params.nx <= 0 || params.ny <= 0 || params.max_iters < 0) die("could not read param file: nx", __LINE__, __FILE__);
  std::vector<int> obstacles(static_cast<size_t>(params.nx) * params.ny);
  int free_cells = 0;
  if (lbm_read_obstacles(argv[2], params.nx, params.ny, obstacles.data(), &free_cells)) die(lbm_last_error(), __LINE__, __FILE__);

  const int ngpus = env_int("LBM_GPUS", 1);
  const unsigned flags = static_cast<unsigned>(env_int("LBM_FLAGS", 0));
  const size_t nx = static_cast<size_t>(params.nx);
  std::vector<float> av_vels(static_cast<size_t>(params.max_iters) + 1);
  std::vector<float> obs(static_cast<size_t>(params.nx) * params.ny * 4);      // u_x, u_y, u, pressure per cell, whole grid
  double tic = 0.0, toc = 0.0;

  if (ngpus <= 1) {
    lbm_ctx* ctx = nullptr;
    if (lbm_create(&ctx, &params, free_cells, obstacles.data(), 0, params.ny, env_int("LBM_DEVICE", 0), flags))
      die(lbm_last_error(), __LINE__, __FILE__);
    tic = wall_seconds();                                                  // :278-279
    if (lbm_run(ctx, params.max_iters, av_vels.data())) die(lbm_last_error(), __LINE__, __FILE__);
    toc = wall_seconds();                                                  // :397-398
    if (lbm_get_observables(ctx, obs.data())) die(lbm_last_error(), __LINE__, __FILE__);
    lbm_destroy(ctx);
  } else {
    // ---- one process, N GPUs: N ranks of the reference's decomposition, one host thread each ------------
    if (ngpus > 64) die("LBM_GPUS: at most 64 ranks (MPI_PROCS, d2q9-bgk.c:67)", __LINE__, __FILE__);
    std::vector<int> device(ngpus);
    {
      int ndev = env_int("LBM_DEVICE_COUNT", 0);
      const char* list = std::getenv("LBM_DEVICES");
      std::string rest = list ? list : "";
      for (int r = 0; r < ngpus; ++r) {
        if (!rest.empty()) {
          device[r] = std::atoi(rest.c_str());
          const size_t comma = rest.find(',');
          rest = comma == std::string::npos ? "" : rest.substr(comma + 1);
        } else {
          device[r] = ndev > 0 ? r % ndev : r;
        }
      }
    }
    {
      // Several ranks on ONE device (a test set-up; a node run has one rank per GPU): each rank's stream needs a
      // hardware queue of its own, or a rank's wait kernel can end up queued in front of the push it waits for.
      // The HIP runtime reads its queue budget from the environment when it initialises, which is still ahead.
      int most = 1;
      for (int r = 0; r < ngpus; ++r) {
        int same = 0;
        for (int q = 0; q < ngpus; ++q) same += device[q] == device[r];
        if (same > most) most = same;
      }
      if (most > 1) setenv("GPU_MAX_HW_QUEUES", std::to_string(2 * most + 4).c_str(), 0);
    }
    std::vector<lbm_layout> lay(ngpus);
    std::vector<lbm_ctx*> ctx(ngpus, nullptr);
    std::vector<lbm_p2p*> ring(ngpus, nullptr);
    std::vector<char> handles(static_cast<size_t>(ngpus) * LBM_P2P_HANDLE_BYTES);
    for (int r = 0; r < ngpus; ++r) {
      if (lbm_rank_layout(&params, ngpus, r, flags, &lay[r])) die(lbm_last_error(), __LINE__, __FILE__);
      // the rows this rank needs: its own plus `ghost` rows below and above, wrapping (the scatter of :968-970)
      const int rows = lay[r].ny_local + 2 * lay[r].ghost;
      std::vector<int> window(static_cast<size_t>(rows) * nx);
      for (int i = 0; i < rows; ++i) {
        int g = (lay[r].y0 - lay[r].ghost + i) % params.ny;
        if (g < 0) g += params.ny;
        std::memcpy(window.data() + static_cast<size_t>(i) * nx, obstacles.data() + static_cast<size_t>(g) * nx, sizeof(int) * nx);
      }
      if (lbm_create_rank(&ctx[r], &params, free_cells, window.data(), ngpus, r, device[r], flags)) die(lbm_last_error(), __LINE__, __FILE__);
      if (lbm_p2p_create(&ring[r], ctx[r], ngpus, r)) die(lbm_last_error(), __LINE__, __FILE__);
      if (lbm_p2p_handle(ring[r], handles.data() + static_cast<size_t>(r) * LBM_P2P_HANDLE_BYTES)) die(lbm_last_error(), __LINE__, __FILE__);
    }
    for (int r = 0; r < ngpus; ++r)
      if (lbm_p2p_connect(ring[r], handles.data())) die(lbm_last_error(), __LINE__, __FILE__);
    std::vector<std::vector<double>> tot(ngpus, std::vector<double>(static_cast<size_t>(params.max_iters) + 1));
    std::vector<std::string> failure(ngpus);
    tic = wall_seconds();                                                  // :278-279
    {
      std::vector<std::thread> pool;
      for (int r = 0; r < ngpus; ++r)
        pool.emplace_back([&, r]() {
          if (lbm_p2p_run(ring[r], params.max_iters, tot[r].data())) failure[r] = lbm_last_error();   // :315-396
        });
      for (std::thread& t : pool) t.join();
    }
    toc = wall_seconds();                                                  // :397-398
    for (int r = 0; r < ngpus; ++r)
      if (!failure[r].empty()) die(failure[r].c_str(), __LINE__, __FILE__);
    const float inv = 1.0f / free_cells;                                   // :950
    for (int t = 0; t < params.max_iters; ++t) av_vels[t] = static_cast<float>(tot[0][t] * static_cast<double>(inv));   // :367
    for (int r = 0; r < ngpus; ++r) {
      if (lbm_get_observables(ctx[r], obs.data() + static_cast<size_t>(lay[r].y0) * nx * 4)) die(lbm_last_error(), __LINE__, __FILE__);
      lbm_p2p_destroy(ring[r]);
    }
    for (int r = 0; r < ngpus; ++r) lbm_destroy(ctx[r]);
  }
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);                                             // :399-403
  const double usrtim = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1000000.0;
  const double systim = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1000000.0;

  const float free_cells_inv = 1.0f / free_cells;                          // :950
  const float av = lbm_av_velocity_obs(&params, obs.data(), obstacles.data(), params.ny) * free_cells_inv;   // :753
  std::printf("==done==\n");                                               // :411-415
  std::printf("Reynolds number:\t\t%.12E\n", lbm_reynolds(&params, av));
  std::printf("Elapsed time:\t\t\t%.6lf (s)\n", toc - tic);
  std::printf("Elapsed user CPU time:\t\t%.6lf (s)\n", usrtim);
  std::printf("Elapsed system CPU time:\t%.6lf (s)\n", systim);
  const double mlups = static_cast<double>(params.nx) * params.ny * params.max_iters / (toc - tic) / 1e6;
  std::printf("MLUPS:\t\t\t\t%.1f (%d GPU%s)\n", mlups, ngpus > 1 ? ngpus : 1, ngpus > 1 ? "s, peer-to-peer halos" : "");
  std::printf("HBM roofline (108 B/cell-step @ 8.0 TB/s = 74074 MLUPS per GPU):\t%.1f %%\n",
              100.0 * mlups / (ngpus > 1 ? ngpus : 1) / (8.0e12 / 108.0 / 1e6));

  if (!env_int("LBM_NO_OUTPUT", 0)) {                                      // :419-421
    if (lbm_write_final_state_obs("final_state.dat", &params, obs.data(), obstacles.data(), params.ny, 0, 0))
      die(lbm_last_error(), __LINE__, __FILE__);
    if (lbm_write_av_vels("av_vels.dat", av_vels.data(), params.max_iters)) die(lbm_last_error(), __LINE__, __FILE__);
  }
  return EXIT_SUCCESS;                                                     // :439
}

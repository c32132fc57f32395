// Where the cycles of biquad_pipe_kernel<5> go: the product kernel compiled with GA_BQ_PROBE (s_memtime around the phases of
// every tile, summed over the waves) on V cascades of 5 sections x N frames.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I graphaudio_amd/csrc tools/micro/bq_pipe_probe.hip -o tools/micro/bq_pipe_probe
//   tools/micro/bq_pipe_probe <cascades> <frames> <cascades per wave> [row stride in rows: 3 = the rows of a cascade's input and
//   output two rows apart, as where the engine's slab allocator interleaves them with other nodes' rows]
#define GA_BQ_PROBE 1
#include "../../graphaudio_amd/csrc/ga_kernels.hip"
#include <vector>
#include <cmath>
#include <cstring>
namespace ga { [[noreturn]] void launch_fail(const char* what) { printf("%s\n", what); abort(); } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  using namespace ga;
  const int V = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 120064, jpw = argc > 3 ? atoi(argv[3]) : 8, NS = 5;
  const size_t RS = (size_t)(argc > 4 ? atoi(argv[4]) : 1) * N;   // row stride
  float *in, *out, *state;
  BiquadSection* secs;
  BiquadJob* jobs;
  CK(hipMalloc(&in, (size_t)V * RS * 4));
  CK(hipMalloc(&out, (size_t)V * RS * 4));
  CK(hipMalloc(&state, (size_t)V * NS * 2 * 4));
  CK(hipMalloc(&secs, (size_t)V * NS * sizeof(BiquadSection)));
  CK(hipMalloc(&jobs, (size_t)V * sizeof(BiquadJob)));
  std::vector<float> h((size_t)V * N);
  unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 24)); }
  CK(hipMemcpy2D(in, RS * 4, h.data(), (size_t)N * 4, (size_t)N * 4, V, hipMemcpyHostToDevice));
  CK(hipMemset(state, 0, (size_t)V * NS * 2 * 4));
  std::vector<BiquadSection> hs((size_t)V * NS);
  std::vector<BiquadJob> hj(V);
  for (int v = 0; v < V; v++) {
    for (int q = 0; q < NS; q++) {
      BiquadSection& b = hs[(size_t)v * NS + q];
      const double w0 = 2 * M_PI * (200.0 * (q + 1) + v % 97) / 48000.0, al = sin(w0) / 1.4, A = 1.1;
      const double a0 = 1 + al / A;
      b.b0 = (float)((1 + al * A) / a0); b.b1 = (float)(-2 * cos(w0) / a0); b.b2 = (float)((1 - al * A) / a0);
      b.a1 = b.b1; b.a2 = (float)((1 - al / A) / a0); b.pad_ = 0; b.state = state + ((size_t)v * NS + q) * 2;
    }
    hj[v] = BiquadJob{in + (size_t)v * RS, out + (size_t)v * RS, v * NS, NS, 0, N, nullptr};
  }
  CK(hipMemcpy(secs, hs.data(), hs.size() * sizeof(BiquadSection), hipMemcpyHostToDevice));
  CK(hipMemcpy(jobs, hj.data(), hj.size() * sizeof(BiquadJob), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int waves = (V + jpw - 1) / jpw;
  for (int rep = 0; rep < 3; rep++) {
    unsigned long long zero[8] = {0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ga_bq_probe), zero, sizeof zero));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(biquad_pipe_kernel<5>, dim3(waves), dim3(64), 0, 0, jobs, V, secs, jpw);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long pr[8];
    CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(ga_bq_probe), sizeof pr));
    const double w = waves, tiles = (N + 16 + 255) / 256;
    printf("V %d N %d jpw %d waves %d: %.3f ms = %.1f ns per step | per wave (s_memtime ticks): total %.0f  stage-in %.0f  walk %.0f  stage-out %.0f"
           " | per tile: stage-in %.0f walk %.0f stage-out %.0f | batches per wave: steady %.0f masked %.0f\n",
           V, N, jpw, waves, ms, ms * 1e6 / N, pr[3] / w, pr[0] / w, pr[1] / w, pr[2] / w, pr[0] / w / tiles, pr[1] / w / tiles, pr[2] / w / tiles,
           pr[4] / w, pr[5] / w);
  }
  // checksum against the lane-per-cascade kernel (same arithmetic, one walk)
  std::vector<float> o1((size_t)V * N), o2((size_t)V * N);
  CK(hipMemset(state, 0, (size_t)V * NS * 2 * 4));
  launch_biquad_lanes(0, jobs, V, secs, NS);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy2D(o2.data(), (size_t)N * 4, out, RS * 4, (size_t)N * 4, V, hipMemcpyDeviceToHost));
  // (the timed launches ran 3 times from moving states: compare a fresh single run)
  CK(hipMemset(state, 0, (size_t)V * NS * 2 * 4));
  hipLaunchKernelGGL(biquad_pipe_kernel<5>, dim3(waves), dim3(64), 0, 0, jobs, V, secs, jpw);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy2D(o1.data(), (size_t)N * 4, out, RS * 4, (size_t)N * 4, V, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < o1.size(); i++) bad += memcmp(&o1[i], &o2[i], 4) != 0;
  printf("pipe vs lane kernel: %zu of %zu values differ\n", bad, o1.size());
  return bad != 0;
}

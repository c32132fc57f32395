// biquad_pipe_kernel<NSEC> against the lane-per-cascade kernel, bit for bit, for every cascade length and ragged sizes (frames that
// are no multiple of anything, fewer cascades than a wave holds, cascades of different lengths in one wave, non-zero start states).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I graphaudio_amd/csrc tools/micro/bq_pipe_check.hip -o tools/micro/bq_pipe_check
#include "../../graphaudio_amd/csrc/ga_kernels.hip"
#include <vector>
#include <cmath>
#include <cstring>
namespace ga { [[noreturn]] void launch_fail(const char* what) { printf("%s\n", what); abort(); } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  using namespace ga;
  const int NMAX = 5000, VMAX = 300;
  float *in, *out, *state;
  BiquadSection* secs;
  BiquadJob* jobs;
  CK(hipMalloc(&in, (size_t)VMAX * NMAX * 4));
  CK(hipMalloc(&out, (size_t)VMAX * NMAX * 4));
  CK(hipMalloc(&state, (size_t)VMAX * 8 * 2 * 4));
  CK(hipMalloc(&secs, (size_t)VMAX * 8 * sizeof(BiquadSection)));
  CK(hipMalloc(&jobs, (size_t)VMAX * sizeof(BiquadJob)));
  std::vector<float> h((size_t)VMAX * NMAX);
  unsigned s = 777;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 24)); }
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  size_t total_bad = 0;
  int cases = 0;
  for (int NS = 2; NS <= 8; NS++)
    for (int V : {1, 3, 13, 64, 300})
      for (int N : {1, 2, 5, 17, 127, 128, 129, 255, 256, 257, 300, 1000, 4097}) {
        std::vector<BiquadSection> hs((size_t)V * NS);
        std::vector<BiquadJob> hj(V);
        std::vector<float> st0((size_t)V * NS * 2);
        for (auto& x : st0) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) * (0.2f / (1 << 24)); }
        for (int v = 0; v < V; v++) {
          for (int q = 0; q < NS; q++) {
            BiquadSection& b = hs[(size_t)v * NS + q];
            const double w0 = 2 * M_PI * (150.0 * (q + 1) + 31 * (v % 97)) / 48000.0, al = sin(w0) / (1.0 + 0.3 * q), A = 1.0 + 0.05 * ((v + q) % 5);
            const double a0 = 1 + al / A;
            b.b0 = (float)((1 + al * A) / a0); b.b1 = (float)(-2 * cos(w0) / a0); b.b2 = (float)((1 - al * A) / a0);
            b.a1 = b.b1; b.a2 = (float)((1 - al / A) / a0); b.pad_ = 0; b.state = state + ((size_t)v * NS + q) * 2;
          }
          const int n = (v % 3 == 2) ? std::max(1, N - 7 * (v % 5)) : N;   // ragged lengths inside a wave
          hj[v] = BiquadJob{in + (size_t)v * NMAX + (v % 4), out + (size_t)v * NMAX + (v % 4), v * NS, NS, 0, n, nullptr};
        }
        CK(hipMemcpy(secs, hs.data(), hs.size() * sizeof(BiquadSection), hipMemcpyHostToDevice));
        CK(hipMemcpy(jobs, hj.data(), hj.size() * sizeof(BiquadJob), hipMemcpyHostToDevice));
        std::vector<float> o1((size_t)VMAX * NMAX), o2((size_t)VMAX * NMAX), s1(st0.size()), s2(st0.size());
        CK(hipMemset(out, 0, (size_t)VMAX * NMAX * 4));
        CK(hipMemcpy(state, st0.data(), st0.size() * 4, hipMemcpyHostToDevice));
        launch_biquad(0, jobs, V, secs, NS);   // the pipelined kernel
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o1.data(), out, o1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1.data(), state, s1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0, (size_t)VMAX * NMAX * 4));
        CK(hipMemcpy(state, st0.data(), st0.size() * 4, hipMemcpyHostToDevice));
        launch_biquad_lanes(0, jobs, V, secs, NS);   // lane per cascade
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o2.data(), out, o2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s2.data(), state, s2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < o1.size(); i++) bad += memcmp(&o1[i], &o2[i], 4) != 0;
        for (size_t i = 0; i < s1.size(); i++) bad += memcmp(&s1[i], &s2[i], 4) != 0;
        if (bad) printf("NSEC %d V %d N %d: %zu values differ\n", NS, V, N, bad);
        total_bad += bad;
        cases++;
      }
  printf("%d cases, %zu values differ\n", cases, total_bad);
  return total_bad != 0;
}

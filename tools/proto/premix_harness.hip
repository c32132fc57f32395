// The product's pre-mix kernel (graphaudio_amd/csrc/ga_coarse.hip, included as is) on synthetic job tables: is the kernel or its
// surroundings (allocation layout, the hand-over jobs riding along, the histories it writes) what separates it from the plain
// row-sum of tools/proto/hbm_peak.hip?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I graphaudio_amd/csrc -I include -o tools/proto/premix_harness tools/proto/premix_harness.hip
#include "../../graphaudio_amd/csrc/ga_coarse.hip"
#include <vector>
namespace ga {
[[noreturn]] void launch_fail(const char* what) {
  fprintf(stderr, "launch_fail: %s\n", what);
  exit(1);
}
}  // namespace ga
using namespace ga;
static hipEvent_t e0, e1;
template <class F>
static double timeit(F f, int reps = 7) {
  f();
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    float ms;
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}
__global__ void fill(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v + (float)(i & 1023) * 1e-3f;
}
int main() {
  const int rows = 1024;
  const int64_t frames = 480000, hl = 65536;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float* mixed;
  (void)hipMalloc(&mixed, (frames + hl) * 4);
  for (int layout = 0; layout < 3; layout++) {   // 0: one arena, natural stride ; 1: separate allocations ; 2: separate + 1 KB skew
    std::vector<void*> bases;
    std::vector<PremixTerm> terms(rows);
    const size_t rb = ((size_t)frames * 4 + 1023) / 1024 * 1024 + 1024;
    if (layout == 0) {
      void* arena;
      (void)hipMalloc(&arena, rb * rows);
      fill<<<4096, 256>>>((float*)arena, rb * rows / 4, 0.1f);
      bases.push_back(arena);
      for (int r = 0; r < rows; r++) terms[r].in = (const float*)((char*)arena + rb * r);
    } else {
      for (int r = 0; r < rows; r++) {
        void* b;
        (void)hipMalloc(&b, (size_t)frames * 4 + 65536);
        fill<<<256, 256>>>((float*)b, ((size_t)frames * 4 + 65536) / 4, 0.1f);
        bases.push_back(b);
        terms[r].in = (const float*)((char*)b + (layout == 2 ? (size_t)(r % 64) * 1024 : 0));
      }
    }
    float* hist;   // the members' next histories: [rows][hl]
    (void)hipMalloc(&hist, (size_t)rows * hl * 4);
    for (int carry = 0; carry < 2; carry++) {
      for (int r = 0; r < rows; r++) {
        terms[r].carry = carry ? hist + (size_t)r * hl : nullptr;
      }
      PremixJob job{mixed + hl, 0, rows, frames, frames - hl, 1 | (carry ? 2 : 0), 0};
      PremixTerm* dterms;
      PremixJob* djob;
      (void)hipMalloc(&dterms, rows * sizeof(PremixTerm));
      (void)hipMalloc(&djob, sizeof(PremixJob));
      (void)hipMemcpy(dterms, terms.data(), rows * sizeof(PremixTerm), hipMemcpyHostToDevice);
      (void)hipMemcpy(djob, &job, sizeof(job), hipMemcpyHostToDevice);
      const double ms = timeit([&] { launch_coarse_premix(nullptr, djob, 1, dterms, frames); });
      const double bytes = (double)rows * frames * 4 + (carry ? (double)rows * hl * 4 : 0.0) + frames * 4.0;
      printf("layout %d (%s), histories %s: %.4f ms  %.2f TB/s\n", layout,
             layout == 0 ? "one arena" : layout == 1 ? "separate allocations" : "separate allocations, 1 KB skew", carry ? "written" : "not written", ms,
             bytes / (ms * 1e9));
      (void)hipFree(dterms);
      (void)hipFree(djob);
    }
    (void)hipFree(hist);
    for (void* b : bases) (void)hipFree(b);
  }
  return 0;
}

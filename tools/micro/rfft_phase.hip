// Phase timeline of the B-layout rFFT kernels (forward: planes <- signal ; inverse: signal <- planes), standalone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGA_EXP_TIMELINE tools/micro/rfft_phase.hip -o rfft_phase
// Prints the kernel time (HIP events) and, per phase, the mean shader-clock cycles seen by thread 0 of every workgroup.
#include "../../graphaudio_amd/csrc/ga_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  using namespace ga;
  const int rows = argc > 1 ? atoi(argv[1]) : 256, nblocks = argc > 2 ? atoi(argv[2]) : 3750, hist = 512;
  const int tx = hist + ((nblocks + 15) / 16) * 16 + 16, ty = ((nblocks + 255) / 256) * 256;
  float *xr, *xi, *yr, *yi, *sig, *out, *ov;
  CK(hipMalloc(&xr, (size_t)rows * kBins * tx * 4)); CK(hipMalloc(&xi, (size_t)rows * kBins * tx * 4));
  CK(hipMalloc(&yr, (size_t)rows * kBins * ty * 4)); CK(hipMalloc(&yi, (size_t)rows * kBins * ty * 4));
  CK(hipMalloc(&sig, (size_t)rows * nblocks * kBlock * 4)); CK(hipMalloc(&out, (size_t)rows * nblocks * kBlock * 4));
  CK(hipMalloc(&ov, (size_t)rows * 2 * kBlock * 4));
  CK(hipMemset(sig, 0, (size_t)rows * nblocks * kBlock * 4)); CK(hipMemset(yr, 0, (size_t)rows * kBins * ty * 4));
  CK(hipMemset(yi, 0, (size_t)rows * kBins * ty * 4)); CK(hipMemset(ov, 0, (size_t)rows * 2 * kBlock * 4));
  std::vector<ConvRowIO> xio(rows), yio(rows);
  std::vector<const float*> ovin(rows);
  std::vector<float*> ovout(rows);
  for (int r = 0; r < rows; r++) {
    xio[r] = ConvRowIO{sig + (size_t)r * nblocks * kBlock, nullptr};
    yio[r] = ConvRowIO{nullptr, out + (size_t)r * nblocks * kBlock};
    ovin[r] = ov + (size_t)r * 2 * kBlock; ovout[r] = ov + (size_t)r * 2 * kBlock + kBlock;
  }
  ConvRowIO *dx, *dy; const float** doi; float** doo;
  CK(hipMalloc(&dx, rows * sizeof(ConvRowIO))); CK(hipMalloc(&dy, rows * sizeof(ConvRowIO)));
  CK(hipMalloc(&doi, rows * 8)); CK(hipMalloc(&doo, rows * 8));
  CK(hipMemcpy(dx, xio.data(), rows * sizeof(ConvRowIO), hipMemcpyHostToDevice)); CK(hipMemcpy(dy, yio.data(), rows * sizeof(ConvRowIO), hipMemcpyHostToDevice));
  CK(hipMemcpy(doi, ovin.data(), rows * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(doo, ovout.data(), rows * 8, hipMemcpyHostToDevice));
  std::vector<double2> w128(64), w256(129);
  for (int j = 0; j < 64; j++) w128[j] = double2{cos(-2 * M_PI * j / 128), sin(-2 * M_PI * j / 128)};
  for (int k = 0; k <= 128; k++) w256[k] = double2{cos(-2 * M_PI * k / 256), sin(-2 * M_PI * k / 256)};
  double2 *d128, *d256;
  CK(hipMalloc(&d128, 64 * 16)); CK(hipMalloc(&d256, 129 * 16));
  CK(hipMemcpy(d128, w128.data(), 64 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d256, w256.data(), 129 * 16, hipMemcpyHostToDevice));
  Twiddles tw{d128, d256};
  ConvPlanesB pl{xr, xi, yr, yi, tx, ty};
  const int runs = (nblocks + 31) / 32;
  const size_t nwg = (size_t)runs * rows;
  unsigned long long* tl;
  CK(hipMalloc(&tl, nwg * 8 * 8));
  CK(hipMemset(tl, 0, nwg * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ga::ga_tl), &tl, sizeof(tl)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> h(nwg * 8);
  for (int which = 0; which < 2; which++) {
    const int nph = which == 0 ? 3 : 6;
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0, 0));
      if (which == 0) launch_rfft_fwd_b(0, dx, rows, nblocks, hist, pl, tw, false);
      else launch_irfft_ola_b(0, dy, rows, nblocks, pl, doi, doo, tw, false);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    CK(hipMemcpy(h.data(), tl, nwg * 64, hipMemcpyDeviceToHost));
    std::vector<double> acc(nph, 0.0);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < nwg; w++) {
      for (int p = 0; p < nph; p++) acc[p] += (double)(h[w * 8 + p + 1] - h[w * 8 + p]);
      tmin = std::min(tmin, h[w * 8]); tmax = std::max(tmax, h[w * 8 + nph]);
    }
    const double bytes = which == 0 ? (double)rows * nblocks * (kBlock * 4 + kBins * 8) : (double)rows * nblocks * (kBlock * 4 + kBins * 8);
    printf("%s: %.3f ms  %.2f TB/s  span %.0f kcycles (memtime), per-WG mean cycles by phase:", which == 0 ? "fwd" : "inv", best, bytes / best / 1e9, (tmax - tmin) / 1e3);
    double tot = 0;
    for (int p = 0; p < nph; p++) { printf(" %.0f", acc[p] / nwg); tot += acc[p] / nwg; }
    printf("  total %.0f ; WGs %zu\n", tot, nwg);
  }
  return 0;
}

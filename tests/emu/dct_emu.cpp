// dct_emu.cpp -- TEST INFRASTRUCTURE ONLY: runs the device source of the fast cosine transforms
// (adaflo_amd/csrc/fdm_dct_kernel.hpp) on the host lane emulator (hip_emu.hpp) so that tests/test_dct_emulated.py can
// compare it with the plain cosine sums without a GPU.  The product never loads this library.
#include "hip_emu.hpp"

#include <cmath>
#include <vector>

#include "../../adaflo_amd/csrc/fdm_dct_kernel.hpp"

using namespace adaflo_hip::dct;

namespace
{
  template <int N>
  int run(const DctArgs &A, const int fused)
  {
    using G = Geo<N>;
    static_assert(G::L_TOTAL * 8 <= (int)emu::LDS_BYTES, "LDS");
    long nb = (A.n_lines + G::LB - 1) / G::LB;
    nb      = nb > 3 ? 3 : nb; // (a small persistent grid: every workgroup loops over several batches)
    double *lds = reinterpret_cast<double *>(emu::g_lds);
    if (fused)
      emu::launch((unsigned)nb, NT, [&] { dct_body<N, true, 2>(A, lds); });
    else if (A.axis == 0)
      emu::launch((unsigned)nb, NT, [&] { dct_body<N, false, 0>(A, lds); });
    else if (A.axis == 1)
      emu::launch((unsigned)nb, NT, [&] { dct_body<N, false, 1>(A, lds); });
    else
      emu::launch((unsigned)nb, NT, [&] { dct_body<N, false, 2>(A, lds); });
    return 0;
  }
} // namespace

// out = cosine sums of `in` ([z][y][x]) along `axis`; fused (axis 2 only): forward, scaling, back
// (pitch: elements per x-row of both arrays, >= nx)
extern "C" int dct_emu_apply(const int axis, const int fused, const int nx, const int ny, const int nz, const int pitch, const double *in,
                             double *out, const double *lx, const double *ly, const double *lz, const double *ax,
                             const double *ay, const double *az, const double cm, const double cl, const double eps)
{
  const int n = axis == 0 ? nx : (axis == 1 ? ny : nz), N = n - 1;
  if (!dct_length_supported(N) || (fused && axis != 2))
    return 1;
  std::vector<double> tw(2 * (N + 1));
  for (int m = 0; m <= N; ++m)
    {
      tw[2 * m]     = std::cos(M_PI * m / N);
      tw[2 * m + 1] = -std::sin(M_PI * m / N);
    }
  DctArgs A{};
  A.in = in, A.out = out, A.tw = tw.data();
  A.n_lines = axis == 0 ? (long)ny * nz : (axis == 1 ? (long)pitch * nz : (long)pitch * ny);
  A.axis = axis, A.nx = nx, A.ny = ny, A.nz = nz, A.pitch_in = A.pitch_out = pitch;
  A.lx = lx, A.ly = ly, A.lz = lz, A.ax = ax, A.ay = ay, A.az = az, A.cm = cm, A.cl = cl, A.eps = eps;
  switch (N)
    {
      case 64:
        return run<64>(A, fused);
      case 128:
        return run<128>(A, fused);
      case 256:
        return run<256>(A, fused);
      case 512:
        return run<512>(A, fused);
      case 1024:
        return run<1024>(A, fused);
      case 80:
        return run<80>(A, fused);
      case 160:
        return run<160>(A, fused);
      case 320:
        return run<320>(A, fused);
      case 640:
        return run<640>(A, fused);
      case 96:
        return run<96>(A, fused);
      case 192:
        return run<192>(A, fused);
      case 384:
        return run<384>(A, fused);
      case 768:
        return run<768>(A, fused);
    }
  return 1;
}

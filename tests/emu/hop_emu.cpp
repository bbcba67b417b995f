// hop_emu.cpp -- TEST INFRASTRUCTURE ONLY: runs the device source of the plane-per-lane Q4/Q3 kernel
// (adaflo_amd/csrc/ns_hop_kernel.hpp: state conversion, main kernel; seam fix-up of ns_hox_kernel.hpp) on the host lane
// emulator (hip_emu.hpp) so that tests/test_hop_emulated.py can compare it with the oracle without a GPU.  The product
// never loads this library.
#include "hip_emu.hpp"

#include <type_traits>
#include <vector>

#include "../../adaflo_amd/csrc/basis.hpp"
#include "../../adaflo_amd/csrc/ns_hox_kernel.hpp"
#include "../../adaflo_amd/csrc/ns_hop_kernel.hpp"

using namespace adaflo_hip;
using namespace adaflo_hip::hop;

namespace
{
  template <int LM, bool WP>
  void run_main(const hox::HXArgs &A, const long ntiles)
  {
    if (ntiles > 0)
      emu::launch((unsigned)((ntiles + NW - 1) / NW), NTH, [&] { ns_hop_kernel<LM, WP>(A); });
  }
} // namespace

// coef: cA, cB, beta, tau_gd, tmu
extern "C" int hop_emu_vmult(const int *ncell, const double *h, const int op, const int lin_mode, const int integrate_p,
                             const double *coef, const unsigned con_u, const unsigned con_p, const double *lin_generic,
                             const double *src_u, const double *src_p, double *dst_u, double *dst_p, const int lx,
                             const unsigned iface, const int phased)
{
  hox::HXArgs A{};
  hop_geometry(A, ncell, lx);
  std::vector<double> tab;
  {
    const Quadrature1D        qu = gauss(N);
    const Shape1D             su = shape_fe_q(K, qu), sp = shape_fe_q(K - 1, qu);
    const std::vector<double> dc = collocation_derivative(qu);
    tab = hop_table(su.S.data(), dc.data(), sp.S.data(), qu.w.data(), h, coef[0], coef[1], coef[2], coef[3], coef[4]);
  }
  A.tab         = tab.data();
  A.integrate_p = integrate_p;
  A.con_u       = con_u;
  A.con_p       = con_p;
  A.src_u       = src_u;
  A.src_p       = src_p;
  A.dst_u       = dst_u;
  A.dst_p       = dst_p;
  std::vector<double> state((size_t)A.ngz * A.ngy * A.ncx * state_cell_doubles(lin_mode) + 2);
  if (npc_of(lin_mode) > 0)
    emu::launch(4, 256, [&] { hop_convert_state_kernel(state.data(), lin_generic, A.ncx, A.ncy, A.ncz, A.ngy, A.ngz, lin_mode); });
  A.lin = state.data();
  const size_t        n_t = (size_t)A.tiles_y * A.tiles_z * A.n_chunks;
  const double        nan = std::nan("");
  std::vector<double> slab_u(n_t * PGeo::RIMU * (K * A.LX + 1) * 3, nan), xslab_u(n_t * PGeo::TNY * PGeo::TNZ * 3, nan),
    slab_p(n_t * PGeo::RIMP * (KP * A.LX + 1), nan), xslab_p(n_t * PGeo::TPY * PGeo::TPZ, nan);
  A.slab_u  = slab_u.data();
  A.xslab_u = xslab_u.data();
  A.slab_p  = slab_p.data();
  A.xslab_p = xslab_p.data();
  const bool with_p = op == 0;
  auto       main_k = [&](const long nt) {
    if (with_p)
      {
        if (lin_mode == 0)
          run_main<0, true>(A, nt);
        else if (lin_mode == 1)
          run_main<1, true>(A, nt);
        else
          run_main<2, true>(A, nt);
      }
    else
      {
        if (lin_mode == 0)
          run_main<0, false>(A, nt);
        else if (lin_mode == 1)
          run_main<1, false>(A, nt);
        else
          run_main<2, false>(A, nt);
      }
  };
  const bool fix_p = with_p && integrate_p;
  auto       fixup = [&] {
    if (hox::hox_fix_blocks(A, fix_p) > 0)
      emu::launch(3, 256, [&] { hox::ns_hox_fixup_kernel<K, PGeo>(A, fix_p ? 1 : 0); });
  };
  if (!phased)
    {
      main_k((long)n_t);
      fixup();
    }
  else
    {
      std::vector<int> list;
      int              counts[3];
      hox::hox_wg_lists(A, iface, list, counts);
      list.push_back(0);
      A.wg_list = list.data();
      A.iface   = iface;
      for (const int phase : {0, 1, 2})
        {
          A.wg_offset = phase == 1 ? 0 : (phase == 0 ? counts[0] : counts[0] + counts[1]);
          A.wg_count  = phase == 1 ? counts[0] : (phase == 0 ? counts[1] : counts[2]);
          A.fix_mode  = phase;
          main_k(A.wg_count);
          if (phase > 0)
            fixup();
        }
    }
  return 0;
}

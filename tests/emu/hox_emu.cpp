// hox_emu.cpp -- TEST INFRASTRUCTURE ONLY: runs the device source of the x-marching Q3..Q5 kernel
// (adaflo_amd/csrc/ns_hox_kernel.hpp: state conversion, main kernel, seam fix-up) on the host lane emulator
// (hip_emu.hpp) so that tests/test_hox_emulated.py can compare it with the oracle without a GPU.  The product
// never loads this library.
#include "hip_emu.hpp"

#include <type_traits>
#include <vector>

#include "../../adaflo_amd/csrc/basis.hpp"
#include "../../adaflo_amd/csrc/ns_hox_kernel.hpp"

using namespace adaflo_hip;
using namespace adaflo_hip::hox;

namespace
{
  template <int K, int LM, bool WP, bool VARCO = false, bool RCP = false>
  void run_main(const HXArgs &A, const long nwg)
  {
    if (nwg > 0)
      emu::launch((unsigned)nwg, NTH, [&] { ns_hox_kernel<K, LM, WP, false, VARCO, RCP>(A); });
  }

  template <int K>
  int run(const int *ncell, const double *h, const int op, const int lin_mode, const int integrate_p, const double *coef,
          const unsigned con_u, const unsigned con_p, const double *lin_generic, const double *src_u, const double *src_p,
          double *dst_u, double *dst_p, const int lx, const unsigned iface, const int phased, const double *rho = nullptr,
          const double *mu = nullptr, const double *damp = nullptr, const double *lin_nodal = nullptr)
  {
    using G         = Geo<K>;
    constexpr int N = K + 1;
    HXArgs        A{};
    hox_geometry<K>(A, ncell, lx);
    std::vector<double> tab;
    {
      const Quadrature1D        qu = gauss(N);
      const Shape1D             su = shape_fe_q(K, qu), sp = shape_fe_q(K - 1, qu);
      const std::vector<double> dc = collocation_derivative(qu);
      tab = hox_table<K>(su.S.data(), dc.data(), sp.S.data(), qu.w.data(), h, coef[0], coef[1], coef[2], coef[3], coef[4],
                         coef[5], coef[6], coef[7], coef[8]);
    }
    A.tab         = tab.data();
    A.integrate_p = integrate_p;
    A.con_u       = con_u;
    A.con_p       = con_p;
    A.src_u       = src_u;
    A.src_p       = src_p;
    A.dst_u       = dst_u;
    A.dst_p       = dst_p;
    const bool          varco = rho != nullptr;
    const int           npl = nst_of(lin_mode) / 2, npc = npl + (varco ? 2 : 0);
    std::vector<double> state((size_t)A.ngz * A.ngy * A.ncx * N * npc * G::CPW * G::NL * 2 + 2);
    if (npc > 0 && !lin_nodal)
      emu::launch(4, 256, [&] {
        hox_convert_state_kernel<K>(state.data(), lin_generic, A.ncx, A.ncy, A.ncz, A.ngy, A.ngz, npc, varco ? npl : -1, rho, mu, damp);
      });
    A.lin   = lin_nodal ? nullptr : state.data();
    A.lin_u = lin_nodal; // recompute-state mode: the kernel interpolates the nodal linearisation point itself
    const size_t        n_wg = (size_t)A.tiles_y * A.tiles_z * A.n_chunks;
    const double        nan  = std::nan("");
    std::vector<double> slab_u(n_wg * G::RIMU * (K * A.LX + 1) * 3, nan), xslab_u(n_wg * G::TNY * G::TNZ * 3, nan),
      slab_p(n_wg * G::RIMP * ((K - 1) * A.LX + 1), nan), xslab_p(n_wg * G::TPY * G::TPZ, nan);
    A.slab_u  = slab_u.data();
    A.xslab_u = xslab_u.data();
    A.slab_p  = slab_p.data();
    A.xslab_p = xslab_p.data();
    const bool with_p = op == 0;
    auto       main_k = [&](const long nwg) {
      if (varco)
        {
          if (with_p)
            {
              if (lin_mode == 0)
                run_main<K, 0, true, true>(A, nwg);
              else if (lin_mode == 1)
                run_main<K, 1, true, true>(A, nwg);
              else
                run_main<K, 2, true, true>(A, nwg);
            }
          else
            {
              if (lin_mode == 0)
                run_main<K, 0, false, true>(A, nwg);
              else if (lin_mode == 1)
                run_main<K, 1, false, true>(A, nwg);
              else
                run_main<K, 2, false, true>(A, nwg);
            }
          return;
        }
      if (lin_nodal)
        {
          if (with_p)
            {
              if (lin_mode == 0)
                run_main<K, 0, true, false, true>(A, nwg);
              else
                run_main<K, 1, true, false, true>(A, nwg);
            }
          else
            {
              if (lin_mode == 0)
                run_main<K, 0, false, false, true>(A, nwg);
              else
                run_main<K, 1, false, false, true>(A, nwg);
            }
          return;
        }
      if (with_p)
        {
          if (lin_mode == 0)
            run_main<K, 0, true>(A, nwg);
          else if (lin_mode == 1)
            run_main<K, 1, true>(A, nwg);
          else
            run_main<K, 2, true>(A, nwg);
        }
      else
        {
          if (lin_mode == 0)
            run_main<K, 0, false>(A, nwg);
          else if (lin_mode == 1)
            run_main<K, 1, false>(A, nwg);
          else
            run_main<K, 2, false>(A, nwg);
        }
    };
    const bool fix_p = with_p && integrate_p;
    auto       fixup = [&] {
      if (hox_fix_blocks(A, fix_p) > 0)
        emu::launch(3, 256, [&] { ns_hox_fixup_kernel<K>(A, fix_p ? 1 : 0); });
    };
    if (!phased)
      {
        main_k((long)n_wg);
        fixup();
      }
    else
      {
        std::vector<int> list;
        int              counts[3];
        hox_wg_lists(A, iface, list, counts);
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
} // namespace

namespace
{
  // residual mode: cell-loop sums, and the state the kernel wrote (converted back to the generic layout)
  template <int K>
  int run_residual(const int *ncell, const double *h, const int lin_mode, const double *coef, const double c_old,
                   const unsigned con_u, const unsigned con_p, const double *src_u, const double *src_p,
                   const double *old_comb, double *sum_u, double *sum_p, double *lin_generic, const int lx,
                   const double *ext_comb = nullptr, const double *rho = nullptr, const double *mu = nullptr,
                   const double *damp = nullptr)
  {
    using G         = Geo<K>;
    constexpr int N = K + 1;
    HXArgs        A{};
    hox_geometry<K>(A, ncell, lx);
    std::vector<double> tab;
    {
      const Quadrature1D        qu = gauss(N);
      const Shape1D             su = shape_fe_q(K, qu), sp = shape_fe_q(K - 1, qu);
      const std::vector<double> dc = collocation_derivative(qu);
      if (rho) // (variable coefficients: the table carries gamma, tau1, 1, tau1 as for the two-phase vmult: coef[5..8])
        tab = hox_table<K>(su.S.data(), dc.data(), sp.S.data(), qu.w.data(), h, coef[0], coef[1], coef[2], coef[3], coef[4],
                           coef[5], coef[6], coef[7], coef[8]);
      else
        tab = hox_table<K>(su.S.data(), dc.data(), sp.S.data(), qu.w.data(), h, coef[0], coef[1], coef[2], coef[3], coef[4]);
    }
    A.tab         = tab.data();
    A.integrate_p = 1;
    A.con_u       = con_u;
    A.con_p       = con_p;
    A.src_u       = src_u;
    A.src_p       = src_p;
    A.dst_u       = sum_u;
    A.dst_p       = sum_p;
    A.old_u       = old_comb;
    A.c_old       = c_old;
    A.lin_u       = ext_comb; // (extrapolating schemes, template EXT)
    const int           npl = nst_of(lin_mode) / 2, npc = npl + (rho ? 2 : 0);
    const size_t        n_state = (size_t)A.ngz * A.ngy * A.ncx * N * npc * G::CPW * G::NL * 2;
    std::vector<double> state(n_state + (size_t)N * npc * G::CPW * G::NL * 2 + 2, std::nan("")); // (+ the sink of cells beyond the mesh)
    A.lin_out  = state.data();
    A.lin_sink = state.data() + n_state;
    // variable-coefficient residual (RES + VARCO): the coefficient stream, two pieces per point
    std::vector<double> coef_stream;
    if (rho)
      {
        coef_stream.assign((size_t)A.ngz * A.ngy * A.ncx * N * 2 * G::CPW * G::NL * 2 + 2, std::nan(""));
        emu::launch(4, 256, [&] {
          hox_convert_state_kernel<K>(coef_stream.data(), nullptr, A.ncx, A.ncy, A.ncz, A.ngy, A.ngz, 2, 0, rho, mu, damp);
        });
        A.lin = coef_stream.data();
      }
    const size_t        n_wg = (size_t)A.tiles_y * A.tiles_z * A.n_chunks;
    const double        nan  = std::nan("");
    std::vector<double> slab_u(n_wg * G::RIMU * (K * A.LX + 1) * 3, nan), xslab_u(n_wg * G::TNY * G::TNZ * 3, nan),
      slab_p(n_wg * G::RIMP * ((K - 1) * A.LX + 1), nan), xslab_p(n_wg * G::TPY * G::TPZ, nan);
    A.slab_u  = slab_u.data();
    A.xslab_u = xslab_u.data();
    A.slab_p  = slab_p.data();
    A.xslab_p = xslab_p.data();
    if (rho && lin_mode == 0)
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 0, true, true, true>(A); });
    else if (rho && lin_mode == 1)
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 1, true, true, true>(A); });
    else if (rho)
      return -3;
    else if (ext_comb && lin_mode == 1)
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 1, true, true, false, false, true>(A); });
    else if (ext_comb)
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 2, true, true, false, false, true>(A); });
    else if (lin_mode == 0)
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 0, true, true>(A); });
    else if (lin_mode == 1)
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 1, true, true>(A); });
    else
      emu::launch((unsigned)n_wg, NTH, [&] { ns_hox_kernel<K, 2, true, true>(A); });
    if (hox_fix_blocks(A, true) > 0)
      emu::launch(3, 256, [&] { ns_hox_fixup_kernel<K>(A, 1); });
    if (npc > 0)
      emu::launch(4, 256, [&] { hox_unconvert_state_kernel<K>(lin_generic, state.data(), A.ncx, A.ncy, A.ncz, A.ngy, A.ngz, npc, npl); });
    if (rho) // the coefficient pieces behind the state pieces: what the two-phase vmult will stream
      {
        const long total = (long)A.ngz * A.ngy * A.ncx * N * npc * G::CPW * G::NL;
        for (long it = 0; it < total; ++it)
          {
            const int piece = (int)((it / (G::NL * G::CPW)) % npc);
            if (piece < npl)
              continue;
            const long pt = it / ((long)G::NL * G::CPW * npc), in_piece = it % ((long)G::NL * G::CPW);
            // (cells of a partial group beyond the mesh store nothing: the decode of hox_convert_state_kernel)
            const int  scw = (int)((it / G::NL) % G::CPW);
            const long grp = pt / ((long)N * A.ncx);
            const int  gy = (int)(grp % A.ngy), gz = (int)(grp / A.ngy);
            if (gy * G::CWY + scw % G::CWY >= A.ncy || gz * G::CWZ + scw / G::CWY >= A.ncz)
              continue;
            const long src = (pt * 2 + (piece - npl)) * (long)G::NL * G::CPW + in_piece;
            const double a0 = state[2 * it], a1 = state[2 * it + 1], b0 = coef_stream[2 * src], b1 = coef_stream[2 * src + 1];
            if (!(a0 == b0 && a1 == b1))
              return -4;
          }
      }
    return 0;
  }
} // namespace

extern "C" int hox_emu_residual(const int K, const int *ncell, const double *h, const int lin_mode, const double *coef,
                                const double c_old, const unsigned con_u, const unsigned con_p, const double *src_u,
                                const double *src_p, const double *old_comb, double *sum_u, double *sum_p,
                                double *lin_generic, const int lx)
{
  switch (K)
    {
      case 3:
        return run_residual<3>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx);
      case 4:
        return run_residual<4>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx);
      case 5:
        return run_residual<5>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx);
    }
  return -1;
}

// variable-coefficient residual (templates RES + VARCO, round 6): coef[0..8] as for the two-phase vmult, c_old = 1 or 0
extern "C" int hox_emu_residual_varco(const int K, const int *ncell, const double *h, const int lin_mode, const double *coef,
                                      const double c_old, const unsigned con_u, const unsigned con_p, const double *src_u,
                                      const double *src_p, const double *old_comb, const double *rho, const double *mu,
                                      const double *damp, double *sum_u, double *sum_p, double *lin_generic, const int lx)
{
  if (!rho || !mu || !damp)
    return -2;
  switch (K)
    {
      case 3:
        return run_residual<3>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx, nullptr, rho, mu, damp);
      case 4:
        return run_residual<4>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx, nullptr, rho, mu, damp);
      case 5:
        return run_residual<5>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx, nullptr, rho, mu, damp);
    }
  return -1;
}

extern "C" int hox_emu_vmult(const int K, const int *ncell, const double *h, const int op, const int lin_mode,
                             const int integrate_p, const double *coef, const unsigned con_u, const unsigned con_p,
                             const double *lin_generic, const double *src_u, const double *src_p, double *dst_u,
                             double *dst_p, const int lx, const unsigned iface, const int phased, const double *rho,
                             const double *mu, const double *damp)
{
  switch (K)
    {
      case 3:
        return run<3>(ncell, h, op, lin_mode, integrate_p, coef, con_u, con_p, lin_generic, src_u, src_p, dst_u, dst_p, lx, iface, phased, rho, mu, damp);
      case 4:
        return run<4>(ncell, h, op, lin_mode, integrate_p, coef, con_u, con_p, lin_generic, src_u, src_p, dst_u, dst_p, lx, iface, phased, rho, mu, damp);
      case 5:
        return run<5>(ncell, h, op, lin_mode, integrate_p, coef, con_u, con_p, lin_generic, src_u, src_p, dst_u, dst_p, lx, iface, phased, rho, mu, damp);
    }
  return -1;
}

// recompute-state mode (template RCP): the state is the interpolation of the nodal field `lin_nodal` (lin_mode 0 or 1)
extern "C" int hox_emu_vmult_recompute(const int K, const int *ncell, const double *h, const int op, const int lin_mode,
                                       const int integrate_p, const double *coef, const unsigned con_u, const unsigned con_p,
                                       const double *lin_nodal, const double *src_u, const double *src_p, double *dst_u,
                                       double *dst_p, const int lx, const unsigned iface, const int phased)
{
  if (lin_mode != 0 && lin_mode != 1)
    return -2;
  switch (K)
    {
      case 3:
        return run<3>(ncell, h, op, lin_mode, integrate_p, coef, con_u, con_p, nullptr, src_u, src_p, dst_u, dst_p, lx, iface, phased, nullptr, nullptr, nullptr, lin_nodal);
      case 4:
        return run<4>(ncell, h, op, lin_mode, integrate_p, coef, con_u, con_p, nullptr, src_u, src_p, dst_u, dst_p, lx, iface, phased, nullptr, nullptr, nullptr, lin_nodal);
      case 5:
        return run<5>(ncell, h, op, lin_mode, integrate_p, coef, con_u, con_p, nullptr, src_u, src_p, dst_u, dst_p, lx, iface, phased, nullptr, nullptr, nullptr, lin_nodal);
    }
  return -1;
}

// residual of the schemes that linearise about the extrapolated old velocity (template EXT): ext_comb = extrap_old u_old
// + extrap_old_old u_old_old at the nodes; lin_mode 1 (semi-implicit: stores (u_ext, div u_ext)) or 2 (explicit)
extern "C" int hox_emu_residual_extrapolated(const int K, const int *ncell, const double *h, const int lin_mode, const double *coef,
                                             const double c_old, const unsigned con_u, const unsigned con_p, const double *src_u,
                                             const double *src_p, const double *old_comb, const double *ext_comb, double *sum_u,
                                             double *sum_p, double *lin_generic, const int lx)
{
  if (!ext_comb || (lin_mode != 1 && lin_mode != 2))
    return -2;
  switch (K)
    {
      case 3:
        return run_residual<3>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx, ext_comb);
      case 4:
        return run_residual<4>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx, ext_comb);
      case 5:
        return run_residual<5>(ncell, h, lin_mode, coef, c_old, con_u, con_p, src_u, src_p, old_comb, sum_u, sum_p, lin_generic, lx, ext_comb);
    }
  return -1;
}

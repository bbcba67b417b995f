// comm.hip -- the ghost-DoF exchange of a brick partition behind the C ABI.
//
// Replaces what MatrixFree::cell_loop does for a LinearAlgebra::distributed::Vector across MPI
// ranks (source/navier_stokes_matrix.cc:232-245: update_ghost_values of src, local cells,
// compress(add) of dst) plus the one-double all-reduce of the pressure-mean projection (:201):
//   * one process per GPU, px x py x pz bricks, interface nodes replicated on the sharing ranks
//     (lowest grid coordinates own);
//   * per exchange ONE message per neighbour (faces, edges, corners; both fields packed back to
//     back), posted as a single ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a second
//     stream -- point-to-point over the xGMI links, no ring collective;
//   * overlap with the three phases of the Q2/Q1 sweep kernel (adaflo_ns_vmult_phase) through
//     HIP events, no host synchronisation anywhere in adaflo_ns_vmult_distributed.
// The transport is a small function table: RCCL (resolved with dlopen at adaflo_comm_create, so
// that single-GPU users never load it) or two caller-supplied callbacks (adaflo_comm_create_custom:
// an MPI-aware application, or the gloo-staged tests that run N ranks on one GPU).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"

using namespace adaflo_hip;

namespace
{
  // ---- RCCL through dlopen (types as in rccl.h; only what is called) ---------------------------
  struct RcclApi
  {
    void *lib = nullptr;
    int (*GetUniqueId)(void *)                                                       = nullptr;
    int (*CommInitRank)(void **, int, adaflo_comm_unique_id, int)                    = nullptr;
    int (*CommDestroy)(void *)                                                       = nullptr;
    int (*GroupStart)()                                                              = nullptr;
    int (*GroupEnd)()                                                                = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t)                 = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t)                       = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t)    = nullptr;
    const char *(*GetErrorString)(int)                                               = nullptr;
  };
  constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0; // ncclFloat64, ncclSum (rccl.h:448,467)

  RcclApi &rccl()
  {
    static RcclApi api;
    return api;
  }

  std::string load_rccl()
  {
    RcclApi &a = rccl();
    if (a.lib)
      return "";
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
        break;
    if (!a.lib)
      return std::string("cannot load RCCL: ") + dlerror();
#define SYM(field, sym)                                                    \
  if (!(*(void **)(&a.field) = dlsym(a.lib, sym)))                         \
    {                                                                      \
      a.lib = nullptr;                                                     \
      return std::string("RCCL symbol missing: ") + sym;                   \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return "";
  }

  struct Neighbour
  {
    int o[3];
    int rank;
    int cls; // number of non-zero offsets: 1 face, 2 edge, 3 corner
  };

  struct Field
  {
    int degree, ncomp, nn[3];
  };

  // one exchange round (who sends what where), built once
  struct Round
  {
    std::vector<int>  send_nb, recv_nb;     // indices into nbrs
    std::vector<long> send_off, send_cnt;   // per send neighbour: start / doubles in sbuf (all fields)
    std::vector<long> recv_off, recv_cnt;
    std::vector<int>  send_peer, recv_peer;
    // per field: halo plans over the send / recv regions (recv split by class for the copy mode)
    HaloPlan pack[2];
    HaloPlan unpack_add[2];
    HaloPlan unpack_copy[2][3];
    HaloPlan unpack_copy_all[2]; // all classes, ordered faces, edges, corners (one launch, "last region wins")
    long     n_send = 0, n_recv = 0;
  };
} // namespace

struct adaflo_comm
{
  adaflo_ctx *ctx = nullptr;
  int         rank = 0, world = 1, grid[3] = {1, 1, 1}, coords[3] = {0, 0, 0};
  unsigned    iface = 0; // brick faces shared with another rank (bit 2 d + side)
  std::vector<Neighbour> nbrs;
  Field       field[2];
  Round       copy, add;
  double     *sbuf = nullptr, *rbuf = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t  ev_packed = nullptr, ev_arrived = nullptr;
  // second compute stream of the overlapped schedule: packs, unpacks and the cells at the interface run here, beside
  // the interior cells on the engine's stream
  hipStream_t aux_stream = nullptr;
  hipEvent_t  ev_src = nullptr, ev_iface = nullptr, ev_done = nullptr;
  // transport
  void                *nccl = nullptr; // ncclComm_t
  adaflo_exchange_fn   exchange_cb = nullptr;
  adaflo_allreduce_fn  allreduce_cb = nullptr;
  void                *user = nullptr;
  // global pressure-mean projection (:117-168, :191-205): weights of the OWNED pressure rows,
  // 1 / (global sum) on the device
  double *w_owned = nullptr, *d_inv = nullptr, *d_dot = nullptr;
  bool    projection = false;
  bool    force_phased = false; // measurement aid: the three-phase schedule also with world == 1
  // get_matvec_statistics (:1194-1206): device time of every adaflo_ns_vmult_distributed of this rank, events on the
  // engine stream around the whole sequence (the stream waits for both exchanges inside it)
  adaflo_hip::EventTimer matvec_timer;
  // optional device times of the pieces of adaflo_ns_vmult_distributed (adaflo_comm_set_phase_timing): 0 ghost update of src
  // (pack, messages, unpack), 1 cells at the interface + their seam sums, 2 compress(add) of dst, 3 interior cells, 4 tail
  // (remaining seam sums, constrained rows, mean-value projection).  0-2 run on the auxiliary stream beside 3.
  adaflo_hip::EventTimer phase_timer[5];
  bool                   phase_timing = false;
  double                *d_stats = nullptr; // `world` doubles for the reduction of the per-rank times
  std::string last_error;
};

namespace
{
  int cfail(adaflo_comm *c, const int code, const std::string &msg)
  {
    if (c)
      {
        c->last_error = msg;
        if (c->ctx)
          c->ctx->last_error = msg;
      }
    return code;
  }

  void region_of(const int o[3], const int nn[3], int lo[3], int hi[3])
  {
    for (int d = 0; d < 3; ++d)
      {
        lo[d] = o[d] <= 0 ? 0 : nn[d] - 1;
        hi[d] = o[d] == 0 ? nn[d] : lo[d] + 1;
      }
  }

  long region_size(const int o[3], const Field &f)
  {
    int lo[3], hi[3];
    region_of(o, f.nn, lo, hi);
    return (long)(hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]) * f.ncomp;
  }

  // messages are [neighbour][field]: one contiguous message per neighbour
  void build_round(adaflo_comm *c, Round &R, const bool add_mode)
  {
    for (int n = 0; n < (int)c->nbrs.size(); ++n)
      {
        const Neighbour &nb = c->nbrs[n];
        const bool positive = nb.o[0] >= 0 && nb.o[1] >= 0 && nb.o[2] >= 0; // I am on the low side: owner
        const bool negative = nb.o[0] <= 0 && nb.o[1] <= 0 && nb.o[2] <= 0;
        if (add_mode || positive)
          R.send_nb.push_back(n);
        if (add_mode || negative)
          R.recv_nb.push_back(n);
      }
    auto layout = [&](const std::vector<int> &list, std::vector<long> &off, std::vector<long> &cnt,
                      std::vector<int> &peer, long &total) {
      total = 0;
      for (int n : list)
        {
          off.push_back(total);
          long m = 0;
          for (int f = 0; f < 2; ++f)
            m += region_size(c->nbrs[n].o, c->field[f]);
          cnt.push_back(m);
          peer.push_back(c->nbrs[n].rank);
          total += m;
        }
    };
    layout(R.send_nb, R.send_off, R.send_cnt, R.send_peer, R.n_send);
    layout(R.recv_nb, R.recv_off, R.recv_cnt, R.recv_peer, R.n_recv);
    // by_rank: regions in the order of their senders' ranks (add mode: one global summation order)
    auto plan_for = [&](const std::vector<int> &list, const std::vector<long> &off, const int f, const int only_cls,
                        const bool by_rank = false, const bool by_class = false) {
      HaloPlan P{};
      P.ncomp = c->field[f].ncomp;
      for (int d = 0; d < 3; ++d)
        P.nn[d] = c->field[f].nn[d];
      P.n_regions = 0;
      P.offset[0] = 0;
      P.self_pos  = 0;
      std::vector<size_t> order(list.size());
      for (size_t q = 0; q < list.size(); ++q)
        order[q] = q;
      if (by_class)
        std::stable_sort(order.begin(), order.end(),
                         [&](const size_t a, const size_t b) { return c->nbrs[list[a]].cls < c->nbrs[list[b]].cls; });
      if (by_rank)
        {
          std::stable_sort(order.begin(), order.end(),
                           [&](const size_t a, const size_t b) { return c->nbrs[list[a]].rank < c->nbrs[list[b]].rank; });
          for (size_t q = 0; q < list.size(); ++q)
            P.self_pos += c->nbrs[list[q]].rank < c->rank;
        }
      for (size_t qq = 0; qq < list.size(); ++qq)
        {
          const size_t     q  = order[qq];
          const Neighbour &nb = c->nbrs[list[q]];
          if (only_cls > 0 && nb.cls != only_cls)
            continue;
          const int r = P.n_regions++;
          region_of(nb.o, c->field[f].nn, P.lo[r], P.hi[r]);
          P.start[r]      = off[q] + (f == 1 ? region_size(nb.o, c->field[0]) : 0);
          P.offset[r + 1] = P.offset[r] + region_size(nb.o, c->field[f]);
        }
      return P;
    };
    for (int f = 0; f < 2; ++f)
      {
        R.pack[f]       = plan_for(R.send_nb, R.send_off, f, 0);
        R.unpack_add[f] = plan_for(R.recv_nb, R.recv_off, f, 0, true);
        for (int cls = 1; cls <= 3; ++cls)
          R.unpack_copy[f][cls - 1] = plan_for(R.recv_nb, R.recv_off, f, cls);
        R.unpack_copy_all[f] = plan_for(R.recv_nb, R.recv_off, f, 0, false, true);
      }
  }

  int transport_exchange(adaflo_comm *c, const Round &R)
  {
    if (R.send_nb.empty() && R.recv_nb.empty())
      return 0;
    if (c->exchange_cb)
      {
        std::vector<int64_t> so(R.send_off.begin(), R.send_off.end()), sc(R.send_cnt.begin(), R.send_cnt.end());
        std::vector<int64_t> ro(R.recv_off.begin(), R.recv_off.end()), rc(R.recv_cnt.begin(), R.recv_cnt.end());
        return c->exchange_cb(c->user, c->sbuf, so.data(), sc.data(), R.send_peer.data(), (int)so.size(), c->rbuf,
                              ro.data(), rc.data(), R.recv_peer.data(), (int)ro.size(), c->comm_stream) == 0 ?
                 0 :
                 cfail(c, ADAFLO_EHIP, "exchange callback failed");
      }
    RcclApi &a = rccl();
    int      e = a.GroupStart();
    for (size_t q = 0; q < R.send_nb.size() && e == 0; ++q)
      e = a.Send(c->sbuf + R.send_off[q], (size_t)R.send_cnt[q], NCCL_DOUBLE, R.send_peer[q], c->nccl, c->comm_stream);
    for (size_t q = 0; q < R.recv_nb.size() && e == 0; ++q)
      e = a.Recv(c->rbuf + R.recv_off[q], (size_t)R.recv_cnt[q], NCCL_DOUBLE, R.recv_peer[q], c->nccl, c->comm_stream);
    const int e2 = a.GroupEnd();
    if (e != 0 || e2 != 0)
      return cfail(c, ADAFLO_EHIP, std::string("RCCL send/recv failed: ") + a.GetErrorString(e != 0 ? e : e2));
    return 0;
  }

  // sum over the ranks of n doubles at buf, ordered after everything enqueued on `stream` so far;
  // `stream` continues after the result has arrived.  RCCL: all operations of the communicator
  // stay on the communication stream.
  int transport_allreduce(adaflo_comm *c, double *buf, const int n, hipStream_t stream)
  {
    if (c->world == 1)
      return 0;
    if (c->allreduce_cb)
      return c->allreduce_cb(c->user, buf, n, stream) == 0 ? 0 : cfail(c, ADAFLO_EHIP, "allreduce callback failed");
    if (hipEventRecord(c->ev_packed, stream) != hipSuccess || hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0) != hipSuccess)
      return cfail(c, ADAFLO_EHIP, "event failed");
    const int e = rccl().AllReduce(buf, buf, (size_t)n, NCCL_DOUBLE, NCCL_SUM, c->nccl, c->comm_stream);
    if (e != 0)
      return cfail(c, ADAFLO_EHIP, std::string("RCCL all-reduce failed: ") + rccl().GetErrorString(e));
    if (hipEventRecord(c->ev_arrived, c->comm_stream) != hipSuccess || hipStreamWaitEvent(stream, c->ev_arrived, 0) != hipSuccess)
      return cfail(c, ADAFLO_EHIP, "event failed");
    return 0;
  }

  // pack on the engine stream, then hand the buffer to the communication stream
  int exchange_start(adaflo_comm *c, const Round &R, double *u, double *p)
  {
    adaflo_ctx *ctx = c->ctx;
    if (int e = launch_halo_pair(ctx, u, p, c->sbuf, R.pack[0], R.pack[1], 0)) // both fields, one launch
      return cfail(c, e, "halo pack failed");
    if (hipEventRecord(c->ev_packed, ctx->stream) != hipSuccess ||
        hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0) != hipSuccess)
      return cfail(c, ADAFLO_EHIP, "event failed");
    if (int e = transport_exchange(c, R))
      return e;
    if (hipEventRecord(c->ev_arrived, c->comm_stream) != hipSuccess)
      return cfail(c, ADAFLO_EHIP, "event failed");
    return 0;
  }

  // the engine stream waits for the messages, then unpacks.  copy: faces first, corners last, so
  // that the lowest sharer (the owner) wins; add: every replica ends up with the total, summed in
  // a fixed order
  int exchange_finish(adaflo_comm *c, const Round &R, double *u, double *p, const bool add_mode)
  {
    adaflo_ctx *ctx = c->ctx;
    if (hipStreamWaitEvent(ctx->stream, c->ev_arrived, 0) != hipSuccess)
      return cfail(c, ADAFLO_EHIP, "event failed");
    // one launch for both fields; copy: the regions of all classes in one plan, the last one containing a node wins
    if (add_mode)
      {
        if (int e = launch_halo_pair(ctx, u, p, c->rbuf, R.unpack_add[0], R.unpack_add[1], 2))
          return cfail(c, e, "halo unpack failed");
      }
    else if (int e = launch_halo_pair(ctx, u, p, c->rbuf, R.unpack_copy_all[0], R.unpack_copy_all[1], 3))
      return cfail(c, e, "halo unpack failed");
    // the next pack / exchange must not overwrite the buffers before these kernels have read them
    if (hipEventRecord(c->ev_packed, ctx->stream) != hipSuccess ||
        hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0) != hipSuccess)
      return cfail(c, ADAFLO_EHIP, "event failed");
    return 0;
  }

  int nn_of(const adaflo_ctx *ctx, const int degree, const int d)
  {
    return degree * ctx->desc.ncell[d] + 1;
  }

  int comm_setup(adaflo_comm *c, adaflo_ctx *ctx, const int rank, const int world, const int *grid)
  {
    if (!ctx || !grid || world < 1 || rank < 0 || rank >= world || grid[0] * grid[1] * grid[2] != world)
      return cfail(c, ADAFLO_EINVAL, "invalid communicator arguments");
    c->ctx   = ctx;
    c->rank  = rank;
    c->world = world;
    for (int d = 0; d < 3; ++d)
      c->grid[d] = grid[d];
    c->coords[0] = rank % grid[0];
    c->coords[1] = (rank / grid[0]) % grid[1];
    c->coords[2] = rank / (grid[0] * grid[1]);
    for (int d = 0; d < 3; ++d)
      {
        if (c->coords[d] > 0)
          c->iface |= 1u << (2 * d);
        if (c->coords[d] < grid[d] - 1)
          c->iface |= 1u << (2 * d + 1);
      }
    for (int oz = -1; oz <= 1; ++oz)
      for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox)
          {
            const int o[3] = {ox, oy, oz};
            if (!ox && !oy && !oz)
              continue;
            int  cc[3];
            bool ok = true;
            for (int d = 0; d < 3; ++d)
              {
                cc[d] = c->coords[d] + o[d];
                ok    = ok && cc[d] >= 0 && cc[d] < grid[d];
              }
            if (!ok)
              continue;
            Neighbour nb{{ox, oy, oz}, cc[0] + grid[0] * (cc[1] + grid[1] * cc[2]), (ox != 0) + (oy != 0) + (oz != 0)};
            c->nbrs.push_back(nb);
          }
    std::stable_sort(c->nbrs.begin(), c->nbrs.end(), [](const Neighbour &a, const Neighbour &b) { return a.cls < b.cls; });
    const int k = ctx->k;
    c->field[0] = Field{k, 3, {nn_of(ctx, k, 0), nn_of(ctx, k, 1), nn_of(ctx, k, 2)}};
    c->field[1] = Field{k - 1, 1, {nn_of(ctx, k - 1, 0), nn_of(ctx, k - 1, 1), nn_of(ctx, k - 1, 2)}};
    build_round(c, c->copy, false);
    build_round(c, c->add, true);
    const long ns = std::max(c->copy.n_send, c->add.n_send), nr = std::max(c->copy.n_recv, c->add.n_recv);
    if (hipSetDevice(ctx->desc.device) != hipSuccess ||
        hipMalloc(&c->sbuf, std::max(ns, 1L) * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->rbuf, std::max(nr, 1L) * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_inv, 2 * sizeof(double)) != hipSuccess ||
        hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        // Events that only order streams of ONE device (ev_src, ev_iface, ev_done below) carry no system-scope fence: a
        // default event makes every record a cache write-back + invalidate of the whole device -- behind a sweep kernel
        // that is 25-30 us during which the next kernel cannot start (profiles/r05_trace_through_comm_before.log); the
        // kernel boundaries already carry the device-scope release / acquire such hand-overs need.  The two events next to
        // the message transport keep the default semantics: what the pack kernel wrote must be visible to RCCL's send
        // (possibly a peer's read over xGMI), and what a peer wrote into rbuf must be seen by the unpack kernel -- the
        // configuration every stream-ordered RCCL program runs in; they are recorded on the auxiliary and the communication
        // stream, off the engine stream's critical path
        hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_arrived, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_iface, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess)
      return cfail(c, ADAFLO_ENOMEM, "communicator allocation failed");
    c->d_dot = c->d_inv + 1;
    return 0;
  }

  // global version of :117-168 (mode 0): pressure mass weights, summed over the sharers,
  // restricted to the rows this rank owns; 1 / (global sum) stays on the device
  int setup_projection(adaflo_comm *c)
  {
    adaflo_ctx *ctx = c->ctx;
    const long  np  = ctx->n_nodes_p;
    if (hipMalloc(&c->w_owned, np * sizeof(double)) != hipSuccess)
      return cfail(c, ADAFLO_ENOMEM, "allocation failed");
    if (int e = launch_fill(ctx, c->w_owned, 0., np))
      return cfail(c, e, "fill failed");
    if (int e = adaflo_ns_pressure_mass_weight_add(ctx, c->w_owned))
      return e;
    if (c->world > 1)
      {
        // the velocity part of the message is exchanged as well (one code path); its content is unused
        double *dummy_u = nullptr;
        if (hipMalloc(&dummy_u, 3 * ctx->n_nodes_u * sizeof(double)) != hipSuccess)
          return cfail(c, ADAFLO_ENOMEM, "allocation failed");
        (void)hipMemsetAsync(dummy_u, 0, 3 * ctx->n_nodes_u * sizeof(double), ctx->stream);
        int e = exchange_start(c, c->add, dummy_u, c->w_owned);
        if (!e)
          e = exchange_finish(c, c->add, dummy_u, c->w_owned, true);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(dummy_u);
        if (e)
          return e;
        // a lower neighbour owns my low faces
        unsigned low = 0;
        for (int d = 0; d < 3; ++d)
          if (c->coords[d] > 0)
            low |= 1u << (2 * d);
        if (int e2 = launch_prepare_dst(ctx, c->w_owned, c->w_owned, np, 1, c->field[1].nn[0], c->field[1].nn[1],
                                        c->field[1].nn[2], low, 0., false))
          return cfail(c, e2, "mask failed");
      }
    if (int e = launch_sum_to(ctx, c->w_owned, np, c->d_dot))
      return cfail(c, e, "sum failed");
    if (int e = transport_allreduce(c, c->d_dot, 1, ctx->stream))
      return e;
    if (int e = launch_reciprocal(ctx, c->d_inv, c->d_dot))
      return cfail(c, e, "reciprocal failed");
    c->projection = true;
    return hipStreamSynchronize(ctx->stream) == hipSuccess ? 0 : cfail(c, ADAFLO_EHIP, "synchronize failed");
  }
} // namespace

extern "C" {

const char *adaflo_comm_last_error(const adaflo_comm *c)
{
  return c ? c->last_error.c_str() : "null communicator";
}

int adaflo_comm_get_unique_id(adaflo_comm_unique_id *id)
{
  if (!id)
    return ADAFLO_EINVAL;
  const std::string err = load_rccl();
  if (!err.empty())
    return ADAFLO_EUNSUPPORTED;
  return rccl().GetUniqueId(id) == 0 ? 0 : ADAFLO_EHIP;
}

int adaflo_comm_create(adaflo_ctx *ctx, const adaflo_comm_unique_id *id, int rank, int world, const int *grid,
                       int pressure_average_fix, adaflo_comm **out)
{
  if (ctx && ctx->indexed)
    return ADAFLO_EUNSUPPORTED; // (indexed context: needs the structured brick)
  if (!out || !id)
    return ADAFLO_EINVAL;
  *out = nullptr;
  adaflo_comm *c = new adaflo_comm;
  int          e = comm_setup(c, ctx, rank, world, grid);
  if (!e)
    {
      const std::string err = load_rccl();
      if (!err.empty())
        e = cfail(c, ADAFLO_EUNSUPPORTED, err);
      else if (int r = rccl().CommInitRank(&c->nccl, world, *id, rank))
        e = cfail(c, ADAFLO_EHIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    }
  if (!e && pressure_average_fix)
    e = setup_projection(c);
  if (e)
    {
      if (ctx)
        ctx->last_error = c->last_error;
      adaflo_comm_destroy(c);
      return e;
    }
  *out = c;
  return 0;
}

int adaflo_comm_create_custom(adaflo_ctx *ctx, int rank, int world, const int *grid, adaflo_exchange_fn exchange,
                              adaflo_allreduce_fn allreduce, void *user, int pressure_average_fix, adaflo_comm **out)
{
  if (ctx && ctx->indexed)
    return ADAFLO_EUNSUPPORTED; // (indexed context: needs the structured brick)
  if (!out || (world > 1 && (!exchange || !allreduce)))
    return ADAFLO_EINVAL;
  *out = nullptr;
  adaflo_comm *c  = new adaflo_comm;
  c->exchange_cb  = exchange;
  c->allreduce_cb = allreduce;
  c->user         = user;
  int e           = comm_setup(c, ctx, rank, world, grid);
  if (!e && pressure_average_fix)
    e = setup_projection(c);
  if (e)
    {
      if (ctx)
        ctx->last_error = c->last_error;
      adaflo_comm_destroy(c);
      return e;
    }
  *out = c;
  return 0;
}

int adaflo_comm_destroy(adaflo_comm *c)
{
  if (!c)
    return ADAFLO_ENOTINIT;
  if (c->ctx)
    (void)hipStreamSynchronize(c->ctx->stream);
  if (c->comm_stream)
    (void)hipStreamSynchronize(c->comm_stream);
  if (c->nccl)
    (void)rccl().CommDestroy(c->nccl);
  for (double *p : {c->sbuf, c->rbuf, c->d_inv, c->w_owned, c->d_stats})
    if (p)
      (void)hipFree(p);
  c->matvec_timer.destroy();
  for (auto &t : c->phase_timer)
    t.destroy();
  if (c->aux_stream)
    (void)hipStreamSynchronize(c->aux_stream);
  for (hipEvent_t e : {c->ev_packed, c->ev_arrived, c->ev_src, c->ev_iface, c->ev_done})
    if (e)
      (void)hipEventDestroy(e);
  if (c->comm_stream)
    (void)hipStreamDestroy(c->comm_stream);
  if (c->aux_stream)
    (void)hipStreamDestroy(c->aux_stream);
  delete c;
  return 0;
}

unsigned adaflo_comm_interface_faces(const adaflo_comm *c)
{
  return c ? c->iface : 0u;
}

int adaflo_comm_update_ghost_values(adaflo_comm *c, double *vec_u, double *vec_p)
{
  if (!c || !vec_u || !vec_p)
    return ADAFLO_EINVAL;
  if (c->world == 1)
    return 0;
  if (int e = exchange_start(c, c->copy, vec_u, vec_p))
    return e;
  return exchange_finish(c, c->copy, vec_u, vec_p, false);
}

int adaflo_comm_compress_add(adaflo_comm *c, double *vec_u, double *vec_p)
{
  if (!c || !vec_u || !vec_p)
    return ADAFLO_EINVAL;
  if (c->world == 1)
    return 0;
  if (int e = exchange_start(c, c->add, vec_u, vec_p))
    return e;
  return exchange_finish(c, c->add, vec_u, vec_p, true);
}

// NavierStokesMatrix::get_matvec_statistics (navier_stokes_matrix.cc:1194-1206): Utilities::MPI::min_max_avg of the
// accumulated vmult time over the communicator, the number of applications, counters reset.  Collective.
int adaflo_comm_matvec_statistics(adaflo_comm *c, unsigned *count, adaflo_min_max_avg *stats)
{
  if (!c || !c->ctx)
    return ADAFLO_EINVAL;
  adaflo_ctx *ctx = c->ctx;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess)
    return cfail(c, ADAFLO_EHIP, "synchronize failed");
  c->matvec_timer.fold();
  const double mine = c->matvec_timer.seconds;
  if (count)
    *count = c->matvec_timer.count;
  c->matvec_timer.count   = 0;
  c->matvec_timer.seconds = 0.;
  std::vector<double> all((size_t)c->world, 0.);
  all[c->rank] = mine;
  if (c->world > 1)
    {
      // every rank contributes its time at its own position, zeros elsewhere: one sum gives all ranks all times
      if (!c->d_stats && hipMalloc(&c->d_stats, c->world * sizeof(double)) != hipSuccess)
        return cfail(c, ADAFLO_ENOMEM, "allocation failed");
      if (hipMemcpyAsync(c->d_stats, all.data(), c->world * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return cfail(c, ADAFLO_EHIP, "copy failed");
      if (int e = transport_allreduce(c, c->d_stats, c->world, ctx->stream))
        return e;
      if (hipMemcpyAsync(all.data(), c->d_stats, c->world * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        return cfail(c, ADAFLO_EHIP, "copy failed");
    }
  if (stats)
    {
      stats->sum = 0., stats->min = all[0], stats->max = all[0], stats->min_index = 0, stats->max_index = 0;
      for (int r = 0; r < c->world; ++r)
        {
          stats->sum += all[r];
          if (all[r] < stats->min)
            stats->min = all[r], stats->min_index = r;
          if (all[r] > stats->max)
            stats->max = all[r], stats->max_index = r;
        }
      stats->avg = stats->sum / c->world;
    }
  return 0;
}

int adaflo_comm_force_phased_schedule(adaflo_comm *c, int enabled)
{
  if (!c)
    return ADAFLO_EINVAL;
  c->force_phased = enabled != 0;
  return 0;
}

int adaflo_ns_vmult_distributed(adaflo_ctx *ctx, adaflo_comm *c, double *dst_u, double *dst_p, double *src_u,
                                double *src_p, int src_ghosts_valid)
{
  if (!ctx || !c || c->ctx != ctx || !dst_u || !dst_p || !src_u || !src_p)
    return ADAFLO_EINVAL;
  // (every sweep kernel has the three-phase form: Q2/Q1 with constant or variable coefficients, Q3..Q5)
  const bool phased = (c->world > 1 || c->force_phased) && adaflo_ns_supports_phases(ctx) != 0;
  struct Timed
  {
    adaflo_comm *c;
    hipEvent_t   stop;
    explicit Timed(adaflo_comm *cc)
      : c(cc)
      , stop(cc->ctx->timing ? cc->matvec_timer.start(cc->ctx->stream) : nullptr)
    {}
    ~Timed()
    {
      if (stop)
        (void)hipEventRecord(stop, c->ctx->stream);
      c->matvec_timer.count++;
    }
  } timed(c);
  hipEvent_t tail_stop = nullptr;
  if (c->world == 1 && !phased)
    {
      // (the context was created without the local mean-value fix when a communicator takes care of it)
      if (int e = adaflo_ns_vmult(ctx, dst_u, dst_p, src_u, src_p))
        return e;
    }
  else if (phased)
    {
      // Two compute streams (round 5).  The engine's stream S runs ALL interior cells as one launch; beside it, on the
      // auxiliary stream S2: ghost update of src (pack, messages on the communication stream, unpack), the cells at
      // the interface + the seam sums of the interface nodes, compress(add) of dst (pack, messages, unpack-add).  S joins
      // for the remaining seam sums and at the end.  The former schedule split the interior into two launches on ONE stream
      // (A | interface | B): three partially filled rounds of workgroups where the plain operator has two, and every
      // hand-over to the communication stream sat between two sweep kernels (32^3 Q4/Q3: +26 % with no message at all).
      //   S : set-up . [src ready] . interior cells ........................ [wait iface] seam sums . [wait done] rows
      //   S2:           [wait] pack > msgs > unpack . interface cells + seams . [iface] pack > msgs > unpack-add . [done]
      struct OnStream // the launchers enqueue on ctx->stream
      {
        adaflo_ctx *ctx;
        hipStream_t saved;
        OnStream(adaflo_ctx *x, hipStream_t s)
          : ctx(x)
          , saved(x->stream)
        {
          x->stream = s;
        }
        ~OnStream() { ctx->stream = saved; }
      };
      hipStream_t S = ctx->stream, S2 = c->aux_stream;
      // an error return from inside the schedule must not leave work on the auxiliary stream that the engine stream was
      // never joined with (a caller that continues on S would race with it): wait for it on the way out
      struct JoinOnError
      {
        hipStream_t s;
        bool        armed = true;
        ~JoinOnError()
        {
          if (armed)
            (void)hipStreamSynchronize(s);
        }
      } join{S2};
      // tables and streaming copies of the state are brought up to date on S before S2 may touch them
      if (int e = adaflo_ns_vmult_phase(ctx, dst_u, dst_p, src_u, src_p, 5, c->iface))
        return e;
      if (hipEventRecord(c->ev_src, S) != hipSuccess || hipStreamWaitEvent(S2, c->ev_src, 0) != hipSuccess)
        return cfail(c, ADAFLO_EHIP, "event failed");
      struct Piece // (times one piece on one stream when adaflo_comm_set_phase_timing is on)
      {
        hipEvent_t  stop;
        hipStream_t s;
        Piece(adaflo_comm *cc, const int i, hipStream_t st)
          : stop(cc->phase_timing ? cc->phase_timer[i].start(st) : nullptr)
          , s(st)
        {
          if (cc->phase_timing)
            cc->phase_timer[i].count++;
        }
        void done()
        {
          if (stop)
            (void)hipEventRecord(stop, s);
          stop = nullptr;
        }
      };
      {
        Piece t3(c, 3, S);
        if (int e = adaflo_ns_vmult_phase(ctx, dst_u, dst_p, src_u, src_p, 3, c->iface))
          return e;
        t3.done();
      }
      {
        OnStream on(ctx, S2);
        if (!src_ghosts_valid)
          {
            Piece t0(c, 0, S2);
            if (int e = exchange_start(c, c->copy, src_u, src_p))
              return e;
            if (int e = exchange_finish(c, c->copy, src_u, src_p, false))
              return e;
            t0.done();
          }
        Piece t1(c, 1, S2);
        if (int e = adaflo_ns_vmult_phase(ctx, dst_u, dst_p, src_u, src_p, 1, c->iface))
          return e;
        t1.done();
        if (hipEventRecord(c->ev_iface, S2) != hipSuccess)
          return cfail(c, ADAFLO_EHIP, "event failed");
        Piece t2(c, 2, S2);
        if (int e = exchange_start(c, c->add, dst_u, dst_p))
          return e;
        if (int e = exchange_finish(c, c->add, dst_u, dst_p, true))
          return e;
        // rows on boundary x interface: the sharers added their +-src as well.  Here, behind the unpack-add on the auxiliary
        // stream (which has time to spare beside the interior cells), not as one more kernel behind the tail on the engine
        // stream (round 6: 32^3 Q4/Q3 through the communicator 0.209 -> see profiles/r06_through_comm_q4_32.log); every other
        // writer of these rows writes the same +-src, and the tail's seam sums skip constrained rows
        if (int e = adaflo_ns_apply_constrained_rows(ctx, dst_u, dst_p, src_u, src_p))
          return e;
        t2.done();
        if (hipEventRecord(c->ev_done, S2) != hipSuccess)
          return cfail(c, ADAFLO_EHIP, "event failed");
      }
      if (hipStreamWaitEvent(S, c->ev_iface, 0) != hipSuccess)
        return cfail(c, ADAFLO_EHIP, "event failed");
      tail_stop = c->phase_timing ? c->phase_timer[4].start(S) : nullptr;
      if (c->phase_timing)
        c->phase_timer[4].count++;
      if (int e = adaflo_ns_vmult_phase(ctx, dst_u, dst_p, src_u, src_p, 4, c->iface))
        return e;
      if (hipStreamWaitEvent(S, c->ev_done, 0) != hipSuccess)
        return cfail(c, ADAFLO_EHIP, "event failed");
      join.armed = false; // (S now waits for everything S2 was given)
    }
  else
    {
      if (!src_ghosts_valid)
        if (int e = adaflo_comm_update_ghost_values(c, src_u, src_p))
          return e;
      if (int e = adaflo_ns_vmult(ctx, dst_u, dst_p, src_u, src_p))
        return e;
      if (int e = adaflo_comm_compress_add(c, dst_u, dst_p))
        return e;
    }
  if (c->world > 1 && !phased) // rows on boundary x interface: the sharers added their +-src as well (phased: done above)
    if (int e = adaflo_ns_apply_constrained_rows(ctx, dst_u, dst_p, src_u, src_p))
      return e;
  // :191-205 with the global weights; skipped for the projection scheme and the stationary equation
  if (c->projection && ctx->ns.linearization != ADAFLO_PROJECTION &&
      ctx->ns.physical_type != ADAFLO_INCOMPRESSIBLE_STATIONARY)
    {
      if (int e = launch_dot_to(ctx, c->w_owned, dst_p, ctx->n_nodes_p, c->d_dot))
        return cfail(c, e, "dot failed");
      if (int e = transport_allreduce(c, c->d_dot, 1, ctx->stream))
        return e;
      if (int e = launch_subtract_scaled(ctx, dst_p, c->d_dot, c->d_inv, ctx->n_nodes_p))
        return cfail(c, e, "projection failed");
    }
  if (tail_stop)
    (void)hipEventRecord(tail_stop, ctx->stream);
  return 0;
}

int adaflo_comm_set_phase_timing(adaflo_comm *c, int enabled)
{
  if (!c)
    return ADAFLO_EINVAL;
  c->phase_timing = enabled != 0;
  return 0;
}

int adaflo_comm_phase_statistics(adaflo_comm *c, unsigned *count, double seconds[5])
{
  if (!c || !seconds)
    return ADAFLO_EINVAL;
  if (hipStreamSynchronize(c->ctx->stream) != hipSuccess || hipStreamSynchronize(c->aux_stream) != hipSuccess)
    return cfail(c, ADAFLO_EHIP, "synchronisation failed");
  unsigned n = 0;
  for (int i = 0; i < 5; ++i)
    {
      c->phase_timer[i].fold();
      seconds[i] = c->phase_timer[i].seconds;
      n = c->phase_timer[i].count > n ? c->phase_timer[i].count : n;
      c->phase_timer[i].seconds = 0.;
      c->phase_timer[i].count   = 0;
    }
  if (count)
    *count = n;
  return 0;
}

} // extern "C"

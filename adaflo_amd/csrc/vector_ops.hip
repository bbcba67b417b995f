// vector_ops.hip -- whole-vector helper kernels around the cell loops:
// zero + constrained-row fix-up, pressure-mean projection, state re-layout.
#include "kernels.hpp"

#include <algorithm>

namespace adaflo_hip
{
  namespace
  {
    constexpr int VT = 256;

    inline unsigned grid_for(const long n, const int per_thread = 1)
    {
      long b = (n + (long)VT * per_thread - 1) / ((long)VT * per_thread);
      if (b < 1)
        b = 1;
      if (b > 256 * 16)
        b = 256 * 16; // grid-stride beyond that
      return (unsigned)b;
    }

    // dst = constrained ? sign*src : (zero_rest ? 0 : dst)
    // source/navier_stokes_matrix.cc:229 (dst = 0) + :247-256 (constrained rows)
    __global__ __launch_bounds__(VT) void prepare_dst_kernel(double *__restrict__ dst,
                                                             const double *__restrict__ src,
                                                             const long n_nodes, const int ncomp,
                                                             const int nnx, const int nny,
                                                             const int nnz, const uint32_t mask,
                                                             const double sign, const bool zero_rest)
    {
      const long n = n_nodes * ncomp;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          const long node = i / ncomp;
          const int  c    = (int)(i - node * ncomp);
          const int  I = node % nnx, J = (node / nnx) % nny, K = node / ((long)nnx * nny);
          if (mask != 0u && on_constrained_face(I, J, K, nnx, nny, nnz, mask, ncomp == 1 ? 1 : 3, c))
            dst[i] = sign * src[i];
          else if (zero_rest)
            dst[i] = 0.;
        }
    }

    // the same with the constraint flags of an indexed context
    __global__ __launch_bounds__(VT) void prepare_dst_flags_kernel(double *__restrict__ dst, const double *__restrict__ src,
                                                                   const long n, const unsigned char *__restrict__ flags,
                                                                   const double sign, const bool zero_rest)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          if (flags[i])
            dst[i] = sign * src[i];
          else if (zero_rest)
            dst[i] = 0.;
        }
    }

    // dst = sign * src on the constrained faces only (:247-256): one work item per (face, node of the face,
    // component) -- 6 n^2 instead of n^3 entries; a node on several constrained faces is written more than once,
    // with the same value
    __global__ __launch_bounds__(VT) void constrained_faces_kernel(double *__restrict__ dst, const double *__restrict__ src,
                                                                   const int ncomp, const int nnx, const int nny,
                                                                   const int nnz, const uint32_t mask, const double sign)
    {
      const int  stride = ncomp == 1 ? 1 : 3;
      const long fs[3] = {(long)nny * nnz, (long)nnx * nnz, (long)nnx * nny}; // nodes of an x / y / z face
      const long total = 2 * (fs[0] + fs[1] + fs[2]) * ncomp;
      for (long it = blockIdx.x * (long)VT + threadIdx.x; it < total; it += (long)gridDim.x * VT)
        {
          const int c = (int)(it % ncomp);
          long      r = it / ncomp;
          int       f = 0;
          for (; f < 6; ++f)
            {
              if (r < fs[f / 2])
                break;
              r -= fs[f / 2];
            }
          if (!(mask >> (stride * f + c) & 1u))
            continue;
          const int d = f / 2, side = f & 1;
          int       I, J, K;
          if (d == 0)
            {
              I = side ? nnx - 1 : 0;
              J = (int)(r % nny);
              K = (int)(r / nny);
            }
          else if (d == 1)
            {
              J = side ? nny - 1 : 0;
              I = (int)(r % nnx);
              K = (int)(r / nnx);
            }
          else
            {
              K = side ? nnz - 1 : 0;
              I = (int)(r % nnx);
              J = (int)(r / nnx);
            }
          const long i = (((long)K * nny + J) * nnx + I) * ncomp + c;
          dst[i]       = sign * src[i];
        }
    }

    __global__ __launch_bounds__(VT) void dot_partial_kernel(const double *__restrict__ a,
                                                             const double *__restrict__ b,
                                                             const long n, double *__restrict__ part)
    {
      __shared__ double red[VT / 64];
      double            s = 0.;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        s += b ? a[i] * b[i] : a[i];
      for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
      if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0)
        {
          double t = 0.;
          for (int w = 0; w < VT / 64; ++w)
            t += red[w];
          part[blockIdx.x] = t;
        }
    }

    // One step of the modified Gram-Schmidt of FGMRES in one pass: w -= (*s) v (the update axpy_dev_kernel of krylov.hip
    // does, same expression) and, with the updated values, the partial sums of w . next -- the dot product the next
    // step needs (next == nullptr: w . w).  Thread / block mapping and summation order of dot_partial_kernel, so that
    // the result is bitwise the one of the two separate kernels.
    __global__ __launch_bounds__(VT) void gs_step_kernel(double *__restrict__ w, const double *__restrict__ s,
                                                         const double *__restrict__ v, const double *__restrict__ next,
                                                         const long n, double *__restrict__ part)
    {
      __shared__ double red[VT / 64];
      const double      a = -*s;
      double            sum = 0.;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          const double wi = a * v[i] + 1. * w[i];
          w[i]            = wi;
          sum += wi * (next ? next[i] : wi);
        }
      for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off, 64);
      if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = sum;
      __syncthreads();
      if (threadIdx.x == 0)
        {
          double t = 0.;
          for (int k = 0; k < VT / 64; ++k)
            t += red[k];
          part[blockIdx.x] = t;
        }
    }

    // single block: result[0] = sum(part[0..np))  (fixed order -> deterministic)
    __global__ __launch_bounds__(VT) void dot_final_kernel(const double *__restrict__ part,
                                                           const int np, double *__restrict__ result,
                                                           double *__restrict__ host_result)
    {
      __shared__ double red[VT / 64];
      double            s = 0.;
      for (int i = threadIdx.x; i < np; i += VT)
        s += part[i];
      for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
      if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0)
        {
          double t = 0.;
          for (int w = 0; w < VT / 64; ++w)
            t += red[w];
          result[0] = t;
          if (host_result) // pinned, device-mapped: visible to the host after the stream synchronises
            host_result[0] = t;
        }
    }

    // v -= (*prod) * inv * modes
    // v -= (sum of the np partial dot products) * inv * modes.  Every block adds up the partials itself (a
    // few thousand doubles from L2, the same fixed order in every block: all blocks use the same factor),
    // which saves the separate one-block reduction launch between the dot product and the update.
    __global__ __launch_bounds__(VT) void project_kernel(double *__restrict__ v,
                                                         const double *__restrict__ modes,
                                                         const double *__restrict__ part, const int np,
                                                         const double inv, const long n)
    {
      __shared__ double red[VT / 64];
      double            s = 0.;
      for (int i = threadIdx.x; i < np; i += VT)
        s += part[i];
      for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
      if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
      __syncthreads();
      double t = 0.;
      for (int w = 0; w < VT / 64; ++w)
        t += red[w];
      const double f = t * inv;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        v[i] -= f * modes[i];
    }

    __global__ __launch_bounds__(VT) void sadd_kernel(double *__restrict__ x, const double a,
                                                      const double *__restrict__ y, const long n)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        x[i] = a * x[i] + (y ? y[i] : 0.);
    }

    __global__ __launch_bounds__(VT) void fill_kernel(double *__restrict__ x, const double v,
                                                      const long n)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        x[i] = v;
    }

    // to_generic: dst[cell][comp][q] = src[cell][q][comp]; else the inverse
    __global__ __launch_bounds__(VT) void transpose_state_kernel(double *__restrict__ dst,
                                                                 const double *__restrict__ src,
                                                                 const long n_cells, const int nq,
                                                                 const int ncomp,
                                                                 const bool to_generic)
    {
      const long per = (long)nq * ncomp, n = n_cells * per;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          const long cell = i / per;
          const int  r    = (int)(i - cell * per);
          if (to_generic)
            {
              const int comp = r / nq, q = r % nq; // i indexes dst
              dst[i]         = src[cell * per + (long)q * ncomp + comp];
            }
          else
            {
              const int q = r / ncomp, comp = r % ncomp; // i indexes dst
              dst[i]      = src[cell * per + (long)comp * nq + q];
            }
        }
    }

    // interface-region copy between a brick vector and a packed message buffer
    // mode 0: buf <- vec (pack), 1: vec <- buf (unpack, copy), 2: vec += buf (unpack, add).
    // Regions of different neighbours overlap on edges and corners.  In add mode the FIRST region
    // that contains a node sums the entries of all regions containing it and the vector's own
    // value, in region order: no atomics, and the replicas of an interface DoF end up bitwise
    // identical on every rank and run.
    // one entry t of a plan.  mode 3: copy where the LAST region that contains a node wins -- with the regions ordered
    // faces, edges, corners this equals three passes (faces first, corners last: the lowest sharer, the owner, wins)
    __device__ __forceinline__ void halo_item(double *__restrict__ vec, double *__restrict__ buf, const HaloPlan &plan,
                                              const int mode, const long t)
    {
      int r = 0;
      while (t >= plan.offset[r + 1])
        ++r;
      long       e  = t - plan.offset[r];
      const long el = e;
      const int  nc = plan.ncomp, ni = plan.hi[r][0] - plan.lo[r][0], nj = plan.hi[r][1] - plan.lo[r][1];
      const int  c  = (int)(e % nc);
      e /= nc;
      const int i = (int)(e % ni) + plan.lo[r][0];
      e /= ni;
      const int  j   = (int)(e % nj) + plan.lo[r][1];
      const int  k   = (int)(e / nj) + plan.lo[r][2];
      const long idx = ((long)(k * (long)plan.nn[1] + j) * plan.nn[0] + i) * nc + c;
      auto       inside = [&](const int q) {
        return i >= plan.lo[q][0] && i < plan.hi[q][0] && j >= plan.lo[q][1] && j < plan.hi[q][1] && k >= plan.lo[q][2] &&
               k < plan.hi[q][2];
      };
      if (mode == 0)
        buf[plan.start[r] + el] = vec[idx];
      else if (mode == 1)
        vec[idx] = buf[plan.start[r] + el];
      else if (mode == 3)
        {
          bool last = true;
          for (int q = r + 1; q < plan.n_regions; ++q)
            last = last && !inside(q);
          if (last)
            vec[idx] = buf[plan.start[r] + el];
        }
      else
        {
          bool first = true;
          for (int q = 0; q < r; ++q)
            first = first && !inside(q);
          if (!first)
            return;
          // own value and received partial sums in ONE global order (the regions arrive sorted by
          // the rank of their sender, self_pos = where this rank sits in that order): all sharers
          // add the same numbers in the same sequence
          double sum     = 0.;
          bool   started = false;
          auto   add     = [&](const double x) {
            sum     = started ? sum + x : x;
            started = true;
          };
          for (int q = r; q < plan.n_regions; ++q)
            {
              if (q == plan.self_pos || (q == r && plan.self_pos < r))
                add(vec[idx]);
              if (inside(q))
                {
                  const int  qi = plan.hi[q][0] - plan.lo[q][0], qj = plan.hi[q][1] - plan.lo[q][1];
                  const long le = ((long)((k - plan.lo[q][2]) * qj + (j - plan.lo[q][1])) * qi + (i - plan.lo[q][0])) * nc + c;
                  add(buf[plan.start[q] + le]);
                }
            }
          if (plan.self_pos >= plan.n_regions)
            add(vec[idx]);
          vec[idx] = sum;
        }
    }

    __global__ __launch_bounds__(VT) void halo_kernel(double *__restrict__ vec, double *__restrict__ buf,
                                                      const HaloPlan plan, const int mode)
    {
      const long total = plan.offset[plan.n_regions];
      for (long t = blockIdx.x * (long)VT + threadIdx.x; t < total; t += (long)gridDim.x * VT)
        halo_item(vec, buf, plan, mode, t);
    }
    // both fields of a block vector in ONE launch (the exchanges of a distributed vmult are launch-bound on small bricks)
    __global__ __launch_bounds__(VT) void halo_pair_kernel(double *__restrict__ vec0, double *__restrict__ vec1,
                                                           double *__restrict__ buf, const HaloPlan plan0,
                                                           const HaloPlan plan1, const int mode)
    {
      const long n0 = plan0.offset[plan0.n_regions], total = n0 + plan1.offset[plan1.n_regions];
      for (long t = blockIdx.x * (long)VT + threadIdx.x; t < total; t += (long)gridDim.x * VT)
        {
          if (t < n0)
            halo_item(vec0, buf, plan0, mode, t);
          else
            halo_item(vec1, buf, plan1, mode, t - n0);
        }
    }

    __global__ void reciprocal_kernel(double *out, const double *in)
    {
      *out = 1. / *in;
    }
    __global__ __launch_bounds__(VT) void subtract_scaled_kernel(double *__restrict__ v, const double *__restrict__ s,
                                                                 const double *__restrict__ t, const long n)
    {
      const double f = *s * *t;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        v[i] -= f;
    }

    int check()
    {
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    }
  } // namespace

  int launch_prepare_dst(adaflo_ctx *ctx, double *dst, const double *src, const long n_nodes,
                         const int ncomp, const int nnx, const int nny, const int nnz,
                         const uint32_t mask, const double sign, const bool zero_rest)
  {
    if (ctx->indexed) // (the flags of the space: three components = velocity, one = pressure)
      {
        hipLaunchKernelGGL(prepare_dst_flags_kernel, dim3(grid_for(n_nodes * ncomp)), dim3(VT), 0, ctx->stream, dst, src,
                           n_nodes * ncomp, ncomp == 3 ? ctx->d_flag_u : ctx->d_flag_p, sign, zero_rest);
        return check();
      }
    hipLaunchKernelGGL(prepare_dst_kernel, dim3(grid_for(n_nodes * ncomp)), dim3(VT), 0, ctx->stream,
                       dst, src, n_nodes, ncomp, nnx, nny, nnz, mask, sign, zero_rest);
    return check();
  }

  int launch_constrained_faces(adaflo_ctx *ctx, double *dst, const double *src, const int ncomp, const int nnx, const int nny,
                               const int nnz, const uint32_t mask, const double sign)
  {
    if (ctx->indexed)
      {
        const long n = (ncomp == 3 ? 3 * ctx->n_nodes_u : ctx->n_nodes_p);
        hipLaunchKernelGGL(prepare_dst_flags_kernel, dim3(grid_for(n)), dim3(VT), 0, ctx->stream, dst, src, n,
                           ncomp == 3 ? ctx->d_flag_u : ctx->d_flag_p, sign, false);
        return check();
      }
    if (mask == 0u)
      return 0;
    const long total = 2 * ((long)nny * nnz + (long)nnx * nnz + (long)nnx * nny) * ncomp;
    hipLaunchKernelGGL(constrained_faces_kernel, dim3(grid_for(total)), dim3(VT), 0, ctx->stream, dst, src, ncomp, nnx, nny,
                       nnz, mask, sign);
    return check();
  }

  static int ensure_scratch(adaflo_ctx *ctx, const size_t count)
  {
    if (!ctx->h_result)
      {
        // results of reductions travel through pinned, device-mapped host memory: a pageable
        // hipMemcpy D2H costs ~200 us per Krylov scalar on this platform
        if (hipHostMalloc((void **)&ctx->h_result, 8 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&ctx->h_result_dev, ctx->h_result, 0) != hipSuccess)
          return ADAFLO_ENOMEM;
      }
    if (ctx->scratch_count >= count)
      return 0;
    if (ctx->d_scratch)
      (void)hipFree(ctx->d_scratch);
    if (hipMalloc(&ctx->d_scratch, count * sizeof(double)) != hipSuccess)
      return ADAFLO_ENOMEM;
    ctx->scratch_count = count;
    return 0;
  }

  static int launch_dot(adaflo_ctx *ctx, const double *a, const double *b, const long n, const bool to_host = false)
  {
    const unsigned nb = grid_for(n, 4);
    if (int e = ensure_scratch(ctx, 2 * 32768 + 8)) // (fused Krylov / stencil kernels leave two partials per block)
      return e;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(VT), 0, ctx->stream, a, b, n,
                       ctx->d_scratch + 8);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(VT), 0, ctx->stream, ctx->d_scratch + 8,
                       (int)nb, ctx->d_scratch, to_host ? ctx->h_result_dev : nullptr);
    return check();
  }

  int launch_dot_to(adaflo_ctx *ctx, const double *a, const double *b, const long n, double *out)
  {
    const unsigned nb = grid_for(n, 4);
    if (int e = ensure_scratch(ctx, 2 * 32768 + 8))
      return e;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(VT), 0, ctx->stream, a, b, n, ctx->d_scratch + 8);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(VT), 0, ctx->stream, ctx->d_scratch + 8, (int)nb, out,
                       (double *)nullptr);
    return check();
  }

  // w -= (*coefficient) v, then out = w . next (next == nullptr: w . w), see gs_step_kernel
  int launch_gs_step(adaflo_ctx *ctx, double *w, const double *coefficient, const double *v, const double *next,
                     const long n, double *out)
  {
    const unsigned nb = grid_for(n, 4);
    if (int e = ensure_scratch(ctx, 2 * 32768 + 8))
      return e;
    hipLaunchKernelGGL(gs_step_kernel, dim3(nb), dim3(VT), 0, ctx->stream, w, coefficient, v, next, n, ctx->d_scratch + 8);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(VT), 0, ctx->stream, ctx->d_scratch + 8, (int)nb, out,
                       (double *)nullptr);
    return check();
  }

  int launch_sum_to(adaflo_ctx *ctx, const double *a, const long n, double *out)
  {
    return launch_dot_to(ctx, a, nullptr, n, out);
  }

  int launch_reciprocal(adaflo_ctx *ctx, double *out, const double *in)
  {
    hipLaunchKernelGGL(reciprocal_kernel, dim3(1), dim3(1), 0, ctx->stream, out, in);
    return check();
  }

  int launch_subtract_scaled(adaflo_ctx *ctx, double *v, const double *s, const double *t, const long n)
  {
    hipLaunchKernelGGL(subtract_scaled_kernel, dim3(grid_for(n)), dim3(VT), 0, ctx->stream, v, s, t, n);
    return check();
  }

  int launch_mean_projection(adaflo_ctx *ctx, double *v, const double *w, const double *modes,
                             const long n, const double inv)
  {
    const unsigned nb = std::min(grid_for(n, 4), 512u); // (every block of the update re-reads the partials)
    if (int e = ensure_scratch(ctx, 2 * 32768 + 8))
      return e;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(VT), 0, ctx->stream, w, v, n, ctx->d_scratch + 8);
    hipLaunchKernelGGL(project_kernel, dim3(grid_for(n, 4)), dim3(VT), 0, ctx->stream, v, modes, ctx->d_scratch + 8,
                       (int)nb, inv, n);
    return check();
  }

  double host_dot(adaflo_ctx *ctx, const double *a, const double *b, const long n)
  {
    if (launch_dot(ctx, a, b, n, true) != 0)
      return 0.;
    (void)hipStreamSynchronize(ctx->stream);
    return ctx->h_result[0];
  }

  int launch_halo_pair(adaflo_ctx *ctx, double *vec0, double *vec1, double *buf, const HaloPlan &plan0, const HaloPlan &plan1,
                       const int mode)
  {
    const long total = plan0.offset[plan0.n_regions] + plan1.offset[plan1.n_regions];
    if (total == 0)
      return 0;
    hipLaunchKernelGGL(halo_pair_kernel, dim3(grid_for(total)), dim3(VT), 0, ctx->stream, vec0, vec1, buf, plan0, plan1, mode);
    return check();
  }

  int launch_halo(adaflo_ctx *ctx, double *vec, double *buf, const HaloPlan &plan, const int mode)
  {
    const long total = plan.offset[plan.n_regions];
    if (total == 0)
      return 0;
    hipLaunchKernelGGL(halo_kernel, dim3(grid_for(total)), dim3(VT), 0, ctx->stream, vec, buf, plan, mode);
    return check();
  }

  namespace
  {
    // engine numbering <-> deal.II numbering through a device-resident index map (adaflo_vector_gather / _scatter)
    __global__ __launch_bounds__(VT) void map_gather_kernel(double *__restrict__ eng, const double *__restrict__ ext,
                                                            const long long *__restrict__ map, const long n)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          const long long m = map[i];
          eng[i] = m >= 0 ? ext[m] : 0.;
        }
    }
    __global__ __launch_bounds__(VT) void map_scatter_kernel(double *__restrict__ ext, const double *__restrict__ eng,
                                                             const long long *__restrict__ map, const long n, const int add)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          const long long m = map[i];
          if (m >= 0)
            ext[m] = (add ? ext[m] : 0.) + eng[i];
        }
    }
  } // namespace

  int launch_map_gather(adaflo_ctx *ctx, double *eng, const double *ext, const long long *map, const long n)
  {
    if (n > 0)
      hipLaunchKernelGGL(map_gather_kernel, dim3(grid_for(n)), dim3(VT), 0, ctx->stream, eng, ext, map, n);
    return check();
  }
  int launch_map_scatter(adaflo_ctx *ctx, double *ext, const double *eng, const long long *map, const long n, const int add)
  {
    if (n > 0)
      hipLaunchKernelGGL(map_scatter_kernel, dim3(grid_for(n)), dim3(VT), 0, ctx->stream, ext, eng, map, n, add);
    return check();
  }

  int launch_sadd(adaflo_ctx *ctx, double *x, const double a, const double *y, const long n)
  {
    hipLaunchKernelGGL(sadd_kernel, dim3(grid_for(n)), dim3(VT), 0, ctx->stream, x, a, y, n);
    return check();
  }

  namespace
  {
    __global__ __launch_bounds__(VT) void lincomb_kernel(double *__restrict__ z, const double a, const double *__restrict__ x,
                                                         const double b, const double *__restrict__ y, const long n)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        z[i] = a * x[i] + b * y[i];
    }
    // system_rhs.sadd(-1., 1., user_rhs) after the cell loop added `sum` to it (:266-293)
    __global__ __launch_bounds__(VT) void residual_finish_kernel(double *__restrict__ rhs, const double *__restrict__ sum,
                                                                 const double *__restrict__ user, const long n)
    {
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        rhs[i] = (user ? user[i] : 0.) - rhs[i] - sum[i];
    }
  } // namespace

  namespace
  {
    __global__ __launch_bounds__(VT) void add_unconstrained_kernel(double *__restrict__ dst, const double *__restrict__ src,
                                                                   const long n_nodes, const int ncomp, const int nnx,
                                                                   const int nny, const int nnz, const uint32_t mask)
    {
      const long n = n_nodes * ncomp;
      for (long i = blockIdx.x * (long)VT + threadIdx.x; i < n; i += (long)gridDim.x * VT)
        {
          const long node = i / ncomp;
          const int  c    = (int)(i - node * ncomp);
          const int  I = node % nnx, J = (node / nnx) % nny, K = node / ((long)nnx * nny);
          if (mask == 0u || !on_constrained_face(I, J, K, nnx, nny, nnz, mask, ncomp == 1 ? 1 : 3, c))
            dst[i] += src[i];
        }
    }
  } // namespace

  int launch_add_unconstrained(adaflo_ctx *ctx, double *dst, const double *src, const long n_nodes, const int ncomp,
                               const int nnx, const int nny, const int nnz, const uint32_t mask)
  {
    hipLaunchKernelGGL(add_unconstrained_kernel, dim3(grid_for(n_nodes * ncomp)), dim3(VT), 0, ctx->stream, dst, src,
                       n_nodes, ncomp, nnx, nny, nnz, mask);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_lincomb(adaflo_ctx *ctx, double *z, const double a, const double *x, const double b, const double *y,
                     const long n)
  {
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n, 4)), dim3(VT), 0, ctx->stream, z, a, x, b, y, n);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_residual_finish(adaflo_ctx *ctx, double *rhs, const double *sum, const double *user, const long n)
  {
    hipLaunchKernelGGL(residual_finish_kernel, dim3(grid_for(n, 4)), dim3(VT), 0, ctx->stream, rhs, sum, user, n);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_fill(adaflo_ctx *ctx, double *x, const double v, const long n)
  {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(VT), 0, ctx->stream, x, v, n);
    return check();
  }

  int launch_transpose_state(adaflo_ctx *ctx, double *dst, const double *src, const long n_cells,
                             const int nq, const int ncomp, const bool to_generic)
  {
    hipLaunchKernelGGL(transpose_state_kernel, dim3(grid_for(n_cells * nq * ncomp)), dim3(VT), 0,
                       ctx->stream, dst, src, n_cells, nq, ncomp, to_generic);
    return check();
  }
} // namespace adaflo_hip

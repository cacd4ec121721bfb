// Small-batch critic tail (batch <= 64, two hidden layers of width <= 256): at batch 32 the hidden
// part of the critic is a chain of tiny dependent launches (split-K reduce, Dense+ReLU, Dense(1),
// backward seed; gradient-penalty finalisation, reduce, second-order Dense) whose cost is launch
// latency, not work.  Rows are independent through all of them, so ONE workgroup carries up to four
// rows from the first-layer slabs to the last value the next big product needs:
//
//   mode 0 (tf_utils.py:132-135, :165-167, :213-214): A1 = act(sum_z slab_z [+ interpolation by
//           linearity] + b1); A2 = act(A1 . W2 + b2); out = A2 . w3 + b3; D2 = act'(A2) * c(row) * w3
//   mode 1 (tf_utils.py:168-170 and its derivative): s = sqrt(sum rowsq) (no epsilon);
//           f = coef*(s-1)/s; D1 *= f; E1 = act'(A1) * f * sum_z slab_z  (in place over A1);
//           E2 = act'(A2) * (E1 . W2)  (in place over A2); covariate columns of the rows zeroed
//           (Gram form, TailArgs::gram: the slab is d_1 . W_g^T W_g and s^2 its dot product with d_1)
//   mode 3 (generator step, tf_utils.py:255 through the two hidden layers): d2 = act'(h2) * sum_z slab_z (slabs of
//           dX . W_out^T); d1 = act'(h1) * (d2 . W_2^T) - the split-K reduce and the hidden delta product in one launch
//
// fp32 VALU (4 rows x 256 x 256 MACs per workgroup: nothing for MFMA to win); fixed summation
// orders (deterministic).
//
// NS > 1 (few row groups, i.e. small batches): the hidden Dense of a row group is split by output columns over NS
// workgroups (blockIdx.y).  A workgroup pulling all of W2 (256 KB) is bound by its CU's 64 B/clk (1.7 us) and by four
// dependent batches of 16 loads; a quarter of the columns is 64 KB in ONE batch.  Every workgroup of the group repeats
// the cheap part (slab sums -> A1 / E1 in LDS); only workgroup 0 writes the shared results.  The row's critic output
// then arrives as NS partial dot products out[ns][row] (bias in part 0) that the loss job adds in order.  Mode 1
// overwrites A1 (whose sign is the ReLU mask it needs) in place: with NS > 1 it reads the mask from a copy of the
// interpolated rows' A1 that mode 0 leaves in A1m, so that workgroup 0's store cannot race the others' loads.
#include "agx_common.h"

#include <algorithm>
#include <cstdlib>

namespace {

using agx::wave_sum;

constexpr int TW = 256;  // max hidden width
constexpr int TR = 4;    // rows per workgroup

template <int MODE, int NS>
__global__ void __launch_bounds__(1024) critic_tail_kernel(const agx::TailArgs p) {
  constexpr int HC = TW / NS;      // output columns of the hidden Dense per workgroup
  constexpr int KQ = 4 * NS;       // thread groups over k in the Dense loop (1024 threads = KQ x HC)
  __shared__ float raw[TR][TW];
  __shared__ float4 a1[TW];   // [k] -> the four rows' values: one ds_read_b128 per k in the Dense loop
  __shared__ float part[KQ * TR * HC];   // [q][row][column]: 4096 floats whatever NS
  __shared__ float red[16];
  __shared__ float xin[MODE == 2 ? TR * 128 : 1];   // mode 2: the workgroup's input rows
  const int t = threadIdx.x, i = t >> 8, c = t & 255, lane = t & 63, wave = t >> 6;
  const int ns = NS > 1 ? (int)blockIdx.y : 0;
  const bool lead = ns == 0;       // the workgroup that writes what the column groups share
  const bool interp = MODE == 0 && p.interp_rows > 0;
  const int R = p.interp_rows;
  // slab rows / output rows of this workgroup
  const int r0 = interp ? (int)blockIdx.x : (int)blockIdx.x * p.rpw;
  const int nsl = interp ? 2 : min(p.rpw, p.m_slab - r0);
  const int nout = interp ? 3 : nsl;
  auto srow = [&](int k) { return interp ? r0 + k * R : r0 + k; };
  auto orow = [&](int k) { return interp ? r0 + k * R : p.row0 + r0 + k; };

  float f = 1.f;
  if (MODE == 1 && !p.gram) {
    // slopes from the per-slab partial sums of squares (same order as gp_finalize_kernel)
    float s = 0.f;
    if (i < nsl) {
      const float *q = p.rowsq + (size_t)srow(i) * p.rowsq_ld;
      for (int j = c; j < p.nslab; j += 256) s += q[j];
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float sl = sqrtf(red[4 * i] + red[4 * i + 1] + red[4 * i + 2] + red[4 * i + 3]);
    f = p.coef * (sl - 1.f) / sl;  // sl == 0 -> NaN propagates like the reference
    if (lead && i < nsl && c == 0) {
      p.slopes[srow(i)] = sl;
      p.fscale[srow(i)] = f;
    }
    __syncthreads();
  }

  // ---- first-layer slabs -> rows of layer-1 activations (mode 0) / second-order deltas (mode 1)
  // (one thread per (row, column); a variant with the slabs dealt over all four thread groups and
  // 16 loads in flight per thread measured twice as slow)
  float v = 0.f;
  if (MODE == 2) {
    // first hidden layer straight from the input rows (K_in <= 128): rows in LDS, W1 read coalesced over columns,
    // sixteen loads in flight; k in order (one chain per element: deterministic, independent of the launch shape)
    for (int q = t; q < TR * p.K_in; q += 1024) {
      const int r = q / p.K_in, k = q - r * p.K_in;
      xin[r * 128 + k] = r < nsl ? p.in[(size_t)srow(r) * p.ld_in + k] : 0.f;
    }
    __syncthreads();
    if (i < nsl && c < p.H1) {
      const float *__restrict__ w = p.W1 + c;
      const float *xr = xin + i * 128;
      int k = 0;
      for (; k + 16 <= p.K_in; k += 16) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)(k + u) * p.ldw1];
#pragma unroll
        for (int u = 0; u < 16; ++u) v = fmaf(xr[k + u], wv[u], v);
      }
      for (; k < p.K_in; ++k) v = fmaf(xr[k], w[(size_t)k * p.ldw1], v);
    }
  } else if (MODE == 0 && interp && p.ws2 && i == 1) {
    if (c < p.H1) v = p.ws2[(size_t)r0 * p.H1 + c];   // factored fake row: finished value, no slabs
  } else if (i < nsl && c < p.H1) {
    const size_t stride = (size_t)p.H1;   // slabs [row][z][H1]: a row's partials are contiguous
    const float *s0 = p.ws + (size_t)srow(i) * p.splitk * p.H1 + c;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    int z = 0;
    // sixteen independent loads in flight (same summation order as two batches of eight: bit-identical;
    // 0.5177 against 0.5224 ms per 32-row iteration)
    for (; z + 16 <= p.splitk; z += 16) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = s0[(size_t)(z + u) * stride];
      q0 += x[0]; q1 += x[1]; q2 += x[2]; q3 += x[3];
      q0 += x[4]; q1 += x[5]; q2 += x[6]; q3 += x[7];
      q0 += x[8]; q1 += x[9]; q2 += x[10]; q3 += x[11];
      q0 += x[12]; q1 += x[13]; q2 += x[14]; q3 += x[15];
    }
    for (; z + 8 <= p.splitk; z += 8) {   // eight independent loads in flight
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = s0[(size_t)(z + u) * stride];
      q0 += x[0]; q1 += x[1]; q2 += x[2]; q3 += x[3];
      q0 += x[4]; q1 += x[5]; q2 += x[6]; q3 += x[7];
    }
    for (; z < p.splitk; ++z) q0 += s0[(size_t)z * stride];
    v = (q0 + q1) + (q2 + q3);
  }
  if (MODE == 2) {
    float a = 0.f;
    if (i < nsl && c < p.H1) {
      const float x = v + p.b1[c];
      a = x > 0.f ? x : x * p.slope;
      if (lead) p.A1[(size_t)orow(i) * p.lda1 + c] = a;
    }
    reinterpret_cast<float *>(&a1[c])[i] = a;
  } else if (MODE == 0) {
    raw[i][c] = v;
    __syncthreads();
    float a = 0.f;
    if (i < nout && c < p.H1) {
      float x;
      if (interp && i == 2) {
        const float r = raw[0][c], fk = raw[1][c];
        x = r + p.alpha[r0] * (fk - r);
      } else {
        x = raw[i][c];
      }
      x += p.b1[c];
      a = x > 0.f ? x : x * p.slope;
      if (lead) {
        p.A1[(size_t)orow(i) * p.lda1 + c] = a;
        if (interp && i == 2 && p.A1m) p.A1m[(size_t)r0 * p.lda1 + c] = a;   // mask copy for a column-split mode 1
      }
    }
    reinterpret_cast<float *>(&a1[c])[i] = a;
  } else if (MODE == 3) {
    float e = 0.f;
    if (i < nsl && c < p.H1) {
      const float mk = p.M1[(size_t)srow(i) * p.ldm1 + c];
      e = mk > 0.f ? v : v * p.slope;
      if (lead) p.A1[(size_t)orow(i) * p.lda1 + c] = e;
    }
    reinterpret_cast<float *>(&a1[c])[i] = e;
  } else {
    float d1v = 0.f;
    if (p.gram) {
      // Gram form: v = (d_1 . W_g^T W_g)[row, c], so |g|^2 = |d_1 . W_g^T|^2 = sum_c v * d_1[row, c]   (:168-169)
      if (i < nsl && c < p.H1) d1v = p.D1[(size_t)orow(i) * p.ldd1 + c];
      const float s = wave_sum(v * d1v);
      if (lane == 0) red[wave] = s;
      __syncthreads();
      const float sl = sqrtf((red[4 * i] + red[4 * i + 1]) + (red[4 * i + 2] + red[4 * i + 3]));
      f = p.coef * (sl - 1.f) / sl;  // sl == 0 -> NaN propagates like the reference
      if (lead && i < nsl && c == 0) {
        p.slopes[srow(i)] = sl;
        p.fscale[srow(i)] = f;
      }
    }
    float e = 0.f;
    if (i < nsl && c < p.H1) {
      const size_t o = (size_t)orow(i) * p.lda1 + c;
      const float x = v * f;
      const float mk = (NS > 1) ? p.A1m[(size_t)srow(i) * p.lda1 + c] : p.A1[o];
      e = mk > 0.f ? x : x * p.slope;
      if (lead) {
        p.A1[o] = e;
        if (p.gram) {
          p.gU[(size_t)srow(i) * p.ldg + c] = d1v;
          p.gF[(size_t)srow(i) * p.ldg + c] = d1v * f;
        } else {
          float *d = p.D1 + (size_t)orow(i) * p.ldd1 + c;
          *d = *d * f;
        }
      }
    }
    if (lead && i < nsl && c < p.ncov) p.xcov[(size_t)srow(i) * p.ldx + c] = 0.f;
    reinterpret_cast<float *>(&a1[c])[i] = e;
  }
  __syncthreads();

  // ---- hidden Dense: k split over the four 256-thread groups, W2 rows read coalesced, once
  {
    const int jl = t % HC, q = t / HC, j = ns * HC + jl;
    float acc[TR] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 3) {
      // transposed kernel: output j = row j of W2 [H2, H1] against the rows' deltas; this thread's k range of that
      // row, contiguous: 16-byte loads, all of them in flight
      if (j < p.H2) {
        const int kq = (p.H1 + KQ - 1) / KQ;
        const int k0 = q * kq, k1 = min(p.H1, k0 + kq);
        const float *__restrict__ w = p.W2 + (size_t)j * p.ldw2;
        int k = k0;
        if ((p.ldw2 & 3) == 0 && (k0 & 3) == 0) {
          for (; k + 16 <= k1; k += 16) {
            float4 wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = *reinterpret_cast<const float4 *>(w + k + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float ws4[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float4 av = a1[k + 4 * u + e];
                acc[0] = fmaf(av.x, ws4[e], acc[0]);
                acc[1] = fmaf(av.y, ws4[e], acc[1]);
                acc[2] = fmaf(av.z, ws4[e], acc[2]);
                acc[3] = fmaf(av.w, ws4[e], acc[3]);
              }
            }
          }
        }
        for (; k < k1; ++k) {
          const float wv = w[k];
          const float4 av = a1[k];
          acc[0] = fmaf(av.x, wv, acc[0]);
          acc[1] = fmaf(av.y, wv, acc[1]);
          acc[2] = fmaf(av.z, wv, acc[2]);
          acc[3] = fmaf(av.w, wv, acc[3]);
        }
      }
    } else if (j < p.H2) {
      const int kq = (p.H1 + KQ - 1) / KQ;
      const int k0 = q * kq, k1 = min(p.H1, k0 + kq);
      const float *__restrict__ w = p.W2 + (size_t)k0 * p.ldw2 + j;
      int k = k0;
      // (the first batch requested at kernel start, in front of or right behind the slab loads, so that the two
      // round trips overlap: 0.4957 against 0.4925 ms per 32-row iteration, 2.711 against 2.696 ms at 512 rows)
      // 16 weight loads in flight per thread: the k loop is a chain of L2 round trips, not arithmetic
      // (batches of 8 took 8 round trips per quarter of a 256-wide layer; batches of 32 gain nothing at 32 rows
      // and lose 0.5 % at 512, batches of 64 lose 6 %)
      for (; k + 16 <= k1; k += 16) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)u * p.ldw2];
        w += (size_t)16 * p.ldw2;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float4 av = a1[k + u];
          acc[0] = fmaf(av.x, wv[u], acc[0]);
          acc[1] = fmaf(av.y, wv[u], acc[1]);
          acc[2] = fmaf(av.z, wv[u], acc[2]);
          acc[3] = fmaf(av.w, wv[u], acc[3]);
        }
      }
      for (; k < k1; ++k) {
        const float wv = *w;
        w += p.ldw2;
        const float4 av = a1[k];
        acc[0] = fmaf(av.x, wv, acc[0]);
        acc[1] = fmaf(av.y, wv, acc[1]);
        acc[2] = fmaf(av.z, wv, acc[2]);
        acc[3] = fmaf(av.w, wv, acc[3]);
      }
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) part[(q * TR + r) * HC + jl] = acc[r];
  }
  __syncthreads();
  // finish: thread (row fi, column fl of this workgroup's HC columns); NS = 1 is the old (i, c) numbering and order
  const int fi = t / HC, fl = t % HC, fc = ns * HC + fl;
  float hp = 0.f;
  if (t < TR * HC && fi < nout && fc < p.H2) {
    auto pq = [&](int q) { return part[(q * TR + fi) * HC + fl]; };
    float x = (pq(0) + pq(1)) + (pq(2) + pq(3));
#pragma unroll
    for (int q = 4; q < KQ; q += 4) x += (pq(q) + pq(q + 1)) + (pq(q + 2) + pq(q + 3));
    const int i = fi, c = fc;
    const size_t o = (size_t)orow(i) * p.lda2 + c;
    if (MODE == 2) {
      x += p.b2[c];
      p.A2[o] = x > 0.f ? x : x * p.slope;
    } else if (MODE == 0) {
      x += p.b2[c];
      const float a = x > 0.f ? x : x * p.slope;
      p.A2[o] = a;
      const int row = orow(i);
      const float cc = row < p.n0 ? p.c0 : (row < p.n0 + p.n1 ? p.c1 : p.c2);
      const float wv = p.w3[c], dv = cc * wv;
      p.D2[(size_t)row * p.ldd2 + c] = a > 0.f ? dv : dv * p.slope;
      hp = a * wv;
    } else if (MODE == 3) {
      const float mk = p.M2[(size_t)srow(i) * p.ldm2 + c];
      p.A2[o] = mk > 0.f ? x : x * p.slope;
    } else {
      p.A2[o] = p.A2[o] > 0.f ? x : x * p.slope;
    }
  }
  if (MODE == 0) {
    hp = wave_sum(hp);
    if (lane == 0) red[wave] = hp;
    __syncthreads();
    if (NS == 1) {
      if (i < nout && c == 0) p.out[orow(i)] = (red[4 * i] + red[4 * i + 1]) + (red[4 * i + 2] + red[4 * i + 3]) + p.b3[0];
    } else if (t < TR * HC && fi < nout && fl == 0) {
      constexpr int WR = HC / 64;   // waves per row
      float s = red[fi * WR];
#pragma unroll
      for (int w = 1; w < WR; ++w) s += red[fi * WR + w];
      p.out[(size_t)ns * p.out_ld + orow(fi)] = lead ? s + p.b3[0] : s;
    }
  }
}

}  // namespace

int agx::critic_tail(void *stream, int mode, const agx::TailArgs &a_in) {
  agx::TailArgs a = a_in;
  AGX_CHECK_ARG((mode == 2 || (a.ws && a.splitk >= 1)) && a.m_slab >= 1 && a.H1 >= 1 && a.H1 <= TW && a.H2 >= 1 && a.H2 <= TW &&
                    a.W2 && a.ldw2 >= (mode == 3 ? a.H1 : a.H2) && a.A1 && a.A2 && a.lda1 >= a.H1 && a.lda2 >= a.H2,
                "critic_tail: bad shapes (H1=%d H2=%d m=%d)", a.H1, a.H2, a.m_slab);
  AGX_CHECK_ARG(a.nsplit == 0 || a.nsplit == 1 || mode == 2 || mode == 3 || ((a.nsplit == 2 || a.nsplit == 4) && a.out_ld > 0 && (mode == 0 || a.A1m)),
                "critic_tail: column split %d needs out_ld and the mask copy", a.nsplit);
  hipStream_t st = agx::as_stream(stream);
  // rows per workgroup: every workgroup pulls all of W2 (256 KB at 64 B/clk: 1.7 us) whatever its row count, so
  // few rows are spread over more CUs (their slab traffic per CU shrinks); many rows keep 4 per workgroup
  // (4 rows per workgroup at small batches measured slower; the knob is gone)
  a.rpw = (a.interp_rows == 0 && a.m_slab <= 64) ? 2 : TR;
  if (mode == 2) {
    AGX_CHECK_ARG(a.in && a.W1 && a.b1 && a.b2 && a.K_in >= 1 && a.K_in <= 128 && a.ld_in >= a.K_in && a.ldw1 >= a.H1 &&
                      (a.nsplit == 1 || a.nsplit == 2 || a.nsplit == 4),
                  "critic_tail: two-layer forward arguments (K_in=%d)", a.K_in);
    a.rpw = TR;
    a.interp_rows = 0;
    a.row0 = 0;
    agx::ProfScope prof(st, AGX_K_GEN_FWD_HID, 2.0 * a.m_slab * ((double)a.K_in * a.H1 + (double)a.H1 * a.H2),
                        4.0 * ((double)a.m_slab * a.K_in + (double)a.K_in * a.H1 + (double)a.H1 * a.H2 + (double)a.m_slab * (a.H1 + a.H2)));
    const int blocks = agx::cdiv(a.m_slab, a.rpw);
    if (a.nsplit == 4) hipLaunchKernelGGL((critic_tail_kernel<2, 4>), dim3(blocks, 4), dim3(1024), 0, st, a);
    else if (a.nsplit == 2) hipLaunchKernelGGL((critic_tail_kernel<2, 2>), dim3(blocks, 2), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((critic_tail_kernel<2, 1>), dim3(blocks), dim3(1024), 0, st, a);
  } else if (mode == 3) {
    AGX_CHECK_ARG(a.M1 && a.M2 && a.ldm1 >= a.H1 && a.ldm2 >= a.H2 && agx::aligned16(a.W2) &&
                      (a.nsplit == 1 || a.nsplit == 2 || a.nsplit == 4),
                  "critic_tail: generator backward arguments");
    a.interp_rows = 0;
    a.row0 = 0;
    agx::ProfScope prof(st, AGX_K_GEN_BWD_HID, 2.0 * a.m_slab * a.H1 * (double)a.H2,
                        4.0 * ((double)a.splitk * a.m_slab * a.H1 + (double)a.H1 * a.H2));
    const int blocks = agx::cdiv(a.m_slab, a.rpw);
    if (a.nsplit == 4) hipLaunchKernelGGL((critic_tail_kernel<3, 4>), dim3(blocks, 4), dim3(1024), 0, st, a);
    else if (a.nsplit == 2) hipLaunchKernelGGL((critic_tail_kernel<3, 2>), dim3(blocks, 2), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((critic_tail_kernel<3, 1>), dim3(blocks), dim3(1024), 0, st, a);
  } else if (mode == 0) {
    AGX_CHECK_ARG(a.b1 && a.b2 && a.w3 && a.b3 && a.D2 && a.out && a.ldd2 >= a.H2 &&
                      (a.interp_rows == 0 || (a.alpha && a.m_slab == (a.ws2 ? 1 : 2) * a.interp_rows)),
                  "critic_tail: forward arguments");
    agx::ProfScope prof(st, AGX_K_DISC_FWD_HID, 2.0 * (a.interp_rows ? 3 * a.interp_rows : a.m_slab) * a.H1 * (double)a.H2,
                        4.0 * ((double)a.splitk * a.m_slab * a.H1 + (double)a.H1 * a.H2));
    const int blocks = a.interp_rows > 0 ? a.interp_rows : agx::cdiv(a.m_slab, a.rpw);
    if (a.nsplit == 4) hipLaunchKernelGGL((critic_tail_kernel<0, 4>), dim3(blocks, 4), dim3(1024), 0, st, a);
    else if (a.nsplit == 2) hipLaunchKernelGGL((critic_tail_kernel<0, 2>), dim3(blocks, 2), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((critic_tail_kernel<0, 1>), dim3(blocks), dim3(1024), 0, st, a);
  } else {
    AGX_CHECK_ARG(a.slopes && a.fscale && a.D1 && a.ldd1 >= a.H1 && a.ncov >= 0 && a.ncov <= 256 && (a.ncov == 0 || a.xcov) &&
                      (a.gram ? (a.splitk == 1 && a.gU && a.gF && a.ldg >= a.H1)
                              : (a.rowsq && a.nslab >= 1 && a.rowsq_ld >= a.nslab)),
                  "critic_tail: second-order arguments");
    agx::ProfScope prof(st, AGX_K_GP2_HID, 2.0 * a.m_slab * a.H1 * (double)a.H2,
                        4.0 * ((double)a.splitk * a.m_slab * a.H1 + (double)a.H1 * a.H2));
    const int blocks = agx::cdiv(a.m_slab, a.rpw);
    if (a.nsplit == 4) hipLaunchKernelGGL((critic_tail_kernel<1, 4>), dim3(blocks, 4), dim3(1024), 0, st, a);
    else if (a.nsplit == 2) hipLaunchKernelGGL((critic_tail_kernel<1, 2>), dim3(blocks, 2), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((critic_tail_kernel<1, 1>), dim3(blocks), dim3(1024), 0, st, a);
  }
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

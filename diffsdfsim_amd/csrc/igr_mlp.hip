// igr_mlp.hip -- IGR neural SDF (decode_igr, sdf_physics/physics3d/utils.py:330-350) on gfx950 fp64 MFMA.
//
// Network (IGR_data/train_configs/bob_spot_setup.conf:38-45; ImplicitNet of the external IGR repo, restated from
// its published definition): input = [latent(2), xyz(3)], 9 Linear layers 5->128->128->128->123, skip concat
// [h, input]/sqrt(2) before layer 4, ->128 x4 ->1, Softplus(beta = 100) on all but the last, float64.
// query_sdfs needs phi AND d phi/d xyz (the reference gets the latter from autograd, bodies.py:730-745).
//
// This is the one dense-GEMM shaped piece of the hot path, so it runs on the matrix cores:
//   * MODE_XYZ / MODE_LATENT, forward-mode: every point carries (h, three tangents); the 16-row MFMA tile holds 4 points
//     x 4 quantities, ordered row = 4*quantity + point.  With v_mfma_f64_16x16x4 a lane's four accumulator registers
//     are rows q, q+4, q+8, q+12 of one column (q = lane>>4): exactly (value, t0, t1, t2) of ONE point and ONE
//     neuron, so bias + softplus + sigmoid-scaling of the tangents is lane-local.  No activations are kept for a
//     backward sweep and no transposed weights are needed.
//   * MODE_VALUE: the 16 rows of a tile are 16 points (values only: the candidate test of the narrow phase and the
//     Laplacian probes need no gradient -- a quarter of the work).
//   * a workgroup of NW waves owns NG row groups (4 NG points with tangents, 16 NG without); wave w computes neuron tiles
//     [w*8/NW, (w+1)*8/NW) for all of them (NG x 8/NW accumulator tiles in AGPRs).  Activations of the current layer live
//     in LDS ([16 NG rows][128+1]), weights stream from L2 in MFMA-fragment order (host-packed: one coalesced 512-B load
//     per fragment, reused by all NG row groups), software-pipelined with the k-loop fully unrolled so ~60 fragment loads
//     are in flight.  Two workgroups' waves share a SIMD, so one's softplus epilogue (VALU) hides under the other's MFMAs.
//   * the grid is persistent over the tiles of a point list whose length may live in device memory (the query rounds of
//     the neural narrow phase, narrowphase_igr.hip, fill such lists); every point names its latent code by an index.
// Flops: 4 x 115 456 MAC per point = 0.92 MFLOP (fp64) with tangents; matrix peak 256 CU x 128 flop/clk x 2.4 GHz = 78.6 TFLOP/s.
// Measured (128^3 grid, MI355X): 38.0 ms = 50.9 TFLOP/s = 65 % of peak.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "mfma_f64.h"
#include "wave_utils.h"

namespace {
using namespace dss;

constexpr int H = 128, NL = 9, DIN = 5, LDX = H + 1;
enum { MODE_XYZ = DSS_IGR_XYZ, MODE_LATENT = DSS_IGR_LATENT, MODE_VALUE = DSS_IGR_VALUE };

__device__ inline acc4 mfma(double a, double b, acc4 c) { return mfma_f64_16x16x4(a, b, c); }
__device__ inline double &comp(acc4 &v, int i) { return acc_comp(v, i); }

// softplus(z, beta=100, threshold=20) and its derivative, torch.nn.Softplus semantics:
//   h = z (100 z > 20), else log1p(exp(100 z)) / 100;   dh = sigmoid(100 z).
// Evaluated as max(z, 0) + log1p(u) / 100 with u = exp(-|100 z|) in (0, 1], by hand: the network spends 1024 of these per
// point on the vector ALU next to the matrix work, and the library's exp + log1p + quotient are general-purpose (every range,
// every special case) where this needs one argument range each.
//   exp(-a):   a = k ln2 + r, |r| <= ln2 / 2, Taylor to r^12 (remainder 2e-16), ldexp
//   log1p(u):  w = 1 + u, f = w or w / 2 in (1/sqrt 2, sqrt 2], s = (f - 1) / (f + 1), log f = 2 s (1 + s^2/3 + ... + s^20/21)
//              (|s| <= 0.1716: remainder 2e-17), plus (u - (w - 1)) / w for the bits of u the sum 1 + u rounded away
// Absolute error of h below 3e-18 + 2 ulp over the whole range (tests/emu/check_softplus.cpp against long double).
__device__ inline void softplus100(double z, double &h, double &dh)
{
    const double y = 100.0 * z;
    const bool lin = y > 20.0;
    const double a = fabs(lin ? 20.0 : y);
    // u = exp(-a)
    const double kf = __builtin_rint(-a * 1.4426950408889634074);
    double r = __builtin_fma(-kf, 6.93147180369123816490e-01, -a);
    r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double u = ldexp(p, (int)kf);
    // log1p(u)
    const double w = 1.0 + u, lo = u - (w - 1.0);
    const bool big = w > 1.4142135623730951;
    const double f = big ? 0.5 * w : w;
    const double s = (f - 1.0) * dss_rcp(f + 1.0), s2 = s * s;
    double q = 2.0 / 21.0;
    q = __builtin_fma(q, s2, 2.0 / 19.0);
    q = __builtin_fma(q, s2, 2.0 / 17.0);
    q = __builtin_fma(q, s2, 2.0 / 15.0);
    q = __builtin_fma(q, s2, 2.0 / 13.0);
    q = __builtin_fma(q, s2, 2.0 / 11.0);
    q = __builtin_fma(q, s2, 2.0 / 9.0);
    q = __builtin_fma(q, s2, 2.0 / 7.0);
    q = __builtin_fma(q, s2, 2.0 / 5.0);
    q = __builtin_fma(q, s2, 2.0 / 3.0);
    q = __builtin_fma(q, s2, 2.0);
    const double rw = dss_rcp(w);
    double l1p = __builtin_fma(q, s, big ? 6.93147180559945286227e-01 : 0.0);
    l1p = __builtin_fma(lo, rw, l1p);
    h = lin ? z : (y > 0.0 ? z : 0.0) + l1p * 0.01;
    dh = lin ? 1.0 : (y >= 0.0 ? rw : u * rw);
}

struct Query {
    const double *pts;       // [n][3] points in the network's unit frame
    const int *lat_idx;      // [n] index of the point's latent code, or NULL = code 0
    const double *latents;   // [.][lat_stride] latent codes (first two entries of a row)
    int lat_stride;
    const int *n_dev;        // length of the list in device memory, or NULL = n
    int n;
    double *sdf, *grad;      // [n], [n][3] (grad unused in MODE_VALUE)
};

// Wp: packed weights.  Per hidden->hidden layer (7 of them): [tile t 0..7][kstep 0..31][lane 0..63] = W[16t + (lane&15)][4ks + (lane>>4)]
// NW waves per workgroup share the activations of NG row groups; wave w owns neuron tiles [w*8/NW, (w+1)*8/NW)
// bid / nblk: this workgroup's place among the workgroups that walk Q's tiles (a launch may serve two lists, see below)
template <int NW, int NG, int MODE> __device__ __forceinline__ void
igr_body(const Query &Q, int bid, int nblk, double *X, const double *W0, const double *b0, const double *Wp, const double *bh,
         const double *W8, const double *b8)
{
    constexpr bool TAN = MODE != MODE_VALUE;
    constexpr int PTS = (TAN ? 4 : 16) * NG, ROWS = 16 * NG, NT = 64 * NW, TPW = 8 / NW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int n = Q.n_dev ? *Q.n_dev : Q.n;
    // row of (point p of the tile, quantity) and back
    auto row_of = [](int p, int quant) { return TAN ? 16 * (p / 4) + 4 * quant + (p % 4) : p; };

    for (int base = bid * PTS; base < n; base += nblk * PTS) {
        // ---- layer 0 (K = 5) on the vector ALU: h0 = softplus(W0 [latent, xyz] + b0), tangents = sigma' * W0[:, seed]
        for (int e = tid; e < PTS * H; e += NT) {
            const int p = e / H, j = e % H, gp = base + p;
            double in[DIN] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (gp < n) {
                const double *lat = Q.latents + (size_t)(Q.lat_idx ? Q.lat_idx[gp] : 0) * Q.lat_stride;
                in[0] = lat[0]; in[1] = lat[1];
                in[2] = Q.pts[3 * gp]; in[3] = Q.pts[3 * gp + 1]; in[4] = Q.pts[3 * gp + 2];
            }
            double z = b0[j];
            for (int k = 0; k < DIN; ++k) z += W0[j * DIN + k] * in[k];
            double h, dh;
            softplus100(z, h, dh);
            X[row_of(p, 0) * LDX + j] = h;
            if (TAN) {
                // tangent seeds: the three point coordinates (inputs 2..4), or the two latent coordinates (inputs 0, 1)
                for (int d = 0; d < 3; ++d)
                    X[row_of(p, d + 1) * LDX + j] = MODE == MODE_LATENT ? (d < 2 ? dh * W0[j * DIN + d] : 0.0) : dh * W0[j * DIN + 2 + d];
            }
        }
        __syncthreads();

        for (int layer = 1; layer < NL - 1; ++layer) {
            const double *Wl = Wp + (size_t)(layer - 1) * 8 * 32 * 64, *bl = bh + (size_t)(layer - 1) * H;
            if (layer == 4) {
                // skip connection: x = cat([h3 (123), input (5)]) / sqrt(2)   (value rows get the input, tangent rows its Jacobian)
                for (int e = tid; e < ROWS * H; e += NT) {
                    const int r = e / H, j = e % H;
                    double v = X[r * LDX + j];
                    if (j >= H - DIN) {
                        const int k = j - (H - DIN);
                        const int quant = TAN ? (r % 16) / 4 : 0, p = TAN ? 4 * (r / 16) + (r % 4) : r, gp = base + p;
                        if (quant == 0) {
                            v = 0.0;
                            if (gp < n) {
                                const double *lat = Q.latents + (size_t)(Q.lat_idx ? Q.lat_idx[gp] : 0) * Q.lat_stride;
                                v = k < 2 ? lat[k] : Q.pts[3 * gp + k - 2];
                            }
                        } else {
                            const bool wl = MODE == MODE_LATENT;
                            v = ((wl ? k : k - 2) == quant - 1 && (wl ? k < 2 : k >= 2)) ? 1.0 : 0.0;
                        }
                    }
                    X[r * LDX + j] = v * 0.70710678118654752440;
                }
                __syncthreads();
            }
            acc4 acc[NG][TPW];
            for (int g = 0; g < NG; ++g)
                for (int t = 0; t < TPW; ++t) { comp(acc[g][t], 0) = 0; comp(acc[g][t], 1) = 0; comp(acc[g][t], 2) = 0; comp(acc[g][t], 3) = 0; }
            // software pipeline: the B fragments (and A) of k-step ks+1 are in flight while the MFMAs of ks issue
            double bq[2][TPW], aq[2][NG], bias[TPW];
            const double *Ww = Wl + (size_t)wv * TPW * 32 * 64;
#pragma unroll
            for (int t = 0; t < TPW; ++t) bias[t] = bl[16 * (wv * TPW + t) + col];
#pragma unroll
            for (int t = 0; t < TPW; ++t) bq[0][t] = Ww[(size_t)t * 32 * 64 + lane];
#pragma unroll
            for (int g = 0; g < NG; ++g) aq[0][g] = X[(16 * g + col) * LDX + q];
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < 32) {
#pragma unroll
                    for (int t = 0; t < TPW; ++t) bq[nxt][t] = Ww[((size_t)t * 32 + ks + 1) * 64 + lane];
                    // A fragments: lane holds X[row = lane&15 (+16 g)][k = 4 ks + (lane>>4)]
#pragma unroll
                    for (int g = 0; g < NG; ++g) aq[nxt][g] = X[(16 * g + col) * LDX + 4 * (ks + 1) + q];
                }
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][t] = mfma(aq[cur][g], bq[cur][t], acc[g][t]);
            }
            __syncthreads();
            // epilogue.  With tangents this lane owns (value, t0, t1, t2) of point q (of group g) for neuron 16 t + col;
            // without, the values of points q, q+4, q+8, q+12 of the group
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int j = 16 * (wv * TPW + t) + col;
                    double h, dh;
                    if (TAN) {
                        softplus100(comp(acc[g][t], 0) + bias[t], h, dh);
                        X[(16 * g + q) * LDX + j] = h;
                        for (int d = 1; d < 4; ++d) X[(16 * g + 4 * d + q) * LDX + j] = dh * comp(acc[g][t], d);
                    } else {
                        for (int d = 0; d < 4; ++d) {
                            softplus100(comp(acc[g][t], d) + bias[t], h, dh);
                            X[(16 * g + 4 * d + q) * LDX + j] = h;
                        }
                    }
                }
            __syncthreads();
        }
        // ---- layer 8 (one output): dot products on the vector ALU, rows split over lanes
        for (int r = tid; r < ROWS; r += NT) {
            double acc = 0.0;
            for (int j = 0; j < H; ++j) acc += W8[j] * X[r * LDX + j];
            const int quant = TAN ? (r % 16) / 4 : 0, p = TAN ? 4 * (r / 16) + (r % 4) : r, gp = base + p;
            if (gp < n) {
                if (quant == 0) Q.sdf[gp] = acc + b8[0];
                else Q.grad[3 * gp + quant - 1] = acc;
            }
        }
        __syncthreads();   // the next tile's layer 0 overwrites X
    }
}
template <int NW, int NG, int MODE> __global__ void __launch_bounds__(64 * NW)
igr_query_kernel(Query Q, const double *W0, const double *b0, const double *Wp, const double *bh, const double *W8, const double *b8)
{
    DSS_DYN_LDS(double, X);   // [ROWS][LDX]; with tangents row = 16 g + 4*quantity + point, without row = point
    igr_body<NW, NG, MODE>(Q, (int)blockIdx.x, (int)gridDim.x, X, W0, b0, Wp, bh, W8, b8);
}
// One launch for the two lists of a query round of the neural narrow phase: workgroups [0, split) walk the value list, the
// rest the gradient list.  Both lists are short most of the time (a Frank-Wolfe round: a few points per item), so each on
// its own is a launch whose duration is the latency of one tile through nine layers, with most of the chip idle -- and
// there are 42 rounds per detection.
template <int NW, int NG> __global__ void __launch_bounds__(64 * NW)
igr_query2_kernel(Query Qv, Query Qg, int split, const double *W0, const double *b0, const double *Wp, const double *bh,
                  const double *W8, const double *b8)
{
    DSS_DYN_LDS(double, X);
    const int bid = (int)blockIdx.x;
    if (bid < split) igr_body<NW, NG, MODE_VALUE>(Qv, bid, split, X, W0, b0, Wp, bh, W8, b8);
    else igr_body<NW, NG, MODE_XYZ>(Qg, bid - split, (int)gridDim.x - split, X, W0, b0, Wp, bh, W8, b8);
}

template <int NW, int NG, int MODE>
void launch(const Query &Q, const DssIgrNet &N, int n_cap, int est, hipStream_t stream)
{
    constexpr int PTS = (MODE == MODE_VALUE ? 16 : 4) * NG;
    const size_t lds = (size_t)16 * NG * LDX * sizeof(double);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)igr_query_kernel<NW, NG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long tiles = ((long)n_cap + PTS - 1) / PTS;
    if (Q.n_dev) {
        // The length is only known on the device.  The grid is persistent over the tiles, so ANY size is correct; what it
        // costs is dispatch time (a workgroup that finds nothing to do still takes ~0.1 us to launch: 2048 of them are
        // 200 us, several times the evaluation of a small list).  `est` = the caller's expectation (the length the same
        // list had in the previous detection), or < 0: fill the chip.
        // (twice the expectation, and never fewer than 128 workgroups: a list that turns out far longer than expected --
        // the step in which bodies first touch -- must not crawl through a handful of workgroups)
        long want = est < 0 ? 512 : (2L * est + PTS - 1) / PTS + 2;
        if (want < 128) want = 128;
        const long cap = 256L * 2 * 4;
        tiles = want < tiles ? want : tiles;
        if (tiles > cap) tiles = cap;
    }
    if (tiles < 1) tiles = 1;
    hipLaunchKernelGGL((igr_query_kernel<NW, NG, MODE>), dim3((unsigned)tiles), dim3(64 * NW), lds, stream, Q, N.W0, N.b0, N.Wp,
                       N.bh, N.W8, N.b8);
}

// workgroups for a device-length list of which `est` points are expected (see launch above)
template <int PTS> inline long tiles_for(int n_cap, int est)
{
    long tiles = ((long)n_cap + PTS - 1) / PTS, want = est < 0 ? 512 : (2L * est + PTS - 1) / PTS + 2;
    if (want < 128) want = 128;
    tiles = want < tiles ? want : tiles;
    if (tiles > 256L * 2 * 4) tiles = 256L * 2 * 4;
    return tiles < 1 ? 1 : tiles;
}
template <int NW, int NG> void launch2(const Query &Qv, const Query &Qg, const DssIgrNet &N, int n_cap, int estv, int estg, hipStream_t stream)
{
    const size_t lds = (size_t)16 * NG * LDX * sizeof(double);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)igr_query2_kernel<NW, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long tv = tiles_for<16 * NG>(n_cap, estv), tg = tiles_for<4 * NG>(n_cap, estg);
    hipLaunchKernelGGL((igr_query2_kernel<NW, NG>), dim3((unsigned)(tv + tg)), dim3(64 * NW), lds, stream, Qv, Qg, (int)tv, N.W0, N.b0,
                       N.Wp, N.bh, N.W8, N.b8);
}

template <int MODE> void launch_mode(const Query &Q, const DssIgrNet &N, int n_cap, int est, hipStream_t stream)
{
    // big batches (grid builds, the candidate rounds of a large scene batch): 4 waves share 2 row groups (32 KB of LDS: four
    // workgroups per CU, so that one's softplus epilogue hides under another's MFMAs; measured 41.9 / 58.1 TFLOP/s without /
    // with tangents against 39.7 / 57.8 for 4 row groups); small ones (a Frank-Wolfe round moves a few points per item): one row group per
    // workgroup, a quarter of the latency of a pass and four times as many workgroups to spread over the chip.
    // All variants give bit-identical results (a point's row never mixes with its tile-mates').
    const int n = Q.n_dev ? (est < 0 ? n_cap : est) : n_cap;
    if (n >= 16 * 1024) launch<4, 2, MODE>(Q, N, n_cap, est, stream);
    else if (n >= 2 * 1024) launch<2, 2, MODE>(Q, N, n_cap, est, stream);
    else launch<4, 1, MODE>(Q, N, n_cap, est, stream);
}

inline bool net_ok(const DssIgrNet *N) { return N && N->W0 && N->b0 && N->Wp && N->bh && N->W8 && N->b8; }

}  // namespace

namespace dss {
// used by the neural narrow phase (narrowphase_igr.hip) and the reverse sweep: one evaluation round over a device-side list
int launch_igr_list(const DssIgrNet &N, const double *pts, const int *lat_idx, const double *latents, int lat_stride,
                    const int *n_dev, int n_cap, int mode, double *sdf, double *grad, hipStream_t stream, int est)
{
    Query Q{pts, lat_idx, latents, lat_stride, n_dev, n_cap, sdf, grad};
    if (mode == MODE_VALUE) launch_mode<MODE_VALUE>(Q, N, n_cap, est, stream);
    else if (mode == MODE_LATENT) launch_mode<MODE_LATENT>(Q, N, n_cap, est, stream);
    else launch_mode<MODE_XYZ>(Q, N, n_cap, est, stream);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}
// the value list and the gradient list of one query round in one launch (both lengths in device memory)
int launch_igr_pair(const DssIgrNet &N, const double *pts_v, const int *lat_v, const int *n_v, double *sdf_v, const double *pts_g,
                    const int *lat_g, const int *n_g, double *sdf_g, double *grad_g, const double *latents, int lat_stride, int n_cap,
                    hipStream_t stream, int est_v, int est_g)
{
    Query Qv{pts_v, lat_v, latents, lat_stride, n_v, n_cap, sdf_v, nullptr}, Qg{pts_g, lat_g, latents, lat_stride, n_g, n_cap, sdf_g, grad_g};
    // variant by the work expected (a gradient point is four rows): see launch_mode
    const long rows = (est_v < 0 || est_g < 0) ? (long)n_cap : (long)est_v + 4L * est_g;
    if (rows >= 16 * 1024) launch2<4, 2>(Qv, Qg, N, n_cap, est_v, est_g, stream);
    else if (rows >= 2 * 1024) launch2<2, 2>(Qv, Qg, N, n_cap, est_v, est_g, stream);
    else launch2<4, 1>(Qv, Qg, N, n_cap, est_v, est_g, stream);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}
}  // namespace dss

extern "C" {

size_t dss_igr_packed_doubles(void) { return (size_t)7 * 8 * 32 * 64; }

int dss_igr_query(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                  const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream)
{
    if (!pts || !latent || !W0 || !b0 || !Wp || !bh || !W8 || !b8 || !sdf || !grad || n <= 0) return DSS_E_BADARG;
    const DssIgrNet N{W0, b0, Wp, bh, W8, b8};
    return dss::launch_igr_list(N, pts, nullptr, latent, 2, nullptr, n, MODE_XYZ, sdf, grad, (hipStream_t)stream, -1);
}

// The same network evaluation with the tangents seeded on the latent code instead of the point: grad [n][3] =
// (d phi / d latent_0, d phi / d latent_1, 0).  This is d phi / d theta of the MeshSDF backward (bodies.py:680-702) for
// an IGR body; the vertex normals come from dss_igr_query.
int dss_igr_query_latent_grad(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                              const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream)
{
    if (!pts || !latent || !W0 || !b0 || !Wp || !bh || !W8 || !b8 || !sdf || !grad || n <= 0) return DSS_E_BADARG;
    const DssIgrNet N{W0, b0, Wp, bh, W8, b8};
    return dss::launch_igr_list(N, pts, nullptr, latent, 2, nullptr, n, MODE_LATENT, sdf, grad, (hipStream_t)stream, -1);
}

int dss_igr_query_list(const DssIgrNet *net, const double *pts, const int *lat_idx, const double *latents, int lat_stride,
                       const int *n_dev, int n_cap, int mode, double *sdf, double *grad, void *stream)
{
    if (!net_ok(net) || !pts || !latents || !sdf || n_cap <= 0 || lat_stride < 2) return DSS_E_BADARG;
    if (mode != MODE_VALUE && mode != MODE_XYZ && mode != MODE_LATENT) return DSS_E_BADARG;
    if (mode != MODE_VALUE && !grad) return DSS_E_BADARG;
    return dss::launch_igr_list(*net, pts, lat_idx, latents, lat_stride, n_dev, n_cap, mode, sdf, grad, (hipStream_t)stream, -1);
}

}  // extern "C"

// igr_mlp.hip -- IGR neural SDF (decode_igr, sdf_physics/physics3d/utils.py:330-350) on gfx950 fp64 MFMA.
//
// Network (IGR_data/train_configs/bob_spot_setup.conf:38-45; ImplicitNet of the external IGR repo, restated from
// its published definition): input = [latent(2), xyz(3)], 9 Linear layers 5->128->128->128->123, skip concat
// [h, input]/sqrt(2) before layer 4, ->128 x4 ->1, Softplus(beta = 100) on all but the last, float64.
// query_sdfs needs phi AND d phi/d xyz (the reference gets the latter from autograd, bodies.py:730-745).
//
// This is the one dense-GEMM shaped piece of the hot path, so it runs on the matrix cores:
//   * forward-mode: every point carries (h, dh/dx, dh/dy, dh/dz); the 16-row MFMA tile holds 4 points x 4
//     quantities, ordered row = 4*quantity + point.  With v_mfma_f64_16x16x4 a lane's four accumulator registers
//     are rows q, q+4, q+8, q+12 of one column (q = lane>>4): exactly (value, dx, dy, dz) of ONE point and ONE
//     neuron, so bias + softplus + sigmoid-scaling of the tangents is lane-local.  No activations are kept for a
//     backward sweep and no transposed weights are needed.
//   * a workgroup of NW waves owns 4*NG points; wave w computes neuron tiles [w*8/NW, (w+1)*8/NW) for all of them
//     (NG x 8/NW accumulator tiles in AGPRs).  Activations of the current layer live in LDS ([16 NG rows][128+1]),
//     weights stream from L2 in MFMA-fragment order (host-packed: one coalesced 512-B load per fragment, reused by
//     all NG row groups), software-pipelined with the k-loop fully unrolled so ~60 fragment loads are in flight.
//     Two workgroups' waves share a SIMD, so one's softplus epilogue (VALU) hides under the other's MFMAs.
// Flops: 4 x 115 456 MAC per point = 0.92 MFLOP (fp64); matrix peak 256 CU x 128 flop/clk x 2.4 GHz = 78.6 TFLOP/s.
// Measured (128^3 grid, MI355X): 38.0 ms = 50.9 TFLOP/s = 65 % of peak.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "mfma_f64.h"
#include "wave_utils.h"

namespace {
using namespace dss;

constexpr int H = 128, NL = 9, DIN = 5, LDX = H + 1;

__device__ inline acc4 mfma(double a, double b, acc4 c) { return mfma_f64_16x16x4(a, b, c); }
__device__ inline double &comp(acc4 &v, int i) { return acc_comp(v, i); }

// softplus(z, beta=100, threshold=20) and its derivative, torch.nn.Softplus semantics
__device__ inline void softplus100(double z, double &h, double &dh)
{
    const double bz = 100.0 * z;
    const bool lin = bz > 20.0;
    const double e = exp(lin ? 20.0 : bz);   // branch-free: both sides evaluated on a clamped argument
    h = lin ? z : log1p(e) / 100.0;
    dh = lin ? 1.0 : e / (1.0 + e);
}

// Wp: packed weights.  Per hidden->hidden layer (7 of them): [tile t 0..7][kstep 0..31][lane 0..63] = W[16t + (lane&15)][4ks + (lane>>4)]
// NW waves per workgroup share the activations of NG groups of 4 points; wave w owns neuron tiles [w*8/NW, (w+1)*8/NW)
template <int NW, int NG> __global__ void __launch_bounds__(64 * NW)
igr_query_kernel(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                 const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, int wrt_latent)
{
    DSS_DYN_LDS(double, X);   // [ROWS][LDX]: row = 4*quantity + point (+16 for the second group of 4 points)
    constexpr int PTS = 4 * NG, ROWS = 4 * PTS, NT = 64 * NW, TPW = 8 / NW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, base = blockIdx.x * PTS;
    const int col = lane & 15, q = lane >> 4;

    // ---- layer 0 (K = 5) on the vector ALU: h0 = softplus(W0 [latent, xyz] + b0), tangents = sigma' * W0[:, 2+d]
    for (int e = tid; e < PTS * H; e += NT) {
        const int p = e / H, j = e % H, gp = base + p;
        double in[DIN] = {latent[0], latent[1], 0.0, 0.0, 0.0};
        if (gp < n) { in[2] = pts[3 * gp]; in[3] = pts[3 * gp + 1]; in[4] = pts[3 * gp + 2]; }
        double z = b0[j];
        for (int k = 0; k < DIN; ++k) z += W0[j * DIN + k] * in[k];
        double h, dh;
        softplus100(z, h, dh);
        const int r0 = 16 * (p / 4) + (p % 4);
        X[r0 * LDX + j] = h;
        // tangent seeds: the three point coordinates (inputs 2..4), or the two latent coordinates (inputs 0, 1)
        for (int d = 0; d < 3; ++d) X[(r0 + 4 * (d + 1)) * LDX + j] = wrt_latent ? (d < 2 ? dh * W0[j * DIN + d] : 0.0) : dh * W0[j * DIN + 2 + d];
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
                    const int k = j - (H - DIN), quant = (r % 16) / 4, p = 4 * (r / 16) + (r % 4), gp = base + p;
                    if (quant == 0) v = k < 2 ? latent[k] : (gp < n ? pts[3 * gp + k - 2] : 0.0);
                    else v = ((wrt_latent ? k : k - 2) == quant - 1 && (wrt_latent ? k < 2 : k >= 2)) ? 1.0 : 0.0;
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
        // epilogue: this lane owns (value, dx, dy, dz) of point q (of group g) for neuron 16 t + col
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int j = 16 * (wv * TPW + t) + col;
                double h, dh;
                softplus100(comp(acc[g][t], 0) + bias[t], h, dh);
                X[(16 * g + q) * LDX + j] = h;
                for (int d = 1; d < 4; ++d) X[(16 * g + 4 * d + q) * LDX + j] = dh * comp(acc[g][t], d);
            }
        __syncthreads();
    }
    // ---- layer 8 (one output): dot products on the vector ALU, rows split over lanes
    for (int r = tid; r < ROWS; r += NT) {
        double acc = 0.0;
        for (int j = 0; j < H; ++j) acc += W8[j] * X[r * LDX + j];
        const int quant = (r % 16) / 4, p = 4 * (r / 16) + (r % 4), gp = base + p;
        if (gp < n) {
            if (quant == 0) sdf[gp] = acc + b8[0];
            else grad[3 * gp + quant - 1] = acc;
        }
    }
}

template <int NW, int NG>
void launch(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp, const double *bh,
            const double *W8, const double *b8, int n, double *sdf, double *grad, int wrt_latent, hipStream_t stream)
{
    constexpr int PTS = 4 * NG;
    const size_t lds = (size_t)4 * PTS * LDX * sizeof(double);
#if !defined(DSS_EMU)
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)igr_query_kernel<NW, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
    hipLaunchKernelGGL((igr_query_kernel<NW, NG>), dim3((n + PTS - 1) / PTS), dim3(64 * NW), lds, stream, pts, latent, W0, b0,
                       Wp, bh, W8, b8, n, sdf, grad, wrt_latent);
}

}  // namespace

extern "C" {


size_t dss_igr_packed_doubles(void) { return (size_t)7 * 8 * 32 * 64; }

int dss_igr_query(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                  const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream)
{
    if (!pts || !latent || !W0 || !b0 || !Wp || !bh || !W8 || !b8 || !sdf || !grad || n <= 0) return DSS_E_BADARG;
    // big batches (grid builds): 4 waves share 16 points, halving the L2 weight traffic; small ones (contact queries):
    // 2 waves x 8 points so the grid still covers the chip.  Both give bit-identical results.
    if (n >= 16 * 1024) launch<4, 4>(pts, latent, W0, b0, Wp, bh, W8, b8, n, sdf, grad, 0, (hipStream_t)stream);
    else launch<2, 2>(pts, latent, W0, b0, Wp, bh, W8, b8, n, sdf, grad, 0, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

// The same network evaluation with the tangents seeded on the latent code instead of the point: grad [n][3] =
// (d phi / d latent_0, d phi / d latent_1, 0).  This is d phi / d theta of the MeshSDF backward (bodies.py:680-702) for
// an IGR body; the vertex normals come from dss_igr_query.
int dss_igr_query_latent_grad(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                              const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream)
{
    if (!pts || !latent || !W0 || !b0 || !Wp || !bh || !W8 || !b8 || !sdf || !grad || n <= 0) return DSS_E_BADARG;
    if (n >= 16 * 1024) launch<4, 4>(pts, latent, W0, b0, Wp, bh, W8, b8, n, sdf, grad, 1, (hipStream_t)stream);
    else launch<2, 2>(pts, latent, W0, b0, Wp, bh, W8, b8, n, sdf, grad, 1, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

}  // extern "C"

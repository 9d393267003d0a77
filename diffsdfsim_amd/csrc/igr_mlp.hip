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
//   * one wavefront = 8 points (2 row groups) x 128 neurons: 16 accumulator tiles (128 VGPRs); activations of the
//     current layer live in LDS ([32 rows][128], 32 KB per wave), weights stream from L2 in MFMA-fragment order
//     (host-packed: one coalesced 512-B load per fragment, reused by both row groups).
// Flops: 4 x 114 816 MAC per point = 0.92 MFLOP (fp64); peak 78.6 TFLOP/s.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "wave_utils.h"

namespace {
using namespace dss;

constexpr int H = 128, NL = 9, DIN = 5, PTS = 8, ROWS = 4 * PTS, LDX = H + 1;

#if defined(DSS_EMU)
typedef struct { double x, y, z, w; } acc4;
#else
typedef double acc4 __attribute__((ext_vector_type(4)));
#endif

__device__ inline acc4 mfma(double a, double b, acc4 c)
{
#if defined(DSS_EMU)
    return dss_emu_mfma_f64_16x16x4(a, b, c);
#else
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#endif
}
__device__ inline double &comp(acc4 &v, int i)
{
#if defined(DSS_EMU)
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
#else
    return reinterpret_cast<double *>(&v)[i];
#endif
}

// softplus(z, beta=100, threshold=20) and its derivative, torch.nn.Softplus semantics
__device__ inline void softplus100(double z, double &h, double &dh)
{
    const double bz = 100.0 * z;
    if (bz > 20.0) { h = z; dh = 1.0; }
    else { const double e = exp(bz); h = log1p(e) / 100.0; dh = e / (1.0 + e); }
}

// Wp: packed weights.  Per hidden->hidden layer (7 of them): [tile t 0..7][kstep 0..31][lane 0..63] = W[16t + (lane&15)][4ks + (lane>>4)]
__global__ void __launch_bounds__(64)
igr_query_kernel(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                 const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad)
{
    DSS_DYN_LDS(double, X);   // [ROWS][LDX]: row = 4*quantity + point (+16 for the second group of 4 points)
    const int lane = lane_id(), base = blockIdx.x * PTS;
    const int col = lane & 15, q = lane >> 4;

    // ---- layer 0 (K = 5) on the vector ALU: h0 = softplus(W0 [latent, xyz] + b0), tangents = sigma' * W0[:, 2+d]
    for (int e = lane; e < PTS * H; e += WAVE) {
        const int p = e / H, j = e % H, gp = base + p;
        double in[DIN] = {latent[0], latent[1], 0.0, 0.0, 0.0};
        if (gp < n) { in[2] = pts[3 * gp]; in[3] = pts[3 * gp + 1]; in[4] = pts[3 * gp + 2]; }
        double z = b0[j];
        for (int k = 0; k < DIN; ++k) z += W0[j * DIN + k] * in[k];
        double h, dh;
        softplus100(z, h, dh);
        const int r0 = 16 * (p / 4) + (p % 4);
        X[r0 * LDX + j] = h;
        for (int d = 0; d < 3; ++d) X[(r0 + 4 * (d + 1)) * LDX + j] = dh * W0[j * DIN + 2 + d];
    }
    __syncthreads();

    for (int layer = 1; layer < NL - 1; ++layer) {
        const double *Wl = Wp + (size_t)(layer - 1) * 8 * 32 * 64, *bl = bh + (size_t)(layer - 1) * H;
        if (layer == 4) {
            // skip connection: x = cat([h3 (123), input (5)]) / sqrt(2)   (value rows get the input, tangent rows its Jacobian)
            for (int e = lane; e < ROWS * H; e += WAVE) {
                const int r = e / H, j = e % H;
                double v = X[r * LDX + j];
                if (j >= H - DIN) {
                    const int k = j - (H - DIN), quant = (r % 16) / 4, p = 4 * (r / 16) + (r % 4), gp = base + p;
                    if (quant == 0) v = k < 2 ? latent[k] : (gp < n ? pts[3 * gp + k - 2] : 0.0);
                    else v = (k - 2 == quant - 1) ? 1.0 : 0.0;
                }
                X[r * LDX + j] = v * 0.70710678118654752440;
            }
            __syncthreads();
        }
        acc4 acc[2][8];
        for (int g = 0; g < 2; ++g)
            for (int t = 0; t < 8; ++t) { comp(acc[g][t], 0) = 0; comp(acc[g][t], 1) = 0; comp(acc[g][t], 2) = 0; comp(acc[g][t], 3) = 0; }
        for (int ks = 0; ks < 32; ++ks) {
            // A fragments: lane holds X[row = lane&15 (+16 g)][k = 4 ks + (lane>>4)]
            const double a0 = X[col * LDX + 4 * ks + q], a1 = X[(16 + col) * LDX + 4 * ks + q];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double b = Wl[((size_t)t * 32 + ks) * 64 + lane];
                acc[0][t] = mfma(a0, b, acc[0][t]);
                acc[1][t] = mfma(a1, b, acc[1][t]);
            }
        }
        __syncthreads();
        // epilogue: this lane owns (value, dx, dy, dz) of point q (of group g) for neuron 16 t + col
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = 16 * t + col;
                double h, dh;
                softplus100(comp(acc[g][t], 0) + bl[j], h, dh);
                X[(16 * g + q) * LDX + j] = h;
                for (int d = 1; d < 4; ++d) X[(16 * g + 4 * d + q) * LDX + j] = dh * comp(acc[g][t], d);
            }
        __syncthreads();
    }
    // ---- layer 8 (one output): dot products on the vector ALU, rows split over lanes
    for (int r = lane; r < ROWS; r += WAVE) {
        double acc = 0.0;
        for (int j = 0; j < H; ++j) acc += W8[j] * X[r * LDX + j];
        const int quant = (r % 16) / 4, p = 4 * (r / 16) + (r % 4), gp = base + p;
        if (gp < n) {
            if (quant == 0) sdf[gp] = acc + b8[0];
            else grad[3 * gp + quant - 1] = acc;
        }
    }
}

}  // namespace

extern "C" {

size_t dss_igr_packed_doubles(void) { return (size_t)7 * 8 * 32 * 64; }

int dss_igr_query(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                  const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream)
{
    if (!pts || !latent || !W0 || !b0 || !Wp || !bh || !W8 || !b8 || !sdf || !grad || n <= 0) return DSS_E_BADARG;
    const size_t lds = (size_t)ROWS * LDX * sizeof(double);
    hipLaunchKernelGGL(igr_query_kernel, dim3((n + PTS - 1) / PTS), dim3(64), lds, (hipStream_t)stream, pts, latent, W0, b0,
                       Wp, bh, W8, b8, n, sdf, grad);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

}  // extern "C"

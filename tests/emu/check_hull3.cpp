// Test harness (CPU emulation, test infrastructure): the hull of one normal cluster as the thinning stage of the narrow phase
// takes it (csrc/np_common.h: cluster_hull, workgroup flavour, points in the global scratch) on a point set read from a file.
// usage: check_hull3 <points.bin (m doubles x 3)> <m> [wave]  -> prints the indices of the kept points
#define DSS_EMU 1
#define DSS_ALL_SHAPES 1
#include "../../diffsdfsim_amd/csrc/np_common.h"

namespace {
__global__ void hull_kernel(double *cb, int mc, int m, double eps)
{
    __shared__ ScratchT<BlockGroup> S;
    const HullGlobal P{cb, mc};
    cluster_hull(S, P, m, eps);
}
// the wavefront flavour: one wavefront, the cluster in its LDS scratch (up to WaveGroup::HCAP points)
__global__ void hull_wave_kernel(const double *pts, int m, double eps, int *flags)
{
    __shared__ ScratchT<WaveGroup> S;
    for (int k = threadIdx.x; k < m; k += 64) for (int d = 0; d < 3; ++d) S.hp[3 * k + d] = pts[3 * k + d];
    dss_wave_sync();
    cluster_hull(S, HullLds<WaveGroup>{&S}, m, eps);
    for (int k = threadIdx.x; k < m; k += 64) flags[k] = S.hflag[k];
}
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int m = atoi(argv[2]);
    std::vector<double> pts(3 * (size_t)m), cb(8 * (size_t)m, 0.0);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(pts.data(), sizeof(double), 3 * (size_t)m, f) != 3 * (size_t)m) return 3;
    fclose(f);
    for (int k = 0; k < m; ++k) for (int d = 0; d < 3; ++d) cb[(size_t)(3 + d) * m + k] = pts[3 * (size_t)k + d];   // HullGlobal rows 3-5
    if (argc > 3 && m <= WaveGroup::HCAP) {
        std::vector<int> fl(m, 0);
        const double *pp = pts.data(); int *fp = fl.data();
        hipLaunchKernelGGL(hull_wave_kernel, dim3(1), dim3(64), 0, nullptr, pp, m, 1e-3, fp);
        for (int k = 0; k < m; ++k) if (fl[k]) printf("%d\n", k);
        return 0;
    }
    double *cbp = cb.data();      // (the emulated launch captures its arguments by value)
    hipLaunchKernelGGL(hull_kernel, dim3(1), dim3(BlockGroup::BT), 0, nullptr, cbp, m, m, 1e-3);
    for (int k = 0; k < m; ++k) if ((int)cb[(size_t)6 * m + k]) printf("%d\n", k);
    return 0;
}

// Test harness (CPU emulation, test infrastructure): the hull of one normal cluster as the thinning stage of the narrow phase
// takes it (csrc/np_common.h: cluster_hull, workgroup flavour, points in the global scratch) on a point set read from a file.
// usage: check_hull3 <points.bin (m doubles x 3)> <m>  -> prints the indices of the kept points
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
    double *cbp = cb.data();      // (the emulated launch captures its arguments by value)
    hipLaunchKernelGGL(hull_kernel, dim3(1), dim3(BlockGroup::BT), 0, nullptr, cbp, m, m, 1e-3);
    for (int k = 0; k < m; ++k) if ((int)cb[(size_t)6 * m + k]) printf("%d\n", k);
    return 0;
}

// Test harness (CPU): the hand-written softplus of the network kernel (csrc/igr_mlp.hip) against long double.
#define DSS_EMU 1
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include "dss_device.h"
#define __device__
// the function under test, taken from the kernel source verbatim
#define SOFTPLUS_ONLY 1
namespace t {
#include "softplus_extract.inc"
}
int main()
{
    double worst_h = 0, worst_d = 0, wz = 0;
    long n = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (long i = 0; i <= 4000000; ++i) {
            // z from -8 to 0.4 on a fine grid (100 z from -800 to 40), and a denser sweep around 0
            const double z = pass ? -0.05 + 0.1 * (double)i / 4000000.0 : -8.0 + 8.4 * (double)i / 4000000.0;
            double h, dh;
            t::softplus100(z, h, dh);
            const long double y = 100.0L * (long double)z;
            const long double hr = y > 20.0L ? (long double)z : log1pl(expl(y)) / 100.0L;
            const long double dr = y > 20.0L ? 1.0L : 1.0L / (1.0L + expl(-y));
            const double eh = (double)fabsl((long double)h - hr) / (double)fmaxl(fabsl(hr), 1e-3L);   // relative to max(|h|, 1e-3)
            const double ed = (double)fabsl((long double)dh - dr);
            if (eh > worst_h) { worst_h = eh; wz = z; }
            if (ed > worst_d) worst_d = ed;
            ++n;
        }
    printf("%ld points: worst |h - ref| / max(|ref|, 1e-3) = %.3e (at z = %.6f), worst |dh - ref| = %.3e\n", n, worst_h, wz, worst_d);
    return worst_h < 1e-15 && worst_d < 1e-15 ? 0 : 1;
}

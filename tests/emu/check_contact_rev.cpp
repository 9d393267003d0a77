// Test harness (CPU, test infrastructure): the reverse-mode contact adjoint (csrc/contact_rev.h) against forward-mode dual
// numbers through the same geometry code (csrc/contact_geom.h), on random box / sphere / cylinder pairs.
// Prints the largest relative deviation; exit code 1 if it exceeds 1e-9.
#define DSS_EMU 1
#define DSS_ALL_SHAPES 0
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../diffsdfsim_amd/csrc/contact_rev.h"

using namespace dss;
static double rnd() { return rand() / (double)RAND_MAX; }

int main()
{
    srand(7);
    double worst = 0.0;
    int ncase = 0;
    for (int it = 0; it < 20000; ++it) {
        const int ty1 = rand() % 3, ty2 = rand() % 3, stable = rand() % 2, det = (rand() % 4) == 0;
        double P1[7], P2[7], prm1[3], prm2[3], tv[3][3], tg[3][3], abc[3], gbar[9];
        for (int i = 0; i < 4; ++i) { P1[i] = rnd() - 0.5; P2[i] = rnd() - 0.5; }
        double n1 = 0, n2 = 0;
        for (int i = 0; i < 4; ++i) { n1 += P1[i] * P1[i]; n2 += P2[i] * P2[i]; }
        for (int i = 0; i < 4; ++i) { P1[i] /= sqrt(n1); P2[i] /= sqrt(n2); }
        for (int i = 0; i < 3; ++i) { prm1[i] = 0.4 + 0.6 * rnd(); prm2[i] = 0.4 + 0.6 * rnd(); }
        for (int i = 0; i < 3; ++i) { P1[4 + i] = 0.3 * (rnd() - 0.5); P2[4 + i] = P1[4 + i] + 0.5 * (rnd() - 0.5); }
        // a triangle near body 1's surface (anywhere in its query cube will do)
        for (int v = 0; v < 3; ++v) for (int i = 0; i < 3; ++i) { tv[v][i] = 0.5 * prm1[i % 3] * (2 * rnd() - 1); tg[v][i] = rnd() - 0.5; }
        double s = 0;
        for (int i = 0; i < 3; ++i) { abc[i] = rnd(); s += abc[i]; }
        for (int i = 0; i < 3; ++i) abc[i] /= s;
        if (it % 2) {
            // the non-smooth places: the point exactly on a face / edge / corner of the box (q = 0), on the mid planes
            // (p = 0, ties of the inside maximum), on the cylinder's axis or rim, at a body's centre
            abc[0] = 1.0; abc[1] = 0.0; abc[2] = 0.0;
            for (int i = 0; i < 3; ++i) {
                const double h = ty1 == SHAPE_BOX ? prm1[i] / 2.0 : (ty1 == SHAPE_CYLINDER ? (i < 2 ? prm1[0] : prm1[1] / 2.0) : prm1[0]);
                const int pick = rand() % 6;
                tv[0][i] = pick == 0 ? 0.0 : pick == 1 ? h : pick == 2 ? -h : pick == 3 ? 0.5 * h : pick == 4 ? -0.25 * h : 1.2 * h * (2 * rnd() - 1);
            }
            if (ty1 == SHAPE_CYLINDER && it % 4 == 1) tv[0][rand() % 2] = 0.0;      // radial direction along an axis: exact unit vector
            if (ty1 == SHAPE_BOX && it % 6 == 1) { prm1[1] = prm1[0]; if (it % 12 == 1) prm1[2] = prm1[0]; }     // equal dims: ties in scale and inside
            if (it % 10 == 3) for (int i = 0; i < 3; ++i) P2[4 + i] = P1[4 + i];                                  // coincident centres
        }
        for (int i = 0; i < 9; ++i) gbar[i] = rnd() - 0.5;
        // forward mode, 20 seeds
        typedef Dual<20> D;
        BodyG<D> B1, B2;
        D pr1[3], pr2[3];
        for (int i = 0; i < 4; ++i) { B1.q[i] = D(P1[i]); B1.q[i].d[i] = 1.0; B2.q[i] = D(P2[i]); B2.q[i].d[7 + i] = 1.0; }
        for (int i = 0; i < 3; ++i) {
            B1.pos[i] = D(P1[4 + i]); B1.pos[i].d[4 + i] = 1.0;
            B2.pos[i] = D(P2[4 + i]); B2.pos[i].d[11 + i] = 1.0;
            pr1[i] = D(prm1[i]); pr1[i].d[14 + i] = 1.0;
            pr2[i] = D(prm2[i]); pr2[i].d[17 + i] = 1.0;
        }
        make_shape(B1.shape, ty1, pr1);
        make_shape(B2.shape, ty2, pr2);
        D tri[3][3];
        for (int v = 0; v < 3; ++v)
            for (int i = 0; i < 3; ++i) {
                D d(tv[v][i]);
                const int sel = ty1 == SHAPE_BOX ? i : ((ty1 == SHAPE_CYLINDER && i == 2) ? 1 : 0);
                d.d[14 + sel] = tg[v][i];
                tri[v][i] = d;
            }
        D n[3], p1[3], p2[3], pen;
        int st = stable;
        contact_from_bary(B1, B2, tri, abc, 1e-3, n, p1, p2, pen, &st, det);
        double fw[20];
        for (int k = 0; k < 20; ++k) {
            double a = 0;
            for (int i = 0; i < 3; ++i) a += gbar[i] * n[i].d[k] + gbar[3 + i] * p1[i].d[k] + gbar[6 + i] * p2[i].d[k];
            fw[k] = a;
        }
        double out[20];
        contact_vjp_rev(P1, P2, ty1, ty2, prm1, prm2, tv, tg, abc, gbar, stable, det, out);
        // Not comparable (the reference's own formulas amplify rounding noise there, in either mode and in torch -- quantities
        // that are zero in exact arithmetic get multiplied by the 1e12 of a clamped F.normalize):
        //   * a point a hair OUTSIDE a cylinder's curved side (the Newton step leaves it within an ulp of the surface):
        //     max(q, 0) ~ 1e-17, its norm clamped at 1e-12, times the rounding error of the radial unit vector (on the
        //     coordinate axes that vector is exact and the modes agree, which the special cases above cover);
        //   * a point exactly on an EDGE (two of the q_i zero): torch.max(q, 0) splits both ties and the normal (0, 1, 1)/sqrt 2
        //     no longer projects the clamped derivative out exactly.  A face (one zero) does, and is compared.
        {
            Shape<double> S1; make_shape(S1, ty1, prm1);
            double c0[3], dA, nA[3], c1[3];
            for (int i = 0; i < 3; ++i) c0[i] = tv[0][i] * abc[0] + tv[1][i] * abc[1] + tv[2][i] * abc[2];
            query_sdf(S1, c0, dA, nA, true);
            for (int i = 0; i < 3; ++i) c1[i] = c0[i] - dA * nA[i];
            bool noisy = false;
            for (int pass = 0; pass < 2; ++pass) {
                const double *c = pass ? c1 : c0;
                const double pu[3] = {c[0] / S1.scale, c[1] / S1.scale, c[2] / S1.scale};
                if (ty1 == SHAPE_CYLINDER) {
                    const double q0 = sqrt(pu[0] * pu[0] + pu[1] * pu[1]) - S1.hd[0], q1 = fabs(pu[2]) - S1.hd[1];
                    const double r = sqrt(fmax(q0, 0.0) * fmax(q0, 0.0) + fmax(q1, 0.0) * fmax(q1, 0.0));
                    noisy |= (r > 0.0 && r < 1e-12 && pu[0] != 0.0 && pu[1] != 0.0) || (fabs(q0) < 1e-13 && fabs(q1) < 1e-13);
                } else if (ty1 == SHAPE_BOX) {
                    int z = 0;
                    for (int i = 0; i < 3; ++i) z += fabs(fabs(pu[i]) - S1.hd[i]) < 1e-13;
                    noisy |= z >= 2;
                }
            }
            if (noisy) continue;
        }
        double scale = 1e-300, err = 0;
        for (int k = 0; k < 20; ++k) { scale = fmax(scale, fabs(fw[k])); err = fmax(err, fabs(fw[k] - out[k])); }
        if (scale < 1e-12) continue;
        ++ncase;
        if (err / scale > worst) {
            worst = err / scale;
            if (worst > 1e-9) {
                printf("case %d types %d %d stable %d detach %d: rel err %.3e\n", it, ty1, ty2, stable, det, worst);
                printf("  prm1 %.17g %.17g %.17g  tv0 %.17g %.17g %.17g abc %g %g %g\n", prm1[0], prm1[1], prm1[2], tv[0][0], tv[0][1], tv[0][2], abc[0], abc[1], abc[2]);
                for (int k = 0; k < 20; ++k) printf("  %2d fw % .6e rev % .6e\n", k, fw[k], out[k]);
                return 1;
            }
        }
    }
    printf("%d cases, worst relative deviation %.3e\n", ncase, worst);
    return 0;
}
